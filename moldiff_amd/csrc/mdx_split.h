// Split-precision matrix path of the row-owner kernels (round 4; opt-in, the exact fp32 path of mdx_row.h stays the default).
//
// Why: on gfx950 the f32-input MFMA runs at the f32 VECTOR rate (1/16 of the float16 MFMA rate, no xf32), and after three rounds
// the row-owner kernels sit at 0.80 of that ceiling.  Here every fp32 operand is split as  x = hi + lo  with hi = fp16(x),
// lo = fp16(x - hi)  (22 significand bits), and a product is  Whi Xhi + Whi Xlo + Wlo Xhi  on v_mfma_f32_16x16x32_f16 with fp32
// accumulation: 3/16 of the MFMA cycles, the dropped Wlo Xlo term is 2^-22 relative.  Measured on the 256 -> 256 LayerNorm chain
// (tools/ubench_split.hip, profiles/r4_ubench_split.txt): 1.9x the exact kernel at 16 rows x 2 waves per SIMD (bound by the
// L1/L2 weight stream, 32.6 TB/s, no longer by the matrix pipe), error against float64 3.1e-6 max / 3.4e-7 rms where the exact
// fp32 kernel has 2.8e-6 / 3.8e-7.
//
// What carries over from mdx_row.h unchanged: one wave owns 16 rows and all output features of every layer; the accumulator
// layout (lane = 16 q + c holds Y[row c][16 ft + 4 q + s]) is the C/D map of every 16x16 MFMA; weights stream L2 -> VGPR through
// the same 2-KiB-step register ring (WRing / ws_frag).  What changes:
//   * the B operand of k-group g (32 k-values) is, per lane, 8 halves = the lane's accumulators of feature tiles 2g and 2g+1
//     converted in place (to_xs) -- the accumulator-is-next-operand chaining survives, with the k permutation
//     k(g, q, t) = 32 g + 16 (t / 4) + 4 q + t % 4  baked into the weight pack (host: PackCtx::pack_stream_split);
//   * a stream step is a PAIR of half-steps of two 1-KiB fragments each: (hi of tiles 2ftp, 2ftp+1), then (lo of the same) --
//     the same bytes as the fp32 stream for K a multiple of 32 (K is zero-padded to one otherwise);
//   * the LOW halves are stored scaled by 2^MDX_LO_SHIFT (= 2^11): lo = fp16((x - hi) 2^11) is then a NORMAL float16 number for
//     every |x| down to ~1e-4 (unscaled it would fall into float16's subnormal range already for |x| < 0.25 and the operand would
//     keep only ~18 bits at |x| ~ 0.01 -- measured on the guidance gradient, which is that sensitive).  The cross terms
//     Whi Xlo + Wlo Xhi are therefore accumulated apart from Whi Xhi, in two extra accumulators per feature-tile pair that live
//     for one k-loop only (8 registers), and folded in as  y += t 2^-11  when the pair is finished (exact scaling).
//     Nothing else is scaled: biases, LayerNorm, gates and stores are the fp32 code of the exact kernels, unchanged.
//   * range: operands must stay below float16's 65504 (the host refuses weights beyond it; activations are LayerNorm-bounded
//     products of O(1) quantities in this network).  Magnitudes below 6e-5 lose relative, not absolute, precision (<= 3e-8).
#pragma once
#include "mdx_row.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MDX_LO_SHIFT 11
#ifndef MDX_SPLIT_TERMS
// 3: Whi Xhi + Whi Xlo + Wlo Xhi (shipped);  4: + Wlo Xlo in an accumulator pair of its own (scale 2^22) -- measured: 8 % slower on
// the chain, error 1.49e-6 instead of 1.52e-6 (profiles/r4_ubench_split.txt): the operands' 22-bit representation, not the dropped
// term, is what is left
#define MDX_SPLIT_TERMS 3
#endif
constexpr float MDX_LO_UP = 2048.0f;           // 2^MDX_LO_SHIFT: scale of every low half (weights: host pack; activations: to_xs)
constexpr float MDX_LO_DOWN = 1.0f / 2048.0f;

// hi / lo operand of KG k-groups of 32
template <int KG>
struct XS {
  h8 hi[KG][RR], lo[KG][RR];
};

// accumulators (fp32) of FT feature tiles -> split operand of (FT + 1) / 2 k-groups; a missing odd tile is zero
template <int FT>
__device__ __forceinline__ void to_xs(XS<(FT + 1) / 2>& o, const f32x4 (&y)[FT][RR]) {
#pragma unroll
  for (int g = 0; g < (FT + 1) / 2; ++g)
#pragma unroll
    for (int rt = 0; rt < RR; ++rt)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int ft = 2 * g + t / 4;
        const float v = ft < FT ? y[ft][rt][t % 4] : 0.f;
        if (MDX_ABL & 8) {  // timing-only ablation: no conversion arithmetic (wrong results)
          o.hi[g][rt][t] = __builtin_bit_cast(_Float16, (unsigned short)(__float_as_uint(v) >> 16));
          o.lo[g][rt][t] = __builtin_bit_cast(_Float16, (unsigned short)(__float_as_uint(v)));
          continue;
        }
        const _Float16 h = (_Float16)v;
        o.hi[g][rt][t] = h;
        o.lo[g][rt][t] = (_Float16)((v - (float)h) * MDX_LO_UP);
      }
}

// y[ft][rt] += sum_k W[ft][k] x[k][rt]     (FT even; `ring` holds the first MDX_RING half-steps of this stream)
// half-step p = (ftp * KG + g) * 2 + h:  h = 0 the hi fragments of tiles (2 ftp, 2 ftp + 1) for k-group g, h = 1 their lo fragments
// (scaled by 2^MDX_LO_SHIFT like the operand's lo halves).  Per tile pair: y += Whi Xhi directly; t += Whi Xlo + Wlo Xhi; y += t 2^-11.
template <int KG, int FT>
__device__ __forceinline__ void rgemm_s_primed(f32x4 (&y)[FT][RR], const XS<KG>& x, const WS& w, WRing& ring, const WS& wnext) {
  static_assert(FT % 2 == 0, "feature tiles come in pairs");
  constexpr int NP = (FT / 2) * KG * 2;
#ifndef MDX_SPLIT_PRIME_AHEAD
#define MDX_SPLIT_PRIME_AHEAD 3   // half-steps before the end of a GEMM at which the next stream's first fragments are requested
#endif
  constexpr int PRIME_AT = NP > MDX_SPLIT_PRIME_AHEAD ? NP - MDX_SPLIT_PRIME_AHEAD : 0;
  WRing nx;
  f32x4 t0[RR], t1[RR], u0[RR], u1[RR];
  __builtin_amdgcn_s_setprio(0);
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr int ftp = p / (2 * KG), g = (p / 2) % KG, h = p % 2;
    const h8 a0 = __builtin_bit_cast(h8, ring.a[p % MDX_RING][0]), a1 = __builtin_bit_cast(h8, ring.a[p % MDX_RING][1]);
    if constexpr (p + MDX_RING < NP && !(MDX_ABL & 32)) {
      ring.a[p % MDX_RING][0] = ws_frag(w, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING));
      ring.a[p % MDX_RING][1] = ws_frag(w, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING) + 1);
    }
    if constexpr (p == PRIME_AT) ring_prime(nx, wnext);
    if constexpr (g == 0 && h == 0) {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) t0[rt] = t1[rt] = u0[rt] = u1[rt] = splat4(0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((MDX_ABL & 16) != 0 && (p % 8) != 0) return;  // timing-only ablation: 1/8 of the MFMAs
    // MFMAs on the same accumulator are kept two instructions apart
    if constexpr (h == 0) {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x.hi[g][rt], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x.hi[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x.lo[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x.lo[g][rt], t1[rt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x.hi[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x.hi[g][rt], t1[rt], 0, 0, 0);
      }
      if constexpr (MDX_SPLIT_TERMS == 4) {
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) {
          u0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x.lo[g][rt], u0[rt], 0, 0, 0);
          u1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x.lo[g][rt], u1[rt], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g == KG - 1 && h == 1) {  // the pair's k-loop is complete: fold the cross terms in
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        if constexpr (MDX_SPLIT_TERMS == 4) {
          t0[rt] = t0[rt] + u0[rt] * splat4(MDX_LO_DOWN);
          t1[rt] = t1[rt] + u1[rt] * splat4(MDX_LO_DOWN);
        }
        y[2 * ftp][rt] = y[2 * ftp][rt] + t0[rt] * splat4(MDX_LO_DOWN);
        y[2 * ftp + 1][rt] = y[2 * ftp + 1][rt] + t1[rt] * splat4(MDX_LO_DOWN);
      }
    }
  });
  ring = nx;
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) asm volatile("" : "+v"(y[ft][rt]));  // see rgemm (mdx_row.h)
}

// The same product with a ring that runs THROUGH the GEMM boundary (MDX_SPLIT_SEAMLESS builds: edge kernel A, mdx_edge2s.hip): the
// slot a half-step frees is refilled at once with what the wave needs MDX_RING half-steps later -- this stream's, or, in the last
// MDX_RING steps, the next stream's first half-steps -- so there is no second set of ring registers for the next stream's prime
// (64 VGPRs at MDX_RING = 8, live exactly where the accumulators and both operand halves are: the kernel parked operands in AGPRs
// for it, profiles/r4_split_phase_trace.txt) and every GEMM starts with its first MDX_RING half-steps in registers instead of 3.
// The ring position must be the same at every GEMM entry, so a GEMM occupies a multiple of MDX_RING half-steps: the 4- and
// 12-half-step ones run 4 idle steps that only turn the ring (their loads land in the stream pack's zero padding, MDX_RING_PAD).
template <int KG, int FT>
__device__ __forceinline__ void rgemm_s_seamless(f32x4 (&y)[FT][RR], const XS<KG>& xin, const WS& w, WRing& ring, const WS& wnext) {
  static_assert(FT % 2 == 0, "feature tiles come in pairs");
  const XS<KG>& x = xin;
  constexpr int NP = (FT / 2) * KG * 2;
  constexpr int NPL = (NP + MDX_RING - 1) / MDX_RING * MDX_RING;  // half-steps of ring rotation
  static_assert(NPL - NP <= 4, "idle steps read the pack's zero padding (4 half-steps)");
  f32x4 t0[RR], t1[RR];
  __builtin_amdgcn_s_setprio(0);
  static_for<0, NPL>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr int ftp = p / (2 * KG), g = (p / 2) % KG, h = p % 2;
    const h8 a0 = __builtin_bit_cast(h8, ring.a[p % MDX_RING][0]), a1 = __builtin_bit_cast(h8, ring.a[p % MDX_RING][1]);
    if constexpr (!(MDX_ABL & 32)) {
      if constexpr (p + MDX_RING < NPL) {
        ring.a[p % MDX_RING][0] = ws_frag(w, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING));
        ring.a[p % MDX_RING][1] = ws_frag(w, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING) + 1);
      } else {
        ring.a[p % MDX_RING][0] = ws_frag(wnext, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING - NPL));
        ring.a[p % MDX_RING][1] = ws_frag(wnext, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING - NPL) + 1);
      }
    }
    if constexpr (p >= NP) return;  // idle step
    if constexpr (g == 0 && h == 0) {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) t0[rt] = t1[rt] = splat4(0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((MDX_ABL & 16) != 0 && (p % 8) != 0) return;
    if constexpr (h == 0) {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x.hi[g][rt], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x.hi[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x.lo[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x.lo[g][rt], t1[rt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x.hi[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x.hi[g][rt], t1[rt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g == KG - 1 && h == 1) {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        y[2 * ftp][rt] = y[2 * ftp][rt] + t0[rt] * splat4(MDX_LO_DOWN);
        y[2 * ftp + 1][rt] = y[2 * ftp + 1][rt] + t1[rt] * splat4(MDX_LO_DOWN);
      }
    }
  });
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) asm volatile("" : "+v"(y[ft][rt]));
}

#ifndef MDX_SPLIT_SEAMLESS
#define MDX_SPLIT_SEAMLESS 0
#endif
template <int KG, int FT>
__device__ __forceinline__ void rgemm_s(f32x4 (&y)[FT][RR], const XS<KG>& x, const WS& w, WRing& ring, const WS& wnext) {
  if constexpr (MDX_SPLIT_SEAMLESS) rgemm_s_seamless<KG, FT>(y, x, w, ring, wnext);
  else rgemm_s_primed<KG, FT>(y, x, w, ring, wnext);
}

// the same with an fp32 operand in accumulator layout (KG16 tiles of 16 features): converted on the way in
template <int KG16, int FT>
__device__ __forceinline__ void rgemm_x(f32x4 (&y)[FT][RR], const f32x4 (&x)[KG16][RR], const WS& w, WRing& ring, const WS& wnext) {
  XS<(KG16 + 1) / 2> xs;
  to_xs<KG16>(xs, x);
  rgemm_s<(KG16 + 1) / 2, FT>(y, xs, w, ring, wnext);
}

// floats of a split stream of F output features over K inputs (K rounded up to 32, F to 32) without its ring padding: the offset
// of feature-tile pair ftp inside a stream is ftp * (K32 / 32) * 1024 floats
__host__ __device__ constexpr int split_stream_pair_floats(int K) { return ((K + 31) / 32) * 1024; }
