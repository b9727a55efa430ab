// Device-side building blocks of the "row-owner" fused edge kernels (gfx950 / CDNA4 only) -- round 2.
//
// Measured premises (tools/ubench_chain.hip, profiles/r2_ubench_chain.txt): on gfx950 the f32-input
// MFMA runs on the same FMA lanes as the vector ALU, so VALU work never hides under v_mfma_f32_16x16x4_f32 (PMC: zero co-execution
// cycles); what pays is (a) MFMAs issued back to back, (b) as few VALU instructions as possible -- including address arithmetic --
// and (c) no barriers.  A second wave per SIMD buys nothing in a pure GEMM loop, but it covers the gather / store / LayerNorm
// phases of the real kernels, and at 16 rows per wave everything fits 256 registers.  Hence:
//   * one wave owns 16*R consecutive rows (edges) and ALL output features of every layer (R = 1, two waves per SIMD, in the
//     product build);
//   * a layer's output accumulators ARE the next layer's B operand: acc[ft][rt][s] = Y[row 16 rt + c][16 ft + 4 q + s]
//     (lane = 16 q + c) is exactly the B fragment of k-group ft when the weights are packed with k in the order
//     16 g + 4 q + s -- activations never leave the registers, LayerNorm is wave-local, there is no LDS tile and no
//     __syncthreads() after the constant prologue;
//   * weights stream L2 -> VGPR through a register ring of MDX_RING steps in consumption order ("stream pack", host:
//     PackCtx::pack_stream): step p = ftp*KG + g carries the two fragments (2 ftp + j, g), j = 0,1, 2 KiB contiguous,
//     fetched with buffer loads (scalar base + scalar fragment offset: no vector address arithmetic);
//   * GEMM results are pinned at the end of rgemm (LLVM would otherwise sink MFMA chains away from their ring loads).
#pragma once
#include <type_traits>
#include "mdx_tile.h"

#ifndef MDX_ABL
#define MDX_ABL 0  // timing-only ablations (wrong results): 1 no stores, 2 no row gathers, 4 weight stream served from L1, 8 / 16 / 32 (split
                   // kernels, mdx_split.h): no operand conversion arithmetic / one MFMA in eight / no weight loads at all
#endif
#ifndef MDX_RING
#define MDX_RING 2  // steps (2 KiB each) of the weight stream in flight per wave
#endif
struct WRing {
  f32x4 a[MDX_RING][2];
};

// A weight stream as the ring sees it: a buffer resource (4 scalar registers: base, no stride, unbounded, raw dword format)
// plus the lane's byte offset.  buffer_load takes the fragment offset as a scalar/immediate operand, so walking a stream costs
// no vector ALU work and no 64-bit per-lane address -- with global_load every 4 KiB of stream needed a v_add_co/v_addc pair
// (VALU does not overlap the f32 MFMAs on this core) and hoisted lane addresses were the main source of register spills.
struct WS {
  __amdgpu_buffer_rsrc_t r;
  unsigned off;
};
__device__ __forceinline__ WS make_ws(const float* p, unsigned lane_off) {
  asm volatile("" : "+s"(p));  // per use: two scalar registers now instead of a hoisted descriptor per stream for the whole loop
  return WS{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, -1, 0x00020000), lane_off};
}
// fragment `frag` (64 lanes x 16 bytes) of a stream pack
__device__ __forceinline__ f32x4 ws_frag(const WS& w, int frag) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.r, w.off, frag * 1024, 0));
}

// first MDX_RING steps of a stream (every stream pack ends in MDX_RING_PAD zero steps, so priming a stream shorter than the
// ring stays in bounds)
__device__ __forceinline__ void ring_prime(WRing& r, const WS& w) {
  if (MDX_ABL & 32) return;  // timing-only ablation: no weight loads at all (the ring keeps whatever it held)
#pragma unroll
  for (int p = 0; p < MDX_RING; ++p) {
    r.a[p][0] = ws_frag(w, 2 * p);
    r.a[p][1] = ws_frag(w, 2 * p + 1);
  }
}

// sigmoid on the hardware transcendentals: 1 / (1 + 2^(-x log2 e)) with v_exp_f32 and v_rcp_f32 (1 ulp each; exact limits
// 0 and 1 for |x| large).  On this core VALU work does not hide under the f32 MFMAs, and the IEEE expf + division form
// costs ~27 VALU instructions per value against 5 here (tests/test_gpu_fullsize.py arbitrates the accuracy against fp64).
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ f32x4 fast_sigmoid4(f32x4 v) {
  f32x4 r = {fast_sigmoid(v[0]), fast_sigmoid(v[1]), fast_sigmoid(v[2]), fast_sigmoid(v[3])};
  return r;
}
#ifndef MDX_FAST_SIGMOID
#define MDX_FAST_SIGMOID 1
#endif
__device__ __forceinline__ f32x4 row_sigmoid4(f32x4 v) { return MDX_FAST_SIGMOID ? fast_sigmoid4(v) : sigmoid4(v); }

struct NoHook {
  template <class P>
  __device__ __forceinline__ void operator()(P) const {}
};

// y[ft][rt] += sum_g W(ft, g) x[g][rt]      (FT even; `ring` must hold the first MDX_RING steps of this stream)
//   wnext : stream of the NEXT GEMM this wave will run (never null -- a null test is a branch per GEMM): its first steps are requested a few steps before this
//           GEMM ends and are in `ring` on return, so the L2 latency of a layer's first fragments never sits between layers
//   hook(integral_constant<int, p>) runs in the load slot of step p (before its MFMAs): the caller's place for row gathers
//           and other prefetches that should travel under this GEMM
template <int KG, int FT, int R, class Hook = NoHook>
__device__ __forceinline__ void rgemm(f32x4 (&y)[FT][R], const f32x4 (&x)[KG][R], const WS& w, WRing& ring, const WS& wnext,
                                      Hook&& hook = Hook{}) {
  static_assert(FT % 2 == 0, "feature tiles come in pairs");
  constexpr int NP = (FT / 2) * KG;
  constexpr int PRIME_AT = NP > 3 ? NP - 3 : 0;
  WRing nx;
  __builtin_amdgcn_s_setprio(0);  // the wave outside its GEMMs (gathers, stores, LayerNorm) goes first at the issue port: 1% faster
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr int ftp = p / KG, g = p % KG;
    const f32x4 a0 = ring.a[p % MDX_RING][0], a1 = ring.a[p % MDX_RING][1];
    if constexpr (p + MDX_RING < NP) {
      ring.a[p % MDX_RING][0] = ws_frag(w, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING));
      ring.a[p % MDX_RING][1] = ws_frag(w, 2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING) + 1);
    }
    if constexpr (p == PRIME_AT) ring_prime(nx, wnext);
    hook(pc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], x[g][rt][s], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], x[g][rt][s], y[2 * ftp + 1][rt], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  });
  ring = nx;
  __builtin_amdgcn_s_setprio(1);
  // The results are pinned here: MFMA builtins are pure, and when an accumulator's next use is far away (a pairwise product,
  // a running sum consumed sections later) LLVM's code sinking moves its whole MFMA chain down to that use -- across
  // sched_barriers, which only bind the machine scheduler -- while the ring loads stay put: fragments then wait in scratch.
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int rt = 0; rt < R; ++rt) asm volatile("" : "+v"(y[ft][rt]));
}

// y[ft][rt] = v[16 ft + 4 q ..]   (v: LDS or global, never null -- a null test here becomes one branch per feature tile)
template <int FT, int R>
__device__ __forceinline__ void row_bias(f32x4 (&y)[FT][R], const float* __restrict__ v, int q) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const f32x4 b = ldg4(v + 16 * ft + 4 * q);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) y[ft][rt] = b;
  }
}
template <int FT, int R>
__device__ __forceinline__ void row_zero(f32x4 (&y)[FT][R]) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int rt = 0; rt < R; ++rt) y[ft][rt] = splat4(0.f);
}

// v[ft][rt] = base[idx[rt] * ld + 16 ft + 4 q ..]
template <int FT, int R>
__device__ __forceinline__ void row_gather(f32x4 (&v)[FT][R], const float* __restrict__ base, const int (&idx)[R], int ld, int q) {
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    const float* p = base + (size_t)((MDX_ABL & 2) ? 0 : idx[rt]) * ld + 4 * q;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) v[ft][rt] = ldg4(p + 16 * ft);
  }
}

// NT: non-temporal (streaming) stores -- for data nobody reads back soon (the guidance tape), so that it does not displace weights
// and node rows in L2
template <int FT, int R, bool NT = false>
__device__ __forceinline__ void row_store(const f32x4 (&v)[FT][R], float* __restrict__ base, const int (&row)[R],
                                          const bool (&valid)[R], int ld, int q) {
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    if (!valid[rt] || (MDX_ABL & 1)) continue;
    float* p = base + (size_t)row[rt] * ld + 4 * q;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      if (NT) __builtin_nontemporal_store(v[ft][rt], reinterpret_cast<f32x4*>(p + 16 * ft));
      else stg4(p + 16 * ft, v[ft][rt]);
    }
  }
}

// LayerNorm (biased variance, eps 1e-5, affine) + optional ReLU over the FT*16 features of each row; wave-local:
// a row's features live in the four lanes c, c+16, c+32, c+48.  Two-pass moments like models/common.py's nn.LayerNorm.
template <int FT, int R>
__device__ __forceinline__ void row_layernorm(f32x4 (&y)[FT][R], const float* __restrict__ gamma, const float* __restrict__ beta,
                                              int q, bool relu = true) {
  constexpr float inv_n = 1.0f / (float)(FT * 16);
  float mean[R], rstd[R];
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) s += (y[ft][rt][0] + y[ft][rt][1]) + (y[ft][rt][2] + y[ft][rt][3]);
    mean[rt] = red_q(s) * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = y[ft][rt][r] - mean[rt];
        d2 = fmaf(d, d, d2);
      }
    rstd[rt] = 1.0f / sqrtf(red_q(d2) * inv_n + MDX_LN_EPS);
  }
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    // keep the affine-parameter loads next to their use (hoisted together they would cost 8*FT registers)
    if (ft % 4 == 0) __builtin_amdgcn_sched_barrier(0);
    const f32x4 gm = ldg4(gamma + 16 * ft + 4 * q), bt = ldg4(beta + 16 * ft + 4 * q);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const f32x4 v = (y[ft][rt] - splat4(mean[rt])) * splat4(rstd[rt]) * gm + bt;
      y[ft][rt] = relu ? relu4(v) : v;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// ---- backward building blocks (guidance gradient), row-local versions of mdx_tile.h's ln_xhat / ln_apply_relu / ln_relu_bwd ----
// x (pre-LayerNorm) -> x_hat in place, rstd per row tile
template <int FT, int R>
__device__ __forceinline__ void row_ln_xhat(f32x4 (&x)[FT][R], float (&rstd)[R]) {
  constexpr float inv_n = 1.0f / (float)(FT * 16);
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) s += (x[ft][rt][0] + x[ft][rt][1]) + (x[ft][rt][2] + x[ft][rt][3]);
    const float mean = red_q(s) * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      x[ft][rt] = x[ft][rt] - splat4(mean);
#pragma unroll
      for (int r = 0; r < 4; ++r) d2 = fmaf(x[ft][rt][r], x[ft][rt][r], d2);
    }
    rstd[rt] = 1.0f / sqrtf(red_q(d2) * inv_n + MDX_LN_EPS);
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) x[ft][rt] = x[ft][rt] * splat4(rstd[rt]);
  }
}

// y = relu(x_hat * gamma + beta)
template <int FT, int R>
__device__ __forceinline__ void row_ln_apply_relu(f32x4 (&y)[FT][R], const f32x4 (&xhat)[FT][R], const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int q) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const f32x4 gm = ldg4(gamma + 16 * ft + 4 * q), bt = ldg4(beta + 16 * ft + 4 * q);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) y[ft][rt] = relu4(xhat[ft][rt] * gm + bt);
  }
}

// g = dL/d relu(LN(x)) -> dL/dx in place:  gh = g * [x_hat gamma + beta > 0] * gamma ;  dx = rstd (gh - mean(gh) - x_hat mean(gh x_hat))
template <int FT, int R>
__device__ __forceinline__ void row_ln_relu_bwd(f32x4 (&g)[FT][R], const f32x4 (&xhat)[FT][R], const float (&rstd)[R],
                                                const float* __restrict__ gamma, const float* __restrict__ beta, int q) {
  constexpr float inv_n = 1.0f / (float)(FT * 16);
  float s1[R], s2[R];
#pragma unroll
  for (int rt = 0; rt < R; ++rt) s1[rt] = s2[rt] = 0.f;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    if (ft % 4 == 0) __builtin_amdgcn_sched_barrier(0);
    const f32x4 gm = ldg4(gamma + 16 * ft + 4 * q), bt = ldg4(beta + 16 * ft + 4 * q);
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      const f32x4 y = xhat[ft][rt] * gm + bt;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gh = (y[r] > 0.f) ? g[ft][rt][r] * gm[r] : 0.f;
        g[ft][rt][r] = gh;
        s1[rt] += gh;
        s2[rt] = fmaf(gh, xhat[ft][rt][r], s2[rt]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    const float m1 = red_q(s1[rt]) * inv_n, m2 = red_q(s2[rt]) * inv_n;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) g[ft][rt] = (g[ft][rt] - splat4(m1) - xhat[ft][rt] * splat4(m2)) * splat4(rstd[rt]);
  }
}

// sum_f w[f] * y[row][f] over the FT*16 features of each row -> one scalar per row tile (all four q lanes get it)
template <int FT, int R>
__device__ __forceinline__ void row_dot(const f32x4 (&y)[FT][R], const float* __restrict__ w, int q, float (&out)[R]) {
  float s[R];
#pragma unroll
  for (int rt = 0; rt < R; ++rt) s[rt] = 0.f;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const f32x4 wv = ldg4(w + 16 * ft + 4 * q);
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[rt] = fmaf(wv[r], y[ft][rt][r], s[rt]);
  }
#pragma unroll
  for (int rt = 0; rt < R; ++rt) out[rt] = red_q(s[rt]);
}

// ---- shared by the row-owner kernels (mdx_edge2.hip, mdx_bwd2.hip): tile shape, row indices, LDS constants ----
// Rows per wave and waves per SIMD are build parameters of each kernel file.  Measured on the bench workload (ms per step,
// kernel A / kernel B / backward): 32 rows x 1 wave  4.97 / 1.82 / 11.1;  16 rows x 2 waves  4.66 / 1.78 / 9.5 -- the second
// wave covers the gather, store and LayerNorm phases of the first, and at 16 rows every kernel fits 256 registers.
#ifndef MDX_RR
#define MDX_RR 1
#endif
#ifndef MDX_WPS
#define MDX_WPS 2   // waves per SIMD the row-owner kernels are compiled for
#endif
constexpr int RR = MDX_RR;            // row tiles per wave: 16 * RR edges
constexpr int ROWS = 16 * RR;
constexpr int PARK_FLOATS = ROWS * MDX_ND;  // per wave

struct RowTile {
  int row[RR], li[RR], ri[RR];
  float tt[RR];
  bool valid[RR];
  int pf[RR];  // EA_AGG kernels only: partial-row offset of the row's left node
  int cnt;     // EA_AGG kernels only: valid rows of the unit (wave-uniform)
};

__device__ __forceinline__ RowTile load_tile(const int* __restrict__ l, const int* __restrict__ r, const float* __restrict__ te,
                                             int e0, int E, int c) {
  RowTile t;
#pragma unroll
  for (int rt = 0; rt < RR; ++rt) {
    const int e = e0 + 16 * rt + c;
    t.valid[rt] = e < E;
    t.row[rt] = t.valid[rt] ? e : E - 1;  // clamped: loads stay in bounds, stores are predicated on valid
    t.li[rt] = l[t.row[rt]];
    t.ri[rt] = r[t.row[rt]];
    t.tt[rt] = te[t.row[rt]];
  }
  return t;
}

// unit = `cnt` (<= 16 RR) rows from edge e0 (graph-aligned units of the EA_AGG kernels); epo: per-edge partial-row offset
__device__ __forceinline__ RowTile load_tile_u(const int* __restrict__ l, const int* __restrict__ r, const float* __restrict__ te,
                                               const int* __restrict__ epo, int e0, int cnt, int E, int c) {
  RowTile t;
#pragma unroll
  for (int rt = 0; rt < RR; ++rt) {
    t.valid[rt] = 16 * rt + c < cnt;
    t.row[rt] = t.valid[rt] ? e0 + 16 * rt + c : e0;  // clamped to the unit's first row (always a row of the same graph)
    t.li[rt] = l[t.row[rt]];
    t.ri[rt] = r[t.row[rt]];
    t.tt[rt] = te[t.row[rt]];
    t.pf[rt] = epo[t.row[rt]];
  }
  t.cnt = cnt;
  return t;
}

// unit of the guidance backward (round 5): `cnt` (<= 16) positions from j0 of the BY-RIGHT edge order (col_eids); the tile's rows
// are the edge ids found there, its end points come from the plan's col_left / col_right (independent loads), pf = epo_r
template <int R = RR>
__device__ __forceinline__ RowTile load_tile_r(const int* __restrict__ col_eids, const int* __restrict__ col_left,
                                               const int* __restrict__ col_right, const float* __restrict__ te,
                                               const int* __restrict__ epo_r, int j0, int cnt, int c) {
  static_assert(R == 1, "one 16-row tile per wave");
  RowTile t;
  t.valid[0] = c < cnt;
  const int j = t.valid[0] ? j0 + c : j0;  // clamped to the unit's first position (always a row of the same graph)
  t.row[0] = col_eids[j];
  t.li[0] = col_left[j];
  t.ri[0] = col_right[j];
  t.pf[0] = epo_r[j];
  t.tt[0] = te[t.row[0]];
  t.cnt = cnt;
  return t;
}

// In-kernel segment sums of the EA_AGG kernels: y (accumulator layout: lane (q, c) holds features 16 ft + 4 q .. of row c) is
// summed over runs of rows with equal `key` (sorted; the unit's first `cnt` rows are valid) and every run's sum is stored as ONE
// row of `out` (FT*16 floats wide) at row index `prow` (taken from the run's rows; all rows of a run carry the same one).
// The tile goes through the wave's private LDS area once so that each lane owns four FEATURES of every row instead of four
// features of one row: the sums are then plain sequential adds in row order (the order of the CSR segment sum they replace) at
// 2 packed adds per row and lane, a run's sum is one contiguous 16-byte-per-lane store, and no cross-lane VALU work is spent
// (on this core VALU time is additive to MFMA time; a DPP scan of the 64 accumulator registers costs ~1,000 VALU issues per
// unit, this ~50).  XOR swizzle of the 16-byte column by the row: writes (16 rows x 4 adjacent columns per instruction) and reads
// (one row, 64 or 16 adjacent columns) are both bank-conflict free without padding, so the 16 KiB sigmoid parking area is reused.
// first half: the tile goes to the wave's LDS area (XOR-swizzled 16-byte columns)
template <int FT, int R = 1, int RT = 0>   // (R, RT: the tile is row tile RT of a wave that owns R of them)
__device__ __forceinline__ void seg_sum_put(const f32x4 (&y)[FT][R], float* wbuf, int lane) {
  constexpr int C4 = 4 * FT;  // 16-byte columns per row
  static_assert(C4 == 64 || C4 == 32 || C4 == 16 || C4 == 8, "256-, 128-, 64- or 32-wide rows");
  constexpr unsigned SWZ = C4 < 16 ? C4 - 1 : 15;  // rows are XOR-swizzled within the row's own columns
  asm volatile("" : "+v"(lane));  // opaque per call: the swizzled LDS addresses are cheap to rebuild and must not be hoisted out
                                  // of the persistent loop (they would live in scratch)
  const int c = lane & 15, q = lane >> 4;
  char* base = reinterpret_cast<char*>(wbuf);
  // physical column of (row c, column 4 ft + q) = (4 ft + q) ^ c = 16 (ft >> 2) + 4 ((ft & 3) ^ (c >> 2)) + (q ^ (c & 3)):
  // four address registers, the rest is an immediate offset
  const unsigned lo = (unsigned)(c * C4 + (q ^ (c & 3))) * 16u, cc = (unsigned)(c >> 2) & (SWZ >> 2);
  unsigned a4[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a4[j] = lo + (((unsigned)j ^ cc) << 6);
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) *reinterpret_cast<f32x4*>(base + a4[ft & 3] + 256 * (ft >> 2)) = y[ft][RT];
  // lanes read what OTHER lanes of the wave wrote: DS operations of a wave execute in order, the compiler only has to keep them so
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// second half: read the tile back transposed, sum each run in row order, store one row per run (may run any time before the area is
// written again).  RB rows are in flight at a time (4 RB registers).
template <int FT, int RB = 8>
__device__ __forceinline__ void seg_sum_flush(float* wbuf, int lane, int cnt, int key, int prow, float* __restrict__ out) {
  constexpr int C4 = 4 * FT;
  constexpr unsigned SWZ = C4 < 16 ? C4 - 1 : 15;
  static_assert(16 % RB == 0, "rows per burst");
  asm volatile("" : "+v"(lane));
  char* base = reinterpret_cast<char*>(wbuf);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int col = lane & (C4 - 1);
  const unsigned col16 = (unsigned)col * 16u;
  f32x4 acc = splat4(0.f);
  int k_r = __builtin_amdgcn_readlane(key, 0);
  // RB rows at a time; every condition below is wave-uniform (scalar branches)
  static_for<0, 16 / RB>([&](auto hc) {
    constexpr int h = decltype(hc)::value;
    if (RB * h < cnt) {
      f32x4 v[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) v[r] = *reinterpret_cast<const f32x4*>(base + (col16 ^ (16u * ((RB * h + r) & SWZ))) + (RB * h + r) * C4 * 16);
      static_for<0, RB>([&](auto rc) {
        constexpr int r = RB * h + decltype(rc)::value;
        const int k_next = __builtin_amdgcn_readlane(key, r < 15 ? r + 1 : 15);
        if (r < cnt) {
          acc = acc + v[r - RB * h];
          if (r + 1 == cnt || k_next != k_r) {  // last row of a run
            const int row = __builtin_amdgcn_readlane(prow, r);
            if ((C4 == 64 || lane < C4) && !(MDX_ABL & 1)) stg4(out + (size_t)row * (16 * FT) + 4 * col, acc);
            acc = splat4(0.f);
          }
        }
        k_r = k_next;
      });
    }
  });
  __builtin_amdgcn_wave_barrier();  // the area is rewritten (parking, next unit) only after every lane has read it
}

template <int FT, int R = 1, int RT = 0>
__device__ __forceinline__ void seg_sum_store(const f32x4 (&y)[FT][R], float* wbuf, int lane, int cnt, int key, int prow,
                                              float* __restrict__ out) {
  seg_sum_put<FT, R, RT>(y, wbuf, lane);
  seg_sum_flush<FT>(wbuf, lane, cnt, key, prow, out);
}

template <int FT>
__device__ __forceinline__ void mul_inplace(f32x4 (&y)[FT][RR], const f32x4 (&v)[FT][RR]) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) y[ft][rt] = y[ft][rt] * v[ft][rt];
}

// Small per-layer vectors (biases, LayerNorm affine parameters, time columns) are copied once per workgroup into LDS: a
// lone wave per SIMD has nobody to hide the L2 latency of these loads, a ds_read is an order of magnitude closer.
template <int OFF, int N>
__device__ __forceinline__ const float* lds_put(float* base, const float* __restrict__ src, int tid) {
  static_assert(OFF % 4 == 0 && N % 4 == 0 && N <= 4 * MDX_WG, "constant vector layout");
  if (4 * tid < N) sts4(base + OFF + 4 * tid, ldg4(src + 4 * tid));
  return base + OFF;
}

// ---- work distribution of the persistent kernels ------------------------------------------------------------------------
// A CU holds two workgroups of a row-owner kernel, i.e. every SIMD two waves, and the instruction arbiter serves the OLDER wave
// first: on the bench workload the units of the first-dispatched half of the grid (one workgroup per CU) take 140 / 46 / 195 us
// (kernel A / B / guidance backward) while those of the second half take 213 / 70 / 315 us (tools/trace_edge2.py,
// profiles/r3_trace_*).  With a static equal split the first half finishes after ~75 % of the kernel and idles while the second
// runs on alone at well under the two-wave throughput.  So workgroup b of the first half and workgroup b + #CUs of the second
// form a PAIR that shares one contiguous unit range and draws from it through a counter: whatever the two rates are, the pair
// finishes together.  (One counter per XCD or per grid balances just as well on paper but measured 10-25 % SLOWER than the
// static split: ~10^4 returning device-scope atomics per launch on 8 addresses serialise at the memory side, and a wave's
// consecutive units end up on different CUs, away from the node rows its L1 already holds.  Per pair: 8 waves and ~40 atomics
// per counter, each counter on its own 128-byte line, a pair walks ~40 consecutive units.)
// One relaxed device-scope atomic per unit, requested one unit ahead so that its latency sits under the current unit.  Which
// wave computes a unit does not enter any result (every unit owns its output rows).
// The counters live in device memory that is all zero between launches: every wave counts itself out in the pair's second word
// and the last one of the pair clears both for the next launch on the stream (launches sharing a set must be stream-ordered).
constexpr int MDX_WQ_STRIDE = 32;  // ints per pair: [0] next unit (relative to the pair's first), [1] waves that have left
struct WorkQ {
  int* ctr;     // MDX_WQ_PAIRS lines of MDX_WQ_STRIDE ints; nullptr: static split
  int npairs;   // min(grid, #CUs): workgroup b belongs to pair b % npairs
  int nunits;
};
constexpr int MDX_WQ_PAIRS = 512;  // lines per counter set (>= #CUs)

inline WorkQ make_workq(int* ctr, int nunits, int grid, int ncus) {
  WorkQ w{};
  w.npairs = std::max(1, std::min(std::min(grid, ncus), MDX_WQ_PAIRS));
  w.ctr = ctr;
  w.nunits = nunits;
  return w;
}

// the pair's view: its counter line, its unit range [beg, end) and its wave count
struct WorkPair {
  int* line;
  int beg, end, waves;
};
__device__ __forceinline__ WorkPair wq_pair(const WorkQ& w) {
  const int p = blockIdx.x % w.npairs;
  const int ri = xcd_remap(p, w.npairs);  // ranges in XCD order: neighbouring units (same molecule, same node rows) share an L2
  const int q = w.nunits / w.npairs, r = w.nunits % w.npairs;
  WorkPair k;
  k.line = w.ctr + (size_t)MDX_WQ_STRIDE * p;
  k.beg = ri * q + min(ri, r);
  k.end = k.beg + q + (ri < r ? 1 : 0);
  k.waves = (blockDim.x >> 6) * (gridDim.x / w.npairs + (p < (int)(gridDim.x % w.npairs) ? 1 : 0));
  return k;
}

// request (a VGPR that is only read by wq_take: the atomic's latency stays off the critical path) and wave-uniform result
__device__ __forceinline__ int wq_request(int* line, int lane) {
  int v = 0;
  if (lane == 0) v = __hip_atomic_fetch_add(line, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
__device__ __forceinline__ int wq_take(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void wq_leave(const WorkPair& k, int lane) {
  if (lane == 0 && __hip_atomic_fetch_add(k.line + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == k.waves - 1) {
    __hip_atomic_store(k.line, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(k.line + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
