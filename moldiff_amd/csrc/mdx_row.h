// Device-side building blocks of the "row-owner" fused edge kernels (gfx950 / CDNA4 only) -- round 2.
//
// Measured premise (tools/ubench_chain.hip, profiles/r2_ubench_chain.txt): on gfx950 the f32-input MFMA runs on the same
// FMA lanes as the vector ALU, so VALU work does not hide under v_mfma_f32_16x16x4_f32 and a second wave on the SIMD buys
// nothing; what pays is (a) MFMAs issued back to back by ONE wave per SIMD, (b) as few VALU instructions as possible,
// (c) no barriers.  Hence:
//   * one wave owns 16*R consecutive rows (edges) and ALL output features of every layer;
//   * a layer's output accumulators ARE the next layer's B operand: acc[ft][rt][s] = Y[row 16 rt + c][16 ft + 4 q + s]
//     (lane = 16 q + c) is exactly the B fragment of k-group ft when the weights are packed with k in the order
//     16 g + 4 q + s -- activations never leave the registers, LayerNorm is wave-local, there is no LDS traffic and no
//     __syncthreads() anywhere in the kernel;
//   * weights stream L2 -> VGPR through a two-deep register ring in consumption order ("stream pack", host:
//     PackCtx::pack_stream): step p = ftp*KG + g carries the two fragments (2 ftp + j, g), j = 0,1, 2 KiB contiguous;
//   * 512 registers per wave (launch_bounds(256, 1)).
#pragma once
#include <type_traits>
#include "mdx_tile.h"

// compile-time loop: f(integral_constant<int, I>) for I in [B, E) -- guarantees static register-array indices
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

#ifndef MDX_ABL
#define MDX_ABL 0  // timing-only ablations (wrong results): 1 no stores, 2 no row gathers, 4 weight stream served from L1
#endif
#ifndef MDX_RING
#define MDX_RING 2  // steps (2 KiB each) of the weight stream in flight per wave
#endif
struct WRing {
  f32x4 a[MDX_RING][2];
};

// first MDX_RING steps of a stream (w already carries the +lane offset; every stream pack ends in MDX_RING_PAD zero steps,
// so priming a stream shorter than the ring stays in bounds)
__device__ __forceinline__ void ring_prime(WRing& r, const f32x4* __restrict__ w) {
#pragma unroll
  for (int p = 0; p < MDX_RING; ++p) {
    r.a[p][0] = w[(size_t)(2 * p) * 64];
    r.a[p][1] = w[(size_t)(2 * p + 1) * 64];
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
//   wnext : stream of the NEXT GEMM this wave will run (or nullptr): its first steps are requested a few steps before this
//           GEMM ends and are in `ring` on return, so the L2 latency of a layer's first fragments never sits between layers
//   hook(integral_constant<int, p>) runs in the load slot of step p (before its MFMAs): the caller's place for row gathers
//           and other prefetches that should travel under this GEMM
template <int KG, int FT, int R, class Hook = NoHook>
__device__ __forceinline__ void rgemm(f32x4 (&y)[FT][R], const f32x4 (&x)[KG][R], const f32x4* __restrict__ w, WRing& ring,
                                      const f32x4* __restrict__ wnext = nullptr, Hook&& hook = Hook{}) {
  static_assert(FT % 2 == 0, "feature tiles come in pairs");
  constexpr int NP = (FT / 2) * KG;
  constexpr int PRIME_AT = NP > 3 ? NP - 3 : 0;
  WRing nx;
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value;
    constexpr int ftp = p / KG, g = p % KG;
    const f32x4 a0 = ring.a[p % MDX_RING][0], a1 = ring.a[p % MDX_RING][1];
    if constexpr (p + MDX_RING < NP) {
      ring.a[p % MDX_RING][0] = w[(size_t)(2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING)) * 64];
      ring.a[p % MDX_RING][1] = w[(size_t)(2 * ((MDX_ABL & 4) ? p % 4 : p + MDX_RING) + 1) * 64];
    }
    if constexpr (p == PRIME_AT)
      if (wnext) ring_prime(nx, wnext);
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
  if (wnext) ring = nx;
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

template <int FT, int R>
__device__ __forceinline__ void row_store(const f32x4 (&v)[FT][R], float* __restrict__ base, const int (&row)[R],
                                          const bool (&valid)[R], int ld, int q) {
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    if (!valid[rt] || (MDX_ABL & 1)) continue;
    float* p = base + (size_t)row[rt] * ld + 4 * q;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) stg4(p + 16 * ft, v[ft][rt]);
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
