// Device-side building blocks for the fused row-tile MLP kernels (gfx950 / CDNA4 only).
//
// Convention used by every fused kernel in this directory:
//   * a workgroup = 256 threads = 4 wave64s, owns a tile of TE = 16*ET rows (edges or nodes);
//   * activations of the tile live in LDS as X[row][feature] with a padded leading dimension
//     (ldx = K + 8 floats: conflict-free ds_read_b128 / ds_write_b128);
//   * a Linear layer  Y[row][f] = sum_k W[f][k] X[row][k]  is computed TRANSPOSED on the matrix
//     cores with v_mfma_f32_16x16x4_f32 (exact fp32, == an fmaf chain):  A operand = weights
//     (features x k, pre-packed on the host in fragment order so each wave-load is 1 KiB
//     contiguous), B operand = activations (k x rows, one ds_read_b128 per 16 k-values);
//   * each wave owns a slice of FTW feature tiles (16 features each) for ALL rows of the tile, so
//     the weight stream is wave-private (global -> VGPR, no LDS staging, no barrier) and the
//     accumulator layout is: lane (c = lane&15, q = lane>>4), acc[ft][et][r] =
//     Y[row = 16*et + c][feature = 16*(ft0+ft) + 4*q + r]  -- i.e. 4 consecutive features of one
//     row per accumulator, so every epilogue gather/store is a 16-byte vector access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MDX_WG 256
#ifndef MDX_GEMM_PINNED
#define MDX_GEMM_PINNED 1  // pin the two-stage prefetch of gemm_tile with sched_barriers (A/B in DESIGN.md)
#endif
#define MDX_LN_EPS 1e-5f

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void stg4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void sts4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 splat4(float v) { f32x4 r = {v, v, v, v}; return r; }
// IEEE expf + division on purpose: the v_exp_f32/v_rcp_f32 shortcut measured no speed-up (the kernels are
// MFMA-issue bound, not VALU bound) and pushed the 6-block position error from <1e-4 to 1.06e-4.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ f32x4 sigmoid4(f32x4 v) {
  f32x4 r = {sigmoidf_(v[0]), sigmoidf_(v[1]), sigmoidf_(v[2]), sigmoidf_(v[3])};
  return r;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
  f32x4 r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
  return r;
}

// Sum over the four 16-lane rows of a wave (lanes c, c+16, c+32, c+48), result in all four: the q-reduction
// of the accumulator layout.  gfx950's v_permlane16_swap / v_permlane32_swap are plain VALU ops; the
// __shfl_xor they replace lowers to ds_bpermute (an LDS round trip + lgkmcnt(0) per step).  Bit-identical sums.
__device__ __forceinline__ float red_q(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Padded leading dimension for a K-wide activation tile in LDS.
__host__ __device__ constexpr int mdx_ld(int K) { return K + 8; }

// First-group weight fragments of a layer, fetched ahead of time (before the previous layer's epilogue / barrier) so the
// L2 latency of a GEMM's first loads does not sit on the critical path between two layers (measured: with weights served
// from L1 the edge kernels run 10 % faster and the node kernel 24 %, all of it first-load latency).
template <int FTW>
__device__ __forceinline__ void load_w0(f32x4 (&a)[FTW], const float* __restrict__ Wp, int ft0, int lane) {
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)ft0 * 64 + lane;
#pragma unroll
  for (int ft = 0; ft < FTW; ++ft) a[ft] = wp[(size_t)ft * 64];
}

// compile-time loop: f(std::integral_constant<int, i>) for i in [B, E) -- guarantees static register-array indices
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

#if defined(MDX_TILE_RING) && MDX_TILE_RING > 2
// Ring variant of gemm_tile_impl (the kernels that run with fewer than two workgroups per CU, e.g. the node kernel at 256
// molecules): MDX_TILE_RING - 1 weight groups in flight instead of one, fetched with buffer loads (scalar base and group offset,
// one VGPR of lane offset), fully unrolled.  Same arithmetic and the same order of accumulation as the two-stage loop below.
template <int FTW, int ET, int K, bool PRE>
__device__ __forceinline__ void gemm_tile_impl(f32x4 (&acc)[FTW][ET], f32x4 (&a0)[FTW], const float* __restrict__ Wp, int FT,
                                               int ft0, const float* X, int ldx, int lane) {
  static_assert(K % 16 == 0, "K must be a multiple of 16");
  constexpr int G = K / 16, D = MDX_TILE_RING;
  const int c = lane & 15, q = lane >> 4;
  const float* xb = X + c * ldx + 4 * q;
  const float* base = Wp + (size_t)__builtin_amdgcn_readfirstlane(ft0) * 256;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
  const unsigned off = 16u * lane;
  const int gs = __builtin_amdgcn_readfirstlane(FT) * 1024;  // bytes between consecutive k-groups
  f32x4 a[D][FTW], b[2][ET];
  auto load_a = [&](f32x4(&dst)[FTW], int g) {
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft)
      dst[ft] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, g * gs + ft * 1024, 0));
  };
  auto load_b = [&](f32x4(&dst)[ET], int g) {
#pragma unroll
    for (int et = 0; et < ET; ++et) dst[et] = lds4(xb + et * 16 * ldx + g * 16);
  };
  static_for<0, D - 1>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    if constexpr (g < G) {
      if constexpr (PRE && g == 0) {
#pragma unroll
        for (int ft = 0; ft < FTW; ++ft) a[0][ft] = a0[ft];
      } else {
        load_a(a[g], g);
      }
    }
  });
  load_b(b[0], 0);
  static_for<0, G>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    if constexpr (g + D - 1 < G) load_a(a[(g + D - 1) % D], g + D - 1);
    if constexpr (g + 1 < G) load_b(b[(g + 1) & 1], g + 1);
    __builtin_amdgcn_sched_barrier(0);
#ifdef MDX_TILE_ABL_MFMA  // timing-only ablation (wrong results): one MFMA in eight
    if constexpr (g % 2 == 0)
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g % D][0][0], b[g & 1][0][0], acc[0][0], 0, 0, 0);
#else
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et)
          acc[ft][et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g % D][ft][s], b[g & 1][et][s], acc[ft][et], 0, 0, 0);
#endif
    __builtin_amdgcn_sched_barrier(0);
  });
}
#else
template <int FTW, int ET, int K, bool PRE>
__device__ __forceinline__ void gemm_tile_impl(f32x4 (&acc)[FTW][ET], f32x4 (&a0)[FTW], const float* __restrict__ Wp, int FT,
                                               int ft0, const float* X, int ldx, int lane) {
  static_assert(K % 16 == 0, "K must be a multiple of 16");
  const int c = lane & 15, q = lane >> 4;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)ft0 * 64 + lane;
  const float* xb = X + c * ldx + 4 * q;
  constexpr int G = K / 16;
  // Explicit two-stage software pipeline: the loads of group g+1 are issued ahead of the whole MFMA block of group g
  // (pinned with sched_barriers; left alone hipcc sinks them to ~10 MFMAs before their use).  Measured (round 1):
  // pinned == unpinned == a three-stage variant within 1 %.  What caps a lone wave at ~73 % of the MFMA rate in this
  // loop is the issue cost of its 7 memory instructions per 48 MFMAs; a second wave on the SIMD fills those slots
  // (micro-benchmark tools/ubench_gemm_tile.hip: 115 -> 140 TFLOP/s).
  f32x4 a1[FTW], b0[ET], b1[ET];
  auto load_a = [&](f32x4(&a)[FTW], int g) {
#pragma unroll
#ifdef MDX_ABL_WL1  // ablation: every group re-reads group 0 (weights stay in L1); results are wrong, timing only
    for (int ft = 0; ft < FTW; ++ft) a[ft] = wp[((size_t)(g & 0) * FT + ft) * 64];
#else
    for (int ft = 0; ft < FTW; ++ft) a[ft] = wp[((size_t)g * FT + ft) * 64];
#endif
  };
  auto load_b = [&](f32x4(&b)[ET], int g) {
#pragma unroll
    for (int et = 0; et < ET; ++et) b[et] = lds4(xb + et * 16 * ldx + g * 16);
  };
  auto mfma_group = [&](const f32x4(&a)[FTW], const f32x4(&b)[ET]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et)
          acc[ft][et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ft][s], b[et][s], acc[ft][et], 0, 0, 0);
  };
  if (!PRE) load_a(a0, 0);
  load_b(b0, 0);
#pragma unroll 1
  for (int g = 0; g < G; g += 2) {
    if (g + 1 < G) {
      load_a(a1, g + 1);
      load_b(b1, g + 1);
    }
#if MDX_GEMM_PINNED
    __builtin_amdgcn_sched_barrier(0);
#endif
    mfma_group(a0, b0);
#if MDX_GEMM_PINNED
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (g + 2 < G) {
      load_a(a0, g + 2);
      load_b(b0, g + 2);
    }
#if MDX_GEMM_PINNED
    __builtin_amdgcn_sched_barrier(0);
#endif
    if (g + 1 < G) mfma_group(a1, b1);
#if MDX_GEMM_PINNED
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
}

#endif  // MDX_TILE_RING

// ----------------------------------------------------------------------------------------------
// acc[ft][et] += W[16*(ft0+ft) .. +16][0..K) * X[16*et .. +16][0..K)^T
//   Wp : packed weights, float4 index ((g*FT + ft)*64 + lane), g = k/16   (host: PackCtx::pack_dense)
//   X  : LDS tile, row-major, leading dimension ldx (floats, multiple of 4)
// gemm_tile_pre: same, with the group-0 weight fragments already in `a0` (load_w0); a0 is clobbered.
// ----------------------------------------------------------------------------------------------
template <int FTW, int ET, int K>
__device__ __forceinline__ void gemm_tile(f32x4 (&acc)[FTW][ET], const float* __restrict__ Wp, int FT, int ft0,
                                          const float* X, int ldx, int lane) {
  f32x4 a0[FTW];
  gemm_tile_impl<FTW, ET, K, false>(acc, a0, Wp, FT, ft0, X, ldx, lane);
}
template <int FTW, int ET, int K>
__device__ __forceinline__ void gemm_tile_pre(f32x4 (&acc)[FTW][ET], f32x4 (&a0)[FTW], const float* __restrict__ Wp, int FT,
                                              int ft0, const float* X, int ldx, int lane) {
  gemm_tile_impl<FTW, ET, K, true>(acc, a0, Wp, FT, ft0, X, ldx, lane);
}

// Small layers (FTW*K/16 <= ~20 fragments per wave): fetch the wave's whole weight slice into registers ahead of time
// (e.g. before the tile's first barrier, so the L2 latency overlaps the tile load) and run the GEMM from LDS only.
template <int FTW, int K>
__device__ __forceinline__ void load_wfrag(f32x4 (&a)[K / 16][FTW], const float* __restrict__ Wp, int FT, int ft0, int lane) {
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)ft0 * 64 + lane;
#pragma unroll
  for (int g = 0; g < K / 16; ++g)
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) a[g][ft] = wp[((size_t)g * FT + ft) * 64];
}
template <int FTW, int ET, int K>
__device__ __forceinline__ void gemm_tile_reg(f32x4 (&acc)[FTW][ET], const f32x4 (&a)[K / 16][FTW], const float* X, int ldx,
                                              int lane) {
  const int c = lane & 15, q = lane >> 4;
  const float* xb = X + c * ldx + 4 * q;
#pragma unroll
  for (int g = 0; g < K / 16; ++g) {
    f32x4 b[ET];
#pragma unroll
    for (int et = 0; et < ET; ++et) b[et] = lds4(xb + et * 16 * ldx + g * 16);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et)
          acc[ft][et] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g][ft][s], b[et][s], acc[ft][et], 0, 0, 0);
  }
}

template <int FTW, int ET>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[FTW][ET]) {
#pragma unroll
  for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
    for (int et = 0; et < ET; ++et) acc[ft][et] = splat4(0.f);
}

// acc[ft][et] = bias[16*(ft0+ft) + 4q .. +4]   (bias may be nullptr -> zeros)
template <int FTW, int ET>
__device__ __forceinline__ void acc_bias(f32x4 (&acc)[FTW][ET], const float* __restrict__ bias, int ft0, int lane) {
  const int q = lane >> 4;
#pragma unroll
  for (int ft = 0; ft < FTW; ++ft) {
    f32x4 b = bias ? ldg4(bias + 16 * (ft0 + ft) + 4 * q) : splat4(0.f);
#pragma unroll
    for (int et = 0; et < ET; ++et) acc[ft][et] = b;
  }
}

// Store the accumulator tile to LDS X[row][colbase + 16*(ft0+ft) + 4q].
template <int FTW, int ET>
__device__ __forceinline__ void acc_to_lds(const f32x4 (&acc)[FTW][ET], float* X, int ldx, int colbase, int ft0, int lane) {
  const int c = lane & 15, q = lane >> 4;
#pragma unroll
  for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
    for (int et = 0; et < ET; ++et) sts4(X + (16 * et + c) * ldx + colbase + 16 * (ft0 + ft) + 4 * q, acc[ft][et]);
}

// ----------------------------------------------------------------------------------------------
// LayerNorm (biased variance, eps 1e-5, affine) + ReLU over NOUT = NW * FTW * 16 features that are
// spread over the NW consecutive waves wbase .. wbase+NW-1 of the workgroup (several such groups may normalise
// different feature sets side by side).  ALL 256 threads must call this (it contains one __syncthreads);
// waves outside any group pass active=false.
//   red, red2 : LDS scratch, 4*TE floats each.
// ----------------------------------------------------------------------------------------------
// Statistics are combined with the parallel-variance formula (Chan et al.): every wave computes the exact
// two-pass (mean_w, M2_w) of its own NF = FTW*16 features with lane shuffles only, the waves then merge
//   mean = avg_w mean_w ;  M2 = sum_w M2_w + NF * sum_w (mean_w - mean)^2
// through LDS with ONE barrier (robust like the two-pass form, one barrier fewer).
template <int FTW, int ET, int NW>
__device__ __forceinline__ void layernorm_relu(f32x4 (&z)[FTW][ET], const float* __restrict__ gamma,
                                               const float* __restrict__ beta, int ft0, float* red, float* red2,
                                               int wave, int lane, bool active, bool relu = true, int wbase = 0) {
  constexpr int TE = 16 * ET;
  constexpr float inv_nf = 1.0f / (float)(FTW * 16);
  constexpr float inv_n = 1.0f / (float)(NW * FTW * 16);
  const int c = lane & 15, q = lane >> 4;
  if (active) {
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float s = 0.f;
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft) s += (z[ft][et][0] + z[ft][et][1]) + (z[ft][et][2] + z[ft][et][3]);
      s = red_q(s);
      const float mw = s * inv_nf;
      float d2 = 0.f;
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = z[ft][et][r] - mw;
          d2 = fmaf(d, d, d2);
        }
      d2 = red_q(d2);
      if (q == 0) {
        red[wave * TE + 16 * et + c] = mw;
        red2[wave * TE + 16 * et + c] = d2;
      }
    }
  }
  // the affine parameters are fetched ahead of the barrier so their L2 latency overlaps the wait for the slowest wave
  f32x4 gm[FTW], bt[FTW];
  if (active) {
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) {
      gm[ft] = ldg4(gamma + 16 * (ft0 + ft) + 4 * q);
      bt[ft] = ldg4(beta + 16 * (ft0 + ft) + 4 * q);
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float mws[NW], msum = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        mws[w] = red[(wbase + w) * TE + 16 * et + c];
        msum += mws[w];
        m2 += red2[(wbase + w) * TE + 16 * et + c];
      }
      const float mean = msum * (1.0f / (float)NW);
      float dm = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) dm = fmaf(mws[w] - mean, mws[w] - mean, dm);
      const float var = (m2 + (float)(FTW * 16) * dm) * inv_n;
      const float rstd = 1.0f / sqrtf(var + MDX_LN_EPS);
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft) {
        f32x4 y = (z[ft][et] - splat4(mean)) * splat4(rstd) * gm[ft] + bt[ft];
        z[ft][et] = relu ? relu4(y) : y;
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Backward building blocks (bond-predictor guidance gradient).
//   ln_xhat      : x (pre-LN, registers) -> x_hat in place, rstd per row out.   2 barriers.
//   ln_relu_bwd  : g = dL/d relu(LN(x)) -> dL/dx in place, given x_hat/rstd.     1 barrier.
//                  y = x_hat*gamma + beta ; mask = y > 0 ; gh = g*mask*gamma ;
//                  dx = rstd * (gh - mean(gh) - x_hat * mean(gh * x_hat))
// ALL threads call; waves >= NW pass active=false.  red..red4: 4*TE floats each.
// ----------------------------------------------------------------------------------------------
template <int FTW, int ET, int NW>
__device__ __forceinline__ void ln_xhat(f32x4 (&x)[FTW][ET], float (&rstd)[ET], float* red, float* red2, int wave,
                                        int lane, bool active) {
  constexpr int TE = 16 * ET;
  constexpr float inv_n = 1.0f / (float)(NW * FTW * 16);
  const int c = lane & 15, q = lane >> 4;
  if (active) {
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float s = 0.f;
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft) s += (x[ft][et][0] + x[ft][et][1]) + (x[ft][et][2] + x[ft][et][3]);
      s = red_q(s);
      if (q == 0) red[wave * TE + 16 * et + c] = s;
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * TE + 16 * et + c];
      const float mean = s * inv_n;
      float d2 = 0.f;
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft) {
        x[ft][et] = x[ft][et] - splat4(mean);
#pragma unroll
        for (int r = 0; r < 4; ++r) d2 = fmaf(x[ft][et][r], x[ft][et][r], d2);
      }
      d2 = red_q(d2);
      if (q == 0) red2[wave * TE + 16 * et + c] = d2;
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red2[w * TE + 16 * et + c];
      rstd[et] = 1.0f / sqrtf(v * inv_n + MDX_LN_EPS);
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft) x[ft][et] = x[ft][et] * splat4(rstd[et]);
    }
  }
}

// y = relu(x_hat * gamma + beta)
template <int FTW, int ET>
__device__ __forceinline__ void ln_apply_relu(f32x4 (&y)[FTW][ET], const f32x4 (&xhat)[FTW][ET],
                                              const float* __restrict__ gamma, const float* __restrict__ beta, int ft0,
                                              int lane) {
  const int q = lane >> 4;
#pragma unroll
  for (int ft = 0; ft < FTW; ++ft) {
    const f32x4 gm = ldg4(gamma + 16 * (ft0 + ft) + 4 * q), bt = ldg4(beta + 16 * (ft0 + ft) + 4 * q);
#pragma unroll
    for (int et = 0; et < ET; ++et) y[ft][et] = relu4(xhat[ft][et] * gm + bt);
  }
}

template <int FTW, int ET, int NW>
__device__ __forceinline__ void ln_relu_bwd(f32x4 (&g)[FTW][ET], const f32x4 (&xhat)[FTW][ET], const float (&rstd)[ET],
                                            const float* __restrict__ gamma, const float* __restrict__ beta, int ft0,
                                            float* red3, float* red4, int wave, int lane, bool active) {
  constexpr int TE = 16 * ET;
  constexpr float inv_n = 1.0f / (float)(NW * FTW * 16);
  const int c = lane & 15, q = lane >> 4;
  if (active) {
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) {
      const f32x4 gm = ldg4(gamma + 16 * (ft0 + ft) + 4 * q), bt = ldg4(beta + 16 * (ft0 + ft) + 4 * q);
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        const f32x4 y = xhat[ft][et] * gm + bt;
#pragma unroll
        for (int r = 0; r < 4; ++r) g[ft][et][r] = (y[r] > 0.f) ? g[ft][et][r] * gm[r] : 0.f;
      }
    }
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1 += g[ft][et][r];
          s2 = fmaf(g[ft][et][r], xhat[ft][et][r], s2);
        }
      s1 = red_q(s1);
      s2 = red_q(s2);
      if (q == 0) {
        red3[wave * TE + 16 * et + c] = s1;
        red4[wave * TE + 16 * et + c] = s2;
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        s1 += red3[w * TE + 16 * et + c];
        s2 += red4[w * TE + 16 * et + c];
      }
      s1 *= inv_n;
      s2 *= inv_n;
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
        g[ft][et] = (g[ft][et] - splat4(s1) - xhat[ft][et] * splat4(s2)) * splat4(rstd[et]);
    }
  }
}

// Sum over NOUT = NW*FTW*16 features of  w2[f] * z[row][f]  ->  one scalar per row, result broadcast
// to every lane that holds row 16*et + c (used for the 256->1 and 32->1 heads of PosUpdate).
// ALL threads call (one __syncthreads).  red: LDS scratch 4*TE floats.
template <int FTW, int ET, int NW>
__device__ __forceinline__ void dot_rows(const f32x4 (&z)[FTW][ET], const float* __restrict__ w2, int ft0, float* red,
                                         int wave, int lane, bool active, float (&out)[ET]) {
  constexpr int TE = 16 * ET;
  const int c = lane & 15, q = lane >> 4;
  if (active) {
    f32x4 w[FTW];
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) w[ft] = ldg4(w2 + 16 * (ft0 + ft) + 4 * q);
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      float s = 0.f;
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int r = 0; r < 4; ++r) s = fmaf(w[ft][r], z[ft][et][r], s);
      s = red_q(s);
      if (q == 0) red[wave * TE + 16 * et + c] = s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int et = 0; et < ET; ++et) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w * TE + 16 * et + c];
    out[et] = s;
  }
}

// XCD-aware tile remap (MI355X: 8 XCDs, block b is dispatched to XCD b % 8): give every XCD a
// contiguous range of tiles so neighbouring tiles (same molecule -> same node rows) share an L2.
// Bijective for any ntiles.
__device__ __forceinline__ int xcd_remap(int bid, int ntiles) {
  const int q = ntiles >> 3, r = ntiles & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
