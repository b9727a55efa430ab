// Micro-benchmark of the split-precision matrix path ("Next round" 1): the row-owner 256 -> 256 (LayerNorm, ReLU)
// chain of tools/ubench_chain.hip with every fp32 operand split as x = hi + lo in float16 and three v_mfma_f32_16x16x32_f16 per
// k-group (Whi Xhi + Whi Xlo + Wlo Xhi, fp32 accumulation) instead of eight v_mfma_f32_16x16x4_f32.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_split ubench_split.hip
// Reports, per variant: speed (rows x layers per second, as "equivalent fp32 TFLOP/s" = 2*256*256 FLOP per row and layer) and
// the error of the chain's output against the same chain evaluated in float64 on the host, next to the exact-fp32 kernel's own.
//
// Layouts.  Accumulators (both paths): lane = 16 q + c holds Y[row c][feature 16 ft + 4 q + s], s = 0..3  (C/D map of every
// 16x16 MFMA).  f16 path: B operand of k-group g (32 k-values) for lane (q, c) = 8 halves = features 32 g + 4 q + {0..3} and
// 32 g + 16 + 4 q + {0..3} of row c -- i.e. the accumulators of feature tiles 2g and 2g+1, converted in place; the A operand
// (weights, packed on the host) carries the same k permutation.  Weights are pre-scaled by 2^SW on the host so that their low
// halves stay in float16's normal range; LayerNorm absorbs the scale (eps scaled by 2^(2 SW)).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ float red_q(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

struct WS {
  __amdgpu_buffer_rsrc_t r;
  unsigned off;
};
__device__ __forceinline__ f32x4 ws_frag(const WS& w, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w.r, w.off, byte_off, 0));
}

// ---------------------------------------------------------------- exact fp32 (the shipped scheme) --------------------------
template <int R, int DEPTH>
__device__ __forceinline__ void gemm_f32(f32x4 (&y)[16][R], const f32x4 (&x)[16][R], const WS& w, int mbase) {
  constexpr int KG = 16, NP = 8 * KG;
  f32x4 ring[DEPTH][2];
#pragma unroll
  for (int p = 0; p < DEPTH; ++p) {
    ring[p][0] = ws_frag(w, mbase + (2 * p) * 1024);
    ring[p][1] = ws_frag(w, mbase + (2 * p + 1) * 1024);
  }
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value, ftp = p / KG, g = p % KG;
    const f32x4 a0 = ring[p % DEPTH][0], a1 = ring[p % DEPTH][1];
    if constexpr (p + DEPTH < NP) {
      ring[p % DEPTH][0] = ws_frag(w, mbase + (2 * (p + DEPTH)) * 1024);
      ring[p % DEPTH][1] = ws_frag(w, mbase + (2 * (p + DEPTH) + 1) * 1024);
    }
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
}

template <int R>
__device__ __forceinline__ void ln_relu(f32x4 (&y)[16][R], const float* __restrict__ gamma, const float* __restrict__ beta, int q,
                                        float eps) {
  constexpr float inv_n = 1.0f / 256;
#pragma unroll
  for (int rt = 0; rt < R; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int ft = 0; ft < 16; ++ft) s += (y[ft][rt][0] + y[ft][rt][1]) + (y[ft][rt][2] + y[ft][rt][3]);
    const float mean = red_q(s) * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int ft = 0; ft < 16; ++ft)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = y[ft][rt][r] - mean;
        d2 = fmaf(d, d, d2);
      }
    const float rstd = 1.0f / sqrtf(red_q(d2) * inv_n + eps);
#pragma unroll
    for (int ft = 0; ft < 16; ++ft) {
      if (ft % 2 == 0) __builtin_amdgcn_sched_barrier(0);
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * ft + 4 * q);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * ft + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) y[ft][rt][r] = fmaxf((y[ft][rt][r] - mean) * rstd * gm[r] + bt[r], 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------- split float16 -------------------------------------------
// fp32 accumulators of feature tiles 2g, 2g+1 -> hi / lo operand of k-group g  (LS: scale of the lo half)
template <int R>
__device__ __forceinline__ void split_x(const f32x4 (&y)[16][R], h8 (&xh)[8][R], h8 (&xl)[8][R], float LS = 1.0f) {
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = y[2 * g + t / 4][rt][t % 4];
        const _Float16 h = (_Float16)v;
        xh[g][rt][t] = h;
        xl[g][rt][t] = (_Float16)((v - (float)h) * LS);
      }
}

// the shipped form: y += Whi Xhi; t += Whi Xlo' + Wlo' Xhi (lo' = lo 2^11); y += t 2^-11 per feature-tile pair
template <int R, int DEPTH, int TERMS = 3>
__device__ __forceinline__ void gemm_split2(f32x4 (&y)[16][R], const h8 (&xh)[8][R], const h8 (&xl)[8][R], const WS& w, int mbase) {
  constexpr int KG = 8, NP = 8 * KG * 2;  // half-steps of two fragments
  f32x4 ring[DEPTH][2];
#pragma unroll
  for (int p = 0; p < DEPTH; ++p) {
    ring[p][0] = ws_frag(w, mbase + (2 * p) * 1024);
    ring[p][1] = ws_frag(w, mbase + (2 * p + 1) * 1024);
  }
  f32x4 t0[R], t1[R], u0[R], u1[R];
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value, ftp = p / (2 * KG), g = (p / 2) % KG, h = p % 2;
    const h8 a0 = __builtin_bit_cast(h8, ring[p % DEPTH][0]), a1 = __builtin_bit_cast(h8, ring[p % DEPTH][1]);
    if constexpr (p + DEPTH < NP) {
      ring[p % DEPTH][0] = ws_frag(w, mbase + (2 * (p + DEPTH)) * 1024);
      ring[p % DEPTH][1] = ws_frag(w, mbase + (2 * (p + DEPTH) + 1) * 1024);
    }
    if constexpr (g == 0 && h == 0) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) t0[rt] = t1[rt] = u0[rt] = u1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (h == 0) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xh[g][rt], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xh[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xl[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xl[g][rt], t1[rt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xh[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xh[g][rt], t1[rt], 0, 0, 0);
      }
      if constexpr (TERMS == 4) {
#pragma unroll
        for (int rt = 0; rt < R; ++rt) {
          u0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xl[g][rt], u0[rt], 0, 0, 0);
          u1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xl[g][rt], u1[rt], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g == KG - 1 && h == 1) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        if constexpr (TERMS == 4) {
          t0[rt] = t0[rt] + u0[rt] * (1.0f / 2048.0f);
          t1[rt] = t1[rt] + u1[rt] * (1.0f / 2048.0f);
        }
        y[2 * ftp][rt] = y[2 * ftp][rt] + t0[rt] * (1.0f / 2048.0f);
        y[2 * ftp + 1][rt] = y[2 * ftp + 1][rt] + t1[rt] * (1.0f / 2048.0f);
      }
    }
  });
}

// ---- split24 (round 6): THREE float16 pieces per operand, x = hi + lo' 2^-11 + l3'' 2^-22 EXACTLY (11 + 11 + <= 2
// significand bits: every fp32 value is represented without loss, i.e. operands carry the reference's full 24 bits), and the six
// products whose weight is >= 2^-22: Whi Xhi | Whi Xlo' + Wlo' Xhi (x 2^-11) | Wlo' Xlo' + Whi X3'' + W3'' Xhi (x 2^-22); dropped:
// Wlo' X3'', W3'' Xlo' (2^-33) and W3'' X3'' (2^-44).  Three accumulator sets per feature-tile pair, the stream carries three
// half-steps (hi, lo', l3'') per (pair, k-group): 1.5x the bytes of the two-piece stream.
template <int R>
__device__ __forceinline__ void split_x3(const f32x4 (&y)[16][R], h8 (&xh)[8][R], h8 (&xl)[8][R], h8 (&x3)[8][R]) {
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int rt = 0; rt < R; ++rt)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = y[2 * g + t / 4][rt][t % 4];
        const _Float16 h = (_Float16)v;
        const float r1 = (v - (float)h) * 2048.0f;   // exact
        const _Float16 l = (_Float16)r1;
        xh[g][rt][t] = h;
        xl[g][rt][t] = l;
        x3[g][rt][t] = (_Float16)((r1 - (float)l) * 2048.0f);   // exact: <= 2 significant bits are left
      }
}

template <int R, int DEPTH>
__device__ __forceinline__ void gemm_split3(f32x4 (&y)[16][R], const h8 (&xh)[8][R], const h8 (&xl)[8][R], const h8 (&x3)[8][R], const WS& w,
                                            int mbase) {
  constexpr int KG = 8, NP = 8 * KG * 3;  // third-steps of two fragments: hi, lo', l3''
  f32x4 ring[DEPTH][2];
#pragma unroll
  for (int p = 0; p < DEPTH; ++p) {
    ring[p][0] = ws_frag(w, mbase + (2 * p) * 1024);
    ring[p][1] = ws_frag(w, mbase + (2 * p + 1) * 1024);
  }
  f32x4 t0[R], t1[R], u0[R], u1[R];
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value, ftp = p / (3 * KG), g = (p / 3) % KG, h = p % 3;
    const h8 a0 = __builtin_bit_cast(h8, ring[p % DEPTH][0]), a1 = __builtin_bit_cast(h8, ring[p % DEPTH][1]);
    if constexpr (p + DEPTH < NP) {
      ring[p % DEPTH][0] = ws_frag(w, mbase + (2 * (p + DEPTH)) * 1024);
      ring[p % DEPTH][1] = ws_frag(w, mbase + (2 * (p + DEPTH) + 1) * 1024);
    }
    if constexpr (g == 0 && h == 0) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) t0[rt] = t1[rt] = u0[rt] = u1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (h == 0) {          // Whi: all three operand pieces
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xh[g][rt], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xh[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xl[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xl[g][rt], t1[rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        u0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, x3[g][rt], u0[rt], 0, 0, 0);
        u1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, x3[g][rt], u1[rt], 0, 0, 0);
      }
    } else if constexpr (h == 1) {   // Wlo': hi and lo pieces
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xh[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xh[g][rt], t1[rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        u0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xl[g][rt], u0[rt], 0, 0, 0);
        u1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xl[g][rt], u1[rt], 0, 0, 0);
      }
    } else {                         // W3'': the hi piece
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        u0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xh[g][rt], u0[rt], 0, 0, 0);
        u1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xh[g][rt], u1[rt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g == KG - 1 && h == 2) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        t0[rt] = t0[rt] + u0[rt] * (1.0f / 2048.0f);
        t1[rt] = t1[rt] + u1[rt] * (1.0f / 2048.0f);
        y[2 * ftp][rt] = y[2 * ftp][rt] + t0[rt] * (1.0f / 2048.0f);
        y[2 * ftp + 1][rt] = y[2 * ftp + 1][rt] + t1[rt] * (1.0f / 2048.0f);
      }
    }
  });
}

// The shipped form with the operand conversion PIPELINED into the GEMM (round 5): k-group g + 1 is split into
// its hi / lo halves while the first feature-tile pair's MFMAs of group g issue (the float16 MFMA co-executes with VALU work of the
// same wave; the fp32 one does not).  Only the first of the eight passes over the k-groups can carry conversions -- every pass
// needs all eight groups -- so at most 1/8 of the GEMM's MFMAs have VALU work next to them.  MODE 0: conversions placed in the MFMA
// region, scheduling left to the compiler; MODE 1: forced interleave, one MFMA then VPM VALU instructions (sched_group_barrier).
template <int R, int DEPTH, int MODE, int VPM = 10>
__device__ __forceinline__ void gemm_split2_pipe(f32x4 (&y)[16][R], const f32x4 (&x)[16][R], const WS& w, int mbase) {
  constexpr int KG = 8, NP = 8 * KG * 2;
  h8 xh[8][R], xl[8][R];
  auto conv = [&](auto gc, int rt0, int rt1) {
    constexpr int g = decltype(gc)::value;
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      if (rt < rt0 || rt >= rt1) continue;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const float v = x[2 * g + t / 4][rt][t % 4];
        const _Float16 h = (_Float16)v;
        xh[g][rt][t] = h;
        xl[g][rt][t] = (_Float16)((v - (float)h) * 2048.0f);
      }
    }
  };
  f32x4 ring[DEPTH][2];
#pragma unroll
  for (int p = 0; p < DEPTH; ++p) {
    ring[p][0] = ws_frag(w, mbase + (2 * p) * 1024);
    ring[p][1] = ws_frag(w, mbase + (2 * p + 1) * 1024);
  }
  conv(std::integral_constant<int, 0>{}, 0, R);
  f32x4 t0[R], t1[R];
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value, ftp = p / (2 * KG), g = (p / 2) % KG, h = p % 2;
    const h8 a0 = __builtin_bit_cast(h8, ring[p % DEPTH][0]), a1 = __builtin_bit_cast(h8, ring[p % DEPTH][1]);
    if constexpr (p + DEPTH < NP) {
      ring[p % DEPTH][0] = ws_frag(w, mbase + (2 * (p + DEPTH)) * 1024);
      ring[p % DEPTH][1] = ws_frag(w, mbase + (2 * (p + DEPTH) + 1) * 1024);
    }
    if constexpr (g == 0 && h == 0) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) t0[rt] = t1[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
    constexpr bool carry = ftp == 0 && g + 1 < KG;
    if constexpr (carry) {  // group g + 1: the first half of the row tiles next to this half-step's MFMAs, the rest next to the other's
      constexpr int half = (R + 1) / 2;
      if constexpr (h == 0) conv(std::integral_constant<int, (g + 1 < KG ? g + 1 : 0)>{}, 0, half);
      else conv(std::integral_constant<int, (g + 1 < KG ? g + 1 : 0)>{}, half, R);
    }
    if constexpr (h == 0) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xh[g][rt], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xh[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xl[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xl[g][rt], t1[rt], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        t0[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, xh[g][rt], t0[rt], 0, 0, 0);
        t1[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, xh[g][rt], t1[rt], 0, 0, 0);
      }
    }
    if constexpr (carry && MODE == 1) {
      constexpr int nm = h == 0 ? 4 * R : 2 * R;
#pragma unroll
      for (int i = 0; i < nm; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);  // VPM VALU instructions of the conversion
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g == KG - 1 && h == 1) {
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        y[2 * ftp][rt] = y[2 * ftp][rt] + t0[rt] * (1.0f / 2048.0f);
        y[2 * ftp + 1][rt] = y[2 * ftp + 1][rt] + t1[rt] * (1.0f / 2048.0f);
      }
    }
  });
}

// step p = ftp * 8 + g carries four 1-KiB fragments: hi(2ftp, g), hi(2ftp+1, g), lo(2ftp, g), lo(2ftp+1, g)
// TERMS = 3: Whi Xhi + Whi Xlo + Wlo Xhi;  TERMS = 1: plain float16 (speed reference only)
template <int R, int DEPTH, int TERMS>
__device__ __forceinline__ void gemm_split(f32x4 (&y)[16][R], const h8 (&xh)[8][R], const h8 (&xl)[8][R], const WS& w, int mbase) {
  constexpr int KG = 8, NP = 8 * KG, NF = TERMS == 3 ? 4 : 2;
  f32x4 ring[DEPTH][NF];
#pragma unroll
  for (int p = 0; p < DEPTH; ++p)
#pragma unroll
    for (int j = 0; j < NF; ++j) ring[p][j] = ws_frag(w, mbase + (4 * p + j) * 1024);
  static_for<0, NP>([&](auto pc) {
    constexpr int p = decltype(pc)::value, ftp = p / KG, g = p % KG;
    f32x4 a[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) a[j] = ring[p % DEPTH][j];
    if constexpr (p + DEPTH < NP) {
#pragma unroll
      for (int j = 0; j < NF; ++j) ring[p % DEPTH][j] = ws_frag(w, mbase + (4 * (p + DEPTH) + j) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    const h8 h0 = __builtin_bit_cast(h8, a[0]), h1 = __builtin_bit_cast(h8, a[1]);
    // same-accumulator MFMAs are kept 2 R instructions apart
#pragma unroll
    for (int rt = 0; rt < R; ++rt) {
      y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xh[g][rt], y[2 * ftp][rt], 0, 0, 0);
      y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xh[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
    }
    if constexpr (TERMS == 3) {
      const h8 l0 = __builtin_bit_cast(h8, a[2]), l1 = __builtin_bit_cast(h8, a[3]);
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xl[g][rt], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xl[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
      }
#pragma unroll
      for (int rt = 0; rt < R; ++rt) {
        y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, xh[g][rt], y[2 * ftp][rt], 0, 0, 0);
        y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, xh[g][rt], y[2 * ftp + 1][rt], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// phase stamps of one wave (block 0, wave 0) in its second pass over the chain: [layer][0 start, 1 operand split, 2 GEMM, 3 LayerNorm]
__device__ unsigned long long g_stamp[64];
__device__ int g_desync = 0;
#define STAMP(i)                                                                             \
  do {                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    if (blockIdx.x == 0 && tid == 0 && r == 1 && l < 8) g_stamp[l * 4 + (i)] = clock64();    \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)

// KIND 6 / 7: the shipped split form with the operand conversion pipelined into the GEMM (compiler-scheduled / forced interleave);
// KIND 0: exact fp32;  3: split float16, lo halves unscaled, one accumulator (first cut);  4: split float16 as shipped
// (mdx_split.h: lo halves scaled by 2^11, cross terms in accumulators of their own);  1: plain float16 (one product)
template <int KIND, int R, int DEPTH, int WPS, bool LN>
__global__ __launch_bounds__(256, WPS) void chain_kernel(const float* __restrict__ X, const void* __restrict__ W, const float* __restrict__ gb,
                                                         float* __restrict__ out, int reps, int nl, int nmat, float eps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, c = lane & 15;
  const size_t row0 = ((size_t)blockIdx.x * 4 + wave) * 16 * R;
  f32x4 x[16][R];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int rt = 0; rt < R; ++rt) x[g][rt] = *reinterpret_cast<const f32x4*>(X + (row0 + 16 * rt + c) * 256 + 16 * g + 4 * q);
  const WS ws{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(W), 0, -1, 0x00020000), 16u * lane};
  // UB_DESYNC: every wave starts at its own matrix, so the four waves of a CU never stream the same weights at the same time (as in the
  // real kernels, where waves sit at different layers) -- what the L1's cross-wave hits are worth
  int m = g_desync ? __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 4 + wave) % nmat)) : 0;
#pragma unroll 1
  for (int r = 0; r < reps; ++r) {
#pragma unroll 1
    for (int l = 0; l < nl; ++l) {
      f32x4 y[16][R];
#pragma unroll
      for (int ft = 0; ft < 16; ++ft)
#pragma unroll
        for (int rt = 0; rt < R; ++rt) y[ft][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int mbase = __builtin_amdgcn_readfirstlane(m) * 262144;
      STAMP(0);
      if constexpr (KIND == 0) {
        gemm_f32<R, DEPTH>(y, x, ws, mbase);
      } else if constexpr (KIND == 6 || KIND == 7) {
        gemm_split2_pipe<R, DEPTH, KIND - 6>(y, x, ws, mbase);
      } else if constexpr (KIND == 8) {
        h8 xh[8][R], xl[8][R], x3[8][R];
        split_x3<R>(x, xh, xl, x3);
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
          for (int rt = 0; rt < R; ++rt) asm volatile("" : "+v"(xh[g][rt]), "+v"(xl[g][rt]), "+v"(x3[g][rt]));
        STAMP(1);
        gemm_split3<R, DEPTH>(y, xh, xl, x3, ws, mbase * 3 / 2);
      } else if constexpr (KIND == 4 || KIND == 5) {
        h8 xh[8][R], xl[8][R];
        split_x<R>(x, xh, xl, 2048.0f);
#pragma unroll
        for (int g = 0; g < 8; ++g)
#pragma unroll
          for (int rt = 0; rt < R; ++rt) asm volatile("" : "+v"(xh[g][rt]), "+v"(xl[g][rt]));
        STAMP(1);
        gemm_split2<R, DEPTH, KIND == 5 ? 4 : 3>(y, xh, xl, ws, mbase);
      } else {
        h8 xh[8][R], xl[8][R];
        split_x<R>(x, xh, xl);
        gemm_split<R, DEPTH, KIND>(y, xh, xl, ws, mbase);
      }
#pragma unroll
      for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int rt = 0; rt < R; ++rt) asm volatile("" : "+v"(y[g][rt]));
      STAMP(2);
      if (LN) {
        const float* gbp = gb + (size_t)m * 512;
        asm volatile("" : "+s"(gbp));
        ln_relu<R>(y, gbp, gbp + 256, q, eps);
      }
#pragma unroll
      for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int rt = 0; rt < R; ++rt) asm volatile("" : "+v"(y[g][rt]));
      STAMP(3);
#pragma unroll
      for (int g = 0; g < 16; ++g)
#pragma unroll
        for (int rt = 0; rt < R; ++rt) x[g][rt] = y[g][rt];
      m = (m + 1 == nmat) ? 0 : m + 1;
    }
  }
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int rt = 0; rt < R; ++rt) *reinterpret_cast<f32x4*>(out + (row0 + 16 * rt + c) * 256 + 16 * g + 4 * q) = x[g][rt];
}

// ---------------------------------------------------------------- host ----------------------------------------------------
static uint16_t f2h(float f) {  // round to nearest even, subnormals kept
  _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static float h2f(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

struct Problem {
  int nmat;
  std::vector<float> W;   // [nmat][256 out][256 in]
  std::vector<float> gb;  // [nmat][gamma 256 | beta 256]
};

static void pack_f32(const Problem& P, std::vector<float>& out) {
  out.assign((size_t)P.nmat * 65536, 0.f);
  for (int m = 0; m < P.nmat; ++m)
    for (int ftp = 0; ftp < 8; ++ftp)
      for (int g = 0; g < 16; ++g)
        for (int j = 0; j < 2; ++j)
          for (int lane = 0; lane < 64; ++lane)
            for (int s = 0; s < 4; ++s) {
              const int q = lane >> 4, c = lane & 15, f = 16 * (2 * ftp + j) + c, k = 16 * g + 4 * q + s;
              out[(size_t)m * 65536 + ((size_t)((ftp * 16 + g) * 2 + j) * 64 + lane) * 4 + s] = P.W[(size_t)m * 65536 + f * 256 + k];
            }
}
// float16 stream: per matrix 64 steps x 4 fragments x 64 lanes x 8 halves (256 KiB -- the same bytes as fp32)
static void pack_split(const Problem& P, int sw, std::vector<uint16_t>& out) {
  out.assign((size_t)P.nmat * 131072, 0);
  const float sc = ldexpf(1.0f, sw);
  for (int m = 0; m < P.nmat; ++m)
    for (int ftp = 0; ftp < 8; ++ftp)
      for (int g = 0; g < 8; ++g)
        for (int j = 0; j < 2; ++j)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 8; ++t) {
              const int q = lane >> 4, c = lane & 15, f = 16 * (2 * ftp + j) + c, k = 32 * g + 16 * (t / 4) + 4 * q + t % 4;
              const float w = P.W[(size_t)m * 65536 + f * 256 + k] * sc;
              const uint16_t hi = f2h(w), lo = f2h(w - h2f(hi));
              const size_t step = (size_t)m * 64 + ftp * 8 + g;
              out[((step * 4 + j) * 64 + lane) * 8 + t] = hi;
              out[((step * 4 + 2 + j) * 64 + lane) * 8 + t] = lo;
            }
}

// the shipped stream: half-step hs = (ftp*8 + g)*2 + h, h = 0 hi fragments of tiles (2ftp, 2ftp+1), h = 1 lo fragments scaled by 2^11
static void pack_split2(const Problem& P, std::vector<uint16_t>& out) {
  out.assign((size_t)P.nmat * 131072, 0);
  for (int m = 0; m < P.nmat; ++m)
    for (int ftp = 0; ftp < 8; ++ftp)
      for (int g = 0; g < 8; ++g)
        for (int j = 0; j < 2; ++j)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 8; ++t) {
              const int q = lane >> 4, c = lane & 15, f = 16 * (2 * ftp + j) + c, k = 32 * g + 16 * (t / 4) + 4 * q + t % 4;
              const float w = P.W[(size_t)m * 65536 + f * 256 + k];
              const uint16_t hi = f2h(w), lo = f2h((w - h2f(hi)) * 2048.0f);
              const size_t hs = ((size_t)m * 64 + ftp * 8 + g) * 2;
              out[((hs * 2 + j) * 64 + lane) * 8 + t] = hi;
              out[(((hs + 1) * 2 + j) * 64 + lane) * 8 + t] = lo;
            }
}

// the split24 stream: third-step ts = (ftp*8 + g)*3 + h, h = 0 hi, 1 lo' = lo 2^11, 2 l3'' = l3 2^22 (w = hi + lo + l3 exactly)
static void pack_split3(const Problem& P, std::vector<uint16_t>& out) {
  out.assign((size_t)P.nmat * 196608 + 8192, 0);
  size_t inexact = 0;
  for (int m = 0; m < P.nmat; ++m)
    for (int ftp = 0; ftp < 8; ++ftp)
      for (int g = 0; g < 8; ++g)
        for (int j = 0; j < 2; ++j)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 8; ++t) {
              const int q = lane >> 4, c = lane & 15, f = 16 * (2 * ftp + j) + c, k = 32 * g + 16 * (t / 4) + 4 * q + t % 4;
              const float w = P.W[(size_t)m * 65536 + f * 256 + k];
              const uint16_t hi = f2h(w);
              const float r1 = (w - h2f(hi)) * 2048.0f;
              const uint16_t lo = f2h(r1);
              const float r2 = (r1 - h2f(lo)) * 2048.0f;
              const uint16_t l3 = f2h(r2);
              if ((double)h2f(hi) + (double)h2f(lo) / 2048.0 + (double)h2f(l3) / 4194304.0 != (double)w) ++inexact;
              const size_t ts = ((size_t)m * 64 + ftp * 8 + g) * 3;
              out[((ts * 2 + j) * 64 + lane) * 8 + t] = hi;
              out[(((ts + 1) * 2 + j) * 64 + lane) * 8 + t] = lo;
              out[(((ts + 2) * 2 + j) * 64 + lane) * 8 + t] = l3;
            }
  printf("split24 pack: %zu of %zu weights not represented exactly by hi + lo + l3\n", inexact, (size_t)P.nmat * 65536);
}

static void ref_chain(const Problem& P, const float* x0, int nl, bool ln, double* y) {
  std::vector<double> x(x0, x0 + 256), t(256);
  for (int l = 0; l < nl; ++l) {
    const int m = l % P.nmat;
    for (int f = 0; f < 256; ++f) {
      double s = 0;
      for (int k = 0; k < 256; ++k) s += (double)P.W[(size_t)m * 65536 + f * 256 + k] * x[k];
      t[f] = s;
    }
    if (ln) {
      double mean = 0, var = 0;
      for (int f = 0; f < 256; ++f) mean += t[f];
      mean /= 256;
      for (int f = 0; f < 256; ++f) var += (t[f] - mean) * (t[f] - mean);
      var /= 256;
      const double rstd = 1.0 / sqrt(var + 1e-5);
      for (int f = 0; f < 256; ++f) {
        const double v = (t[f] - mean) * rstd * P.gb[(size_t)m * 512 + f] + P.gb[(size_t)m * 512 + 256 + f];
        t[f] = v > 0 ? v : 0;
      }
    }
    x = t;
  }
  for (int f = 0; f < 256; ++f) y[f] = x[f];
}

struct Dev {
  float *X, *gb, *out;
  void *Wf32, *Wsplit0, *WsplitS, *Wsplit2, *Wsplit3;
  float* Xsmall;
  size_t rows;
};

template <int KIND, int R, int DEPTH, int WPS, bool LN>
static void run(const char* name, const Dev& d, const void* W, const Problem& P, float eps, const std::vector<float>& hX,
                const std::vector<double>& ref, int nl_check, int nref, const float* Xacc = nullptr) {
  auto kern = chain_kernel<KIND, R, DEPTH, WPS, LN>;
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)kern);
  const size_t lds = WPS == 1 ? 100000 : 0;  // pad so that exactly one workgroup fits per CU
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // ---- accuracy: nl_check layers once, first nref rows against float64
  const int gcheck = (nref + 64 * R - 1) / (64 * R);
  hipLaunchKernelGGL(kern, dim3(gcheck), dim3(256), lds, 0, Xacc ? Xacc : d.X, W, d.gb, d.out, 1, nl_check, P.nmat, eps);
  std::vector<float> o((size_t)nref * 256);
  hipMemcpy(o.data(), d.out, o.size() * 4, hipMemcpyDeviceToHost);
  double emax = 0, esum = 0, scale = 0;
  for (size_t i = 0; i < o.size(); ++i) {
    const double e = fabs((double)o[i] - ref[i]);
    emax = e > emax ? e : emax;
    esum += e * e;
    scale = fabs(ref[i]) > scale ? fabs(ref[i]) : scale;
  }
  // ---- speed
  const int grid = 256 * WPS * 3, nl = 6, reps = 8 * (R == 1 ? 2 : 1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, d.X, W, d.gb, d.out, reps, nl, P.nmat, eps);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, d.X, W, d.gb, d.out, reps, nl, P.nmat, eps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  if (getenv("UB_STAMPS")) {
    unsigned long long st[64];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_stamp), sizeof(st));
    printf("    phase cycles (layers 1..4 of the second pass): ");
    for (int l = 1; l < 5; ++l)
      printf("[split %llu gemm %llu ln %llu next %llu] ", st[l * 4 + 1] - st[l * 4], st[l * 4 + 2] - st[l * 4 + 1], st[l * 4 + 3] - st[l * 4 + 2],
             st[(l + 1) * 4] - st[l * 4 + 3]);
    printf("\n");
  }
  const double flop = (double)grid * 4 * reps * nl * 2.0 * 256 * 256 * 16 * R;
  printf("%-34s R=%d depth=%d wps=%d LN=%d vgpr=%3d scratch=%d : %7.3f ms %7.1f eq.TFLOP/s (%.2fx of 157.3) | %d layers: max err %.3e rms %.3e (max |y| %.2f)\n",
         name, R, DEPTH, WPS, (int)LN, fa.numRegs, (int)fa.localSizeBytes, best, flop / best / 1e9, flop / best / 1e9 / 157.3, nl_check, emax,
         sqrt(esum / o.size()), scale);
  fflush(stdout);
}

int main(int argc, char** argv) {
  if (getenv("UB_DESYNC")) { int one = 1; hipMemcpyToSymbol(HIP_SYMBOL(g_desync), &one, sizeof(one)); }
  const int nmat = 6, nl_check = 6, nref = 128, sw = argc > 1 ? atoi(argv[1]) : 8;
  Problem P;
  P.nmat = nmat;
  P.W.resize((size_t)nmat * 65536);
  P.gb.resize((size_t)nmat * 512);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {  // uniform (-1, 1)
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  };
  for (auto& w : P.W) w = (float)(rnd() * 0.0625 * 1.7);  // nn.Linear's uniform(-1/sqrt(K), 1/sqrt(K)), a little wider
  for (int m = 0; m < nmat; ++m)
    for (int f = 0; f < 256; ++f) P.gb[(size_t)m * 512 + f] = (float)(1.0 + 0.3 * rnd()), P.gb[(size_t)m * 512 + 256 + f] = (float)(0.2 * rnd());
  Dev d{};
  d.rows = (size_t)256 * 2 * 3 * 4 * 32;
  std::vector<float> hX(d.rows * 256);
  for (auto& v : hX) v = (float)(rnd() * 2.0);
  std::vector<double> refLN((size_t)nref * 256), refNo((size_t)nref * 256);
  for (int r = 0; r < nref; ++r) {
    ref_chain(P, hX.data() + (size_t)r * 256, nl_check, true, refLN.data() + (size_t)r * 256);
    ref_chain(P, hX.data() + (size_t)r * 256, 2, false, refNo.data() + (size_t)r * 256);
  }
  std::vector<float> pf;
  pack_f32(P, pf);
  std::vector<uint16_t> ps0, psS, ps2;
  pack_split(P, 0, ps0);
  pack_split(P, sw, psS);
  pack_split2(P, ps2);
  std::vector<uint16_t> ps3;
  pack_split3(P, ps3);
  // small-magnitude activations (x 2^-7 ~ 0.01, no LayerNorm to renormalise them): where an UNSCALED float16 low half sits in the
  // subnormal range
  std::vector<float> hXs((size_t)nref * 256);
  for (size_t i = 0; i < hXs.size(); ++i) hXs[i] = hX[i] * (1.0f / 128.0f);
  std::vector<double> refSm((size_t)nref * 256);
  for (int r = 0; r < nref; ++r) ref_chain(P, hXs.data() + (size_t)r * 256, 2, false, refSm.data() + (size_t)r * 256);
  hipMalloc(&d.X, hX.size() * 4);
  hipMalloc(&d.out, hX.size() * 4);
  hipMalloc(&d.gb, P.gb.size() * 4);
  hipMalloc(&d.Wf32, pf.size() * 4);
  hipMalloc(&d.Wsplit0, ps0.size() * 2);
  hipMalloc(&d.WsplitS, psS.size() * 2);
  hipMalloc(&d.Wsplit2, ps2.size() * 2);
  hipMalloc(&d.Wsplit3, ps3.size() * 2);
  hipMemcpy(d.Wsplit3, ps3.data(), ps3.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&d.Xsmall, (size_t)4096 * 256 * 4);
  hipMemset(d.Xsmall, 0, (size_t)4096 * 256 * 4);
  hipMemcpy(d.Xsmall, hXs.data(), hXs.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d.Wsplit2, ps2.data(), ps2.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(d.X, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d.gb, P.gb.data(), P.gb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d.Wf32, pf.data(), pf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d.Wsplit0, ps0.data(), ps0.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(d.WsplitS, psS.data(), psS.size() * 2, hipMemcpyHostToDevice);
  const float eps = 1e-5f, epsS = 1e-5f * ldexpf(1.0f, 2 * sw);
  printf("weight scale 2^%d for the scaled split packs; reference = float64 chain of %d layers on %d rows\n", sw, nl_check, nref);
  if (!getenv("UB_PIPE_ONLY")) {
  run<0, 1, 4, 2, true>("exact fp32 16x16x4", d, d.Wf32, P, eps, hX, refLN, nl_check, nref);
  run<0, 2, 4, 1, true>("exact fp32 16x16x4", d, d.Wf32, P, eps, hX, refLN, nl_check, nref);
  run<3, 1, 4, 2, true>("split f16 x3 (unscaled W)", d, d.Wsplit0, P, eps, hX, refLN, nl_check, nref);
  run<3, 1, 4, 2, true>("split f16 x3 (W * 2^sw)", d, d.WsplitS, P, epsS, hX, refLN, nl_check, nref);
  run<3, 1, 2, 2, true>("split f16 x3 (W * 2^sw)", d, d.WsplitS, P, epsS, hX, refLN, nl_check, nref);
  run<3, 2, 2, 1, true>("split f16 x3 (W * 2^sw)", d, d.WsplitS, P, epsS, hX, refLN, nl_check, nref);
  run<3, 2, 4, 1, true>("split f16 x3 (W * 2^sw)", d, d.WsplitS, P, epsS, hX, refLN, nl_check, nref);
  }
  run<4, 1, 4, 2, true>("SHIPPED: lo*2^11, own acc, 3 terms", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  run<4, 2, 4, 1, true>("SHIPPED: lo*2^11, own acc, 3 terms", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  run<8, 1, 4, 2, true>("split24: 3 pieces, 6 terms", d, d.Wsplit3, P, eps, hX, refLN, nl_check, nref);
  run<8, 1, 6, 2, true>("split24: 3 pieces, 6 terms", d, d.Wsplit3, P, eps, hX, refLN, nl_check, nref);
  run<8, 2, 4, 1, true>("split24: 3 pieces, 6 terms", d, d.Wsplit3, P, eps, hX, refLN, nl_check, nref);
  run<8, 2, 6, 1, true>("split24: 3 pieces, 6 terms", d, d.Wsplit3, P, eps, hX, refLN, nl_check, nref);
  if (getenv("UB_SPLIT24_ONLY")) {
    run<0, 1, 4, 2, true>("exact fp32 16x16x4", d, d.Wf32, P, eps, hX, refLN, nl_check, nref);
    run<8, 1, 4, 2, false>("split24, no LN, 2 layers", d, d.Wsplit3, P, eps, hX, refNo, 2, nref);
    run<4, 1, 4, 2, false>("SHIPPED split 3 terms, no LN", d, d.Wsplit2, P, eps, hX, refNo, 2, nref);
    run<0, 1, 4, 2, false>("exact fp32, no LN, 2 layers", d, d.Wf32, P, eps, hX, refNo, 2, nref);
    run<8, 1, 4, 2, false>("split24, small x", d, d.Wsplit3, P, eps, hX, refSm, 2, nref, d.Xsmall);
    run<0, 1, 4, 2, false>("exact fp32, small x", d, d.Wf32, P, eps, hX, refSm, 2, nref, d.Xsmall);
    return 0;
  }
  run<6, 1, 4, 2, true>("pipelined conversion (compiler)", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  run<6, 2, 4, 1, true>("pipelined conversion (compiler)", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  run<7, 1, 4, 2, true>("pipelined conversion (interleave)", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  run<7, 2, 4, 1, true>("pipelined conversion (interleave)", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  if (getenv("UB_PIPE_ONLY")) return 0;
  run<5, 1, 4, 2, true>("variant: + Wlo Xlo (4 terms)", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  run<5, 2, 4, 1, true>("variant: + Wlo Xlo (4 terms)", d, d.Wsplit2, P, eps, hX, refLN, nl_check, nref);
  run<1, 1, 4, 2, true>("plain f16 x1 (speed ref)", d, d.WsplitS, P, epsS, hX, refLN, nl_check, nref);
  run<1, 2, 4, 1, true>("plain f16 x1 (speed ref)", d, d.WsplitS, P, epsS, hX, refLN, nl_check, nref);
  // without LayerNorm: two layers (the error of the bare products; no scale absorption -> unscaled pack)
  run<0, 1, 4, 2, false>("exact fp32, no LN, 2 layers", d, d.Wf32, P, eps, hX, refNo, 2, nref);
  run<3, 1, 4, 2, false>("split f16 x3 unscaled, no LN", d, d.Wsplit0, P, eps, hX, refNo, 2, nref);
  run<4, 1, 4, 2, false>("SHIPPED split 3 terms, no LN", d, d.Wsplit2, P, eps, hX, refNo, 2, nref);
  run<5, 1, 4, 2, false>("variant 4 terms, no LN", d, d.Wsplit2, P, eps, hX, refNo, 2, nref);
  printf("-- inputs scaled by 2^-7 (|x| ~ 0.01), two layers without LayerNorm: errors relative to max |y|\n");
  run<0, 1, 4, 2, false>("exact fp32, small x", d, d.Wf32, P, eps, hX, refSm, 2, nref, d.Xsmall);
  run<3, 1, 4, 2, false>("split unscaled lo, small x", d, d.Wsplit0, P, eps, hX, refSm, 2, nref, d.Xsmall);
  run<4, 1, 4, 2, false>("SHIPPED split 3 terms, small x", d, d.Wsplit2, P, eps, hX, refSm, 2, nref, d.Xsmall);
  run<5, 1, 4, 2, false>("variant 4 terms, small x", d, d.Wsplit2, P, eps, hX, refSm, 2, nref, d.Xsmall);
  return 0;
}
