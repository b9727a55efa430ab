// gemm_tile of the fused tile kernels (mdx_tile.h) on the split float16 matrix path (mdx_split.h): used by mdx_node_s.hip and the split
// node backward in mdx_bondpred.hip.
#pragma once
#include "mdx_split.h"

// acc[ft][et] += W[16 (ft0 + ft) .. + 16][0 .. K) X[16 et .. + 16][0 .. K)^T
//   Wp : dense split pack, 1-KiB fragment index (g * FT + ft) * 2 + h (g = k / 32; h = 0 hi, 1 lo scaled by 2^11), lane (q, c) holds
//        feature 16 ft + c, k = 32 g + 16 (t / 4) + 4 q + t % 4 (host: PackCtx::pack_dense_split) -- the k order of the two 16-byte
//        LDS reads below
//   X  : LDS tile, fp32, row-major, leading dimension ldx
template <int FTW, int ET, int K, int D = 3>
__device__ __forceinline__ void gemm_tile_s(f32x4 (&acc)[FTW][ET], const float* __restrict__ Wp, int FT, int ft0, const float* X,
                                            int ldx, int lane) {
  static_assert(K % 32 == 0, "K must be a multiple of 32");
  constexpr int G = K / 32;
  const int c = lane & 15, q = lane >> 4;
  const float* xb = X + c * ldx + 4 * q;
  const float* base = Wp + (size_t)__builtin_amdgcn_readfirstlane(ft0) * 512;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
  const unsigned off = 16u * lane;
  const int gs = __builtin_amdgcn_readfirstlane(FT) * 2048;  // bytes between consecutive k-groups
  f32x4 a[D][FTW][2], t[FTW][ET];
  auto load_a = [&](f32x4(&dst)[FTW][2], int g) {
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) {
      dst[ft][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, g * gs + ft * 2048, 0));
      dst[ft][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, g * gs + ft * 2048 + 1024, 0));
    }
  };
#pragma unroll
  for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
    for (int et = 0; et < ET; ++et) t[ft][et] = splat4(0.f);
  static_for<0, D - 1>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    if constexpr (g < G) load_a(a[g], g);
  });
  static_for<0, G>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    if constexpr (g + D - 1 < G) load_a(a[(g + D - 1) % D], g + D - 1);
    h8 xh[ET], xl[ET];
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      const f32x4 v0 = lds4(xb + et * 16 * ldx + g * 32), v1 = lds4(xb + et * 16 * ldx + g * 32 + 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = i < 4 ? v0[i] : v1[i - 4];
        const _Float16 h = (_Float16)v;
        xh[et][i] = h;
        xl[et][i] = (_Float16)((v - (float)h) * MDX_LO_UP);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) {
      const h8 whi = __builtin_bit_cast(h8, a[g % D][ft][0]);
#pragma unroll
      for (int et = 0; et < ET; ++et) acc[ft][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xh[et], acc[ft][et], 0, 0, 0);
    }
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) {
      const h8 whi = __builtin_bit_cast(h8, a[g % D][ft][0]);
#pragma unroll
      for (int et = 0; et < ET; ++et) t[ft][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi, xl[et], t[ft][et], 0, 0, 0);
    }
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft) {
      const h8 wlo = __builtin_bit_cast(h8, a[g % D][ft][1]);
#pragma unroll
      for (int et = 0; et < ET; ++et) t[ft][et] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo, xh[et], t[ft][et], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
#pragma unroll
  for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
    for (int et = 0; et < ET; ++et) acc[ft][et] = acc[ft][et] + t[ft][et] * splat4(MDX_LO_DOWN);
}

