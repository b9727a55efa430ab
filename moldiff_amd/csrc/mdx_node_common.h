// Shared by the exact (mdx_node.hip) and the split-precision (mdx_node_s.hip) node kernels: the LDS layout of a 16-node tile and the
// in-order segmented sum used by the fused reductions.
#pragma once
#include "mdx_kernels.h"
#include "mdx_tile.h"

namespace {

constexpr int NT_ = MDX_NT;
constexpr int TN = 16 * NT_;
constexpr int LD64 = mdx_ld(64);
constexpr int LD256 = mdx_ld(256);
constexpr int OFF_HN = 0;
constexpr int OFF_X = OFF_HN + TN * LD256;
constexpr int OFF_S = OFF_X + TN * LD256;
constexpr int OFF_RED = OFF_S + TN * LD64;
constexpr int OFF_RED2 = OFF_RED + 4 * TN;
constexpr int NODE_LDS_FLOATS = OFF_RED2 + 4 * TN;

// sum over the run ptr[v] .. ptr[v+1] of row i (or eids[i]) of src, this lane's four features: 4 independent loads in flight,
// summed in CSR order
template <int C>
__device__ __forceinline__ f32x4 seg_sum(const float* __restrict__ src, const int* __restrict__ ptr, const int* __restrict__ eids,
                                         int v, int c4) {
  const int j0 = ptr[v], j1 = ptr[v + 1];
  f32x4 s0 = splat4(0.f);
  int j = j0;
  for (; j + 4 <= j1; j += 4) {
    const int i0 = eids ? eids[j] : j, i1 = eids ? eids[j + 1] : j + 1, i2 = eids ? eids[j + 2] : j + 2,
              i3 = eids ? eids[j + 3] : j + 3;
    const f32x4 a0 = ldg4(src + (size_t)i0 * C + 4 * c4), a1 = ldg4(src + (size_t)i1 * C + 4 * c4),
                a2 = ldg4(src + (size_t)i2 * C + 4 * c4), a3 = ldg4(src + (size_t)i3 * C + 4 * c4);
    s0 = (((s0 + a0) + a1) + a2) + a3;
  }
  for (; j < j1; ++j) {
    const int i0 = eids ? eids[j] : j;
    s0 = s0 + ldg4(src + (size_t)i0 * C + 4 * c4);
  }
  return s0;
}

}  // namespace
