// Split-precision build of the node kernel (gfx950) -- round 4, opt-in (ND_SPLIT in NodeArgs::flags); the exact kernel is
// mdx_node.hip.  Same stages, tile layout, fused reductions and outputs:
//   MID    = NodeBlock tail   x += out_transform(relu(LN(centroid_lin(x) + sum_row m)))   (models/graph.py:50-54,363)
//   POSMLP = PosUpdate.left/right_lin_edge(x_new)                                         (graph.py:387-388)
//   PRE    = node_net(x) and the hoisted per-node table NT of the next block              (graph.py:39,51-54; mdx_kernels.h)
// Only the matrix products differ.  The tile design keeps activations in LDS as fp32 X[row][feature]; a wave reads its B operand
// (one row, 8 k-values per lane) with the same two conflict-free 16-byte reads per 32 k the exact kernel issues for two 16-wide
// k-groups, splits it into float16 hi / lo halves in registers (lo scaled by 2^11, mdx_split.h) and multiplies it with split packs of
// the weights on v_mfma_f32_16x16x32_f16: acc += Whi Xhi, t += Whi Xlo + Wlo Xhi, acc += t 2^-11 at the end of the k loop.
// The ablation behind this: with one MFMA in thirty-two the exact node kernel takes 28 us per launch instead of 68 -- 59 % of it is
// matrix-pipe time, which this build cuts to 3/16.
#include "mdx_kernels.h"
#include "mdx_tile_split.h"
#include "mdx_node_common.h"

namespace {

__device__ __forceinline__ void mlp_small_s(const MlpW& w, const float* sW1, const float* sW2, const float* Hn, float* S, float* red,
                                            float* red2, float* out, int v0, int N, int wave, int lane) {
  // 256 -> 64 (LN, ReLU) -> 64 ; each wave owns one 16-feature tile
  const int c = lane & 15, q = lane >> 4;
  f32x4 t[1][NT_];
  acc_bias<1, NT_>(t, w.b1, wave, lane);
  gemm_tile_s<1, NT_, 256>(t, sW1, 4, wave, Hn, LD256, lane);
  layernorm_relu<1, NT_, 4>(t, w.g, w.be, wave, red, red2, wave, lane, true);
  acc_to_lds<1, NT_>(t, S, LD64, 0, wave, lane);
  __syncthreads();
  acc_bias<1, NT_>(t, w.b2, wave, lane);
  gemm_tile_s<1, NT_, 64>(t, sW2, 4, wave, S, LD64, lane);
#pragma unroll
  for (int et = 0; et < NT_; ++et) {
    const int v = v0 + 16 * et + c;
    if (v < N) stg4(out + (size_t)v * 64 + 16 * wave + 4 * q, t[0][et]);
  }
  __syncthreads();
}

__global__ __launch_bounds__(MDX_WG, 2) void node_s_kernel(const NodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hn = smem + OFF_HN;
  float* X = smem + OFF_X;
  float* S = smem + OFF_S;
  float* red = smem + OFF_RED;
  float* red2 = smem + OFF_RED2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int N = a.N;
  const int ntile = (N + TN - 1) / TN;
  if ((int)blockIdx.x >= ntile) {  // fused reduction, second role (see node_kernel)
    static_assert(TN * 16 <= MDX_WG, "one thread per (node, float4) of a 64-wide row");
    const int v = ((int)blockIdx.x - ntile) * TN + (tid >> 4), c4 = tid & 15;
    if ((tid >> 4) < TN && v < N) {
      stg4(a.SR + (size_t)v * 64 + 4 * c4, seg_sum<64>(a.PR, a.pbase, nullptr, v, c4));
      stg4(a.SL + (size_t)v * 64 + 4 * c4, seg_sum<64>(a.FL, a.col_ptr, a.col_eids, v, c4));
    }
    return;
  }
  const int v0 = blockIdx.x * TN;
  const int ft0 = 4 * wave;
  bool valid[NT_];
  int vi[NT_];
#pragma unroll
  for (int et = 0; et < NT_; ++et) {
    vi[et] = v0 + 16 * et + c;
    valid[et] = vi[et] < N;
    if (!valid[et]) vi[et] = N - 1;
  }

  if (a.flags & ND_MID) {
    f32x4 z[4][NT_];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NT_; ++et) {
        const int f = 16 * (ft0 + ft) + 4 * q;
        const f32x4 ag = a.P ? seg_sum<256>(a.P, a.pbase, nullptr, vi[et], 4 * (ft0 + ft) + q)
                             : ldg4(a.aggr + (size_t)vi[et] * MDX_ND + f);
        if (a.aggr_out && valid[et]) stg4(a.aggr_out + (size_t)vi[et] * MDX_ND + f, ag);
        z[ft][et] = ldg4(a.NTin + (size_t)vi[et] * MDX_NTW + MDX_NT_C + f) + ag;
      }
    layernorm_relu<4, NT_, 4>(z, a.wmid.lng, a.wmid.lnb, ft0, red, red2, wave, lane, true);
    acc_to_lds<4, NT_>(z, X, LD256, 0, ft0, lane);
    __syncthreads();
    acc_bias<4, NT_>(z, a.wmid.bout, ft0, lane);
    gemm_tile_s<4, NT_, 256>(z, a.smid.Wout, 16, ft0, X, LD256, lane);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NT_; ++et) {
        const int f = 16 * (ft0 + ft) + 4 * q;
        if (a.flags & ND_DELTA) {
          if (valid[et]) stg4(a.dHn + (size_t)vi[et] * MDX_ND + f, z[ft][et]);
        } else {
          z[ft][et] = z[ft][et] + ldg4(a.Hn + (size_t)vi[et] * MDX_ND + f);
          if (valid[et]) stg4(a.Hn + (size_t)vi[et] * MDX_ND + f, z[ft][et]);
        }
      }
    acc_to_lds<4, NT_>(z, Hn, LD256, 0, ft0, lane);
  } else {
    for (int i = tid; i < TN * 64; i += MDX_WG) {
      const int row = i >> 6, c4 = i & 63;
      const int v = v0 + row;
      sts4(Hn + row * LD256 + 4 * c4, v < N ? ldg4(a.Hn + (size_t)v * MDX_ND + 4 * c4) : splat4(0.f));
    }
  }
  __syncthreads();

  if (a.flags & ND_POSMLP) {
    mlp_small_s(a.wmid.left, a.smid.leftW1, a.smid.leftW2, Hn, S, red, red2, a.Lf, v0, N, wave, lane);
    mlp_small_s(a.wmid.right, a.smid.rightW1, a.smid.rightW2, Hn, S, red, red2, a.Rf, v0, N, wave, lane);
  }

  if (a.flags & ND_PRE) {
    {
      f32x4 t[4][NT_];
      acc_bias<4, NT_>(t, a.wpre.nn.b1, ft0, lane);
      gemm_tile_s<4, NT_, 256>(t, a.spre.nnW1, 16, ft0, Hn, LD256, lane);
      layernorm_relu<4, NT_, 4>(t, a.wpre.nn.g, a.wpre.nn.be, ft0, red, red2, wave, lane, true);
      acc_to_lds<4, NT_>(t, X, LD256, 0, ft0, lane);
      __syncthreads();
      acc_bias<4, NT_>(t, a.wpre.nn.b2, ft0, lane);
      gemm_tile_s<4, NT_, 256>(t, a.spre.nnW2, 16, ft0, X, LD256, lane);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < NT_; ++et)
          if (valid[et]) stg4(a.H + (size_t)vi[et] * MDX_ND + 16 * (ft0 + ft) + 4 * q, t[ft][et]);
    }
    // concatenated per-node table: 60 feature tiles, 15 per wave in 3 chunks of 5
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
      const int f0 = 15 * wave + 5 * j;
      f32x4 t[5][NT_];
      acc_bias<5, NT_>(t, a.wpre.bcat, f0, lane);
      gemm_tile_s<5, NT_, 256, 2>(t, a.spre.Wcat, MDX_NTW / 16, f0, Hn, LD256, lane);
#pragma unroll
      for (int ft = 0; ft < 5; ++ft)
#pragma unroll
        for (int et = 0; et < NT_; ++et)
          if (valid[et]) stg4(a.NT + (size_t)vi[et] * MDX_NTW + 16 * (f0 + ft) + 4 * q, t[ft][et]);
    }
  }
}

}  // namespace

void launch_node_s(const NodeArgs& a, hipStream_t s) {
  if (a.N <= 0) return;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)node_s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NODE_LDS_FLOATS * 4);
    attr = true;
  }
  const int ntile = (a.N + TN - 1) / TN;
  const int grid = ((a.flags & ND_MID) && a.P) ? 2 * ntile : ntile;
  hipLaunchKernelGGL(node_s_kernel, dim3(grid), dim3(MDX_WG), NODE_LDS_FLOATS * 4, s, a);
}
