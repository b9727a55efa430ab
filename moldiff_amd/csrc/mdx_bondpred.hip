// Bond-predictor decoder and the hand-written backward pass of NodeEdgeNet(update_pos=False) with respect to
// the atom positions (gfx950).  Reference: models/bond_predictor.py:128-162 (forward) and the autograd call
// models/model.py:312-325 (delta = -scale * d/dpos of a scalar of the predictor's logits).
//
// Positions enter the predictor only through the distance features D = smear(|pos_l - pos_r|) that are
// concatenated to the edge state in EVERY block (models/graph.py:351-357), so dL/dpos = sum over blocks of
// dL/dD_i pushed through smear' and d|rel|/dpos.  That needs dL/dHe'_i in every block, i.e. a full data-gradient
// backward through both the node and the edge stream (no weight gradients).  Strategy: the forward keeps a small
// tape per block (He'_i (E,64) and the per-node tables) and the backward edge kernel RECOMPUTES the per-edge
// activations of its tile from that tape (first layers have K = 64 and are cheap) instead of storing (E,256)
// tensors; transposed weight packs make every dgrad a gemm_tile call.
#include "mdx_kernels.h"
#ifndef MDX_TILE_RING
#define MDX_TILE_RING 4  // weight groups in flight per wave (node_bwd 0.79 -> 0.73, edge_tail_bwd 0.69 -> 0.68 ms per guided step)
#endif
#include "mdx_tile.h"
#include "mdx_tile_split.h"

namespace {

constexpr int LD64 = mdx_ld(64);
constexpr int LD256 = mdx_ld(256);
constexpr int LD32 = mdx_ld(32);
constexpr int LD16 = mdx_ld(16);

// =================================================================================================
// Node backward.
//   TAIL (block i):  Hn_{i+1} = Hn_i + Wout relu(LN(z)) + b,  z = centroid(Hn_i) + aggr   -> GZ = dL/dz
//                    (written into the centroid columns of the gradient table GNT)
//   PRE  (block i):  gHn += Wcat^T GNT + node_net^T-backward(gH)
// =================================================================================================
constexpr int NBT = MDX_NT;
constexpr int NTN = 16 * NBT;
constexpr int N_A = 0;                        // [TN][264]
constexpr int N_B = N_A + NTN * LD256;        // [TN][264]
constexpr int N_RED = N_B + NTN * LD256;
constexpr int N_TOTAL = N_RED + 16 * NTN;

__device__ __forceinline__ void load_rows_ld(const float* __restrict__ src, int src_ld, int ncol, int v0, int N, float* dst,
                                             int ld, int tid) {
  const int c4n = ncol / 4;
  for (int i = tid; i < NTN * c4n; i += MDX_WG) {
    const int row = i / c4n, c4 = i - row * c4n;
    const int v = v0 + row;
    sts4(dst + row * ld + 4 * c4, v < N ? ldg4(src + (size_t)v * src_ld + 4 * c4) : splat4(0.f));
  }
}

__global__ __launch_bounds__(MDX_WG, 2) void node_bwd_kernel(const NodeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A = smem + N_A;
  float* B = smem + N_B;
  float* red = smem + N_RED;
  float *red2 = red + 4 * NTN, *red3 = red + 8 * NTN, *red4 = red + 12 * NTN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int v0 = blockIdx.x * NTN, N = a.N, ft0 = 4 * wave;
  bool valid[NBT];
  int vi[NBT];
#pragma unroll
  for (int et = 0; et < NBT; ++et) {
    vi[et] = v0 + 16 * et + c;
    valid[et] = vi[et] < N;
    if (!valid[et]) vi[et] = N - 1;
  }
  if (a.flags & NB_TAIL) {
    load_rows_ld(a.gHn, MDX_ND, MDX_ND, v0, N, A, LD256, tid);
    f32x4 z[4][NBT];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et) {
        const int f = 16 * (ft0 + ft) + 4 * q;
        z[ft][et] = ldg4(a.NTin + (size_t)vi[et] * MDX_NTW + MDX_NT_C + f) + ldg4(a.aggr + (size_t)vi[et] * MDX_ND + f);
      }
    float rstd[NBT];
    ln_xhat<4, NBT, 4>(z, rstd, red, red2, wave, lane, true);  // (barriers also cover the load of A)
    f32x4 g[4][NBT];
    acc_zero<4, NBT>(g);
    gemm_tile<4, NBT, 256>(g, a.wt.WoutT, 16, ft0, A, LD256, lane);
    ln_relu_bwd<4, NBT, 4>(g, z, rstd, a.w.lng, a.w.lnb, ft0, red3, red4, wave, lane, true);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et)
        if (valid[et]) stg4(a.GNT + (size_t)vi[et] * MDX_NTW + MDX_NT_C + 16 * (ft0 + ft) + 4 * q, g[ft][et]);
  }
  if (a.flags & NB_PRE) {
    f32x4 acc[4][NBT];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et) acc[ft][et] = ldg4(a.gHn + (size_t)vi[et] * MDX_ND + 16 * (ft0 + ft) + 4 * q);
    // Wcat^T GNT in 4 K-chunks
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
      const int kc = ch < 3 ? 256 : 192;
      __syncthreads();
      load_rows_ld(a.GNT + 256 * ch, MDX_NTW, kc, v0, N, A, LD256, tid);
      __syncthreads();
      if (ch < 3)
        gemm_tile<4, NBT, 256>(acc, a.wt.WcatT[ch], 16, ft0, A, LD256, lane);
      else
        gemm_tile<4, NBT, 192>(acc, a.wt.WcatT[ch], 16, ft0, A, LD256, lane);
    }
    __syncthreads();
    // node_net backward: H = W2 relu(LN(q)) + b2, q = W1 Hn + b1
    load_rows_ld(a.gH, MDX_ND, MDX_ND, v0, N, A, LD256, tid);
    load_rows_ld(a.Hn, MDX_ND, MDX_ND, v0, N, B, LD256, tid);
    __syncthreads();
    f32x4 gt[4][NBT], xh[4][NBT];
    acc_zero<4, NBT>(gt);
    gemm_tile<4, NBT, 256>(gt, a.wt.W2T, 16, ft0, A, LD256, lane);
    acc_bias<4, NBT>(xh, a.w.nn.b1, ft0, lane);
    gemm_tile<4, NBT, 256>(xh, a.w.nn.W1, 16, ft0, B, LD256, lane);
    float rstd[NBT];
    ln_xhat<4, NBT, 4>(xh, rstd, red, red2, wave, lane, true);
    ln_relu_bwd<4, NBT, 4>(gt, xh, rstd, a.w.nn.g, a.w.nn.be, ft0, red3, red4, wave, lane, true);
    __syncthreads();
    acc_to_lds<4, NBT>(gt, A, LD256, 0, ft0, lane);
    __syncthreads();
    gemm_tile<4, NBT, 256>(acc, a.wt.W1T, 16, ft0, A, LD256, lane);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et)
        if (valid[et]) stg4(a.gHn + (size_t)vi[et] * MDX_ND + 16 * (ft0 + ft) + 4 * q, acc[ft][et]);
  }
}

// the same kernel on the split float16 matrix path (gemm_tile_s, split packs of the transposed weights): NB_SPLIT launches
__global__ __launch_bounds__(MDX_WG, 2) void node_bwd_s_kernel(const NodeBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A = smem + N_A;
  float* B = smem + N_B;
  float* red = smem + N_RED;
  float *red2 = red + 4 * NTN, *red3 = red + 8 * NTN, *red4 = red + 12 * NTN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int v0 = blockIdx.x * NTN, N = a.N, ft0 = 4 * wave;
  bool valid[NBT];
  int vi[NBT];
#pragma unroll
  for (int et = 0; et < NBT; ++et) {
    vi[et] = v0 + 16 * et + c;
    valid[et] = vi[et] < N;
    if (!valid[et]) vi[et] = N - 1;
  }
  if (a.flags & NB_TAIL) {
    load_rows_ld(a.gHn, MDX_ND, MDX_ND, v0, N, A, LD256, tid);
    f32x4 z[4][NBT];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et) {
        const int f = 16 * (ft0 + ft) + 4 * q;
        z[ft][et] = ldg4(a.NTin + (size_t)vi[et] * MDX_NTW + MDX_NT_C + f) + ldg4(a.aggr + (size_t)vi[et] * MDX_ND + f);
      }
    float rstd[NBT];
    ln_xhat<4, NBT, 4>(z, rstd, red, red2, wave, lane, true);  // (barriers also cover the load of A)
    f32x4 g[4][NBT];
    acc_zero<4, NBT>(g);
    gemm_tile_s<4, NBT, 256>(g, a.wts.WoutT, 16, ft0, A, LD256, lane);
    ln_relu_bwd<4, NBT, 4>(g, z, rstd, a.w.lng, a.w.lnb, ft0, red3, red4, wave, lane, true);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et)
        if (valid[et]) stg4(a.GNT + (size_t)vi[et] * MDX_NTW + MDX_NT_C + 16 * (ft0 + ft) + 4 * q, g[ft][et]);
  }
  if (a.flags & NB_PRE) {
    f32x4 acc[4][NBT];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et) acc[ft][et] = ldg4(a.gHn + (size_t)vi[et] * MDX_ND + 16 * (ft0 + ft) + 4 * q);
    // Wcat^T GNT in 4 K-chunks
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
      const int kc = ch < 3 ? 256 : 192;
      __syncthreads();
      load_rows_ld(a.GNT + 256 * ch, MDX_NTW, kc, v0, N, A, LD256, tid);
      __syncthreads();
      if (ch < 3)
        gemm_tile_s<4, NBT, 256>(acc, a.wts.WcatT[ch], 16, ft0, A, LD256, lane);
      else
        gemm_tile_s<4, NBT, 192>(acc, a.wts.WcatT[ch], 16, ft0, A, LD256, lane);
    }
    __syncthreads();
    // node_net backward: H = W2 relu(LN(q)) + b2, q = W1 Hn + b1
    load_rows_ld(a.gH, MDX_ND, MDX_ND, v0, N, A, LD256, tid);
    load_rows_ld(a.Hn, MDX_ND, MDX_ND, v0, N, B, LD256, tid);
    __syncthreads();
    f32x4 gt[4][NBT], xh[4][NBT];
    acc_zero<4, NBT>(gt);
    gemm_tile_s<4, NBT, 256>(gt, a.wts.W2T, 16, ft0, A, LD256, lane);
    acc_bias<4, NBT>(xh, a.w.nn.b1, ft0, lane);
    gemm_tile_s<4, NBT, 256>(xh, a.ws.nnW1, 16, ft0, B, LD256, lane);
    float rstd[NBT];
    ln_xhat<4, NBT, 4>(xh, rstd, red, red2, wave, lane, true);
    ln_relu_bwd<4, NBT, 4>(gt, xh, rstd, a.w.nn.g, a.w.nn.be, ft0, red3, red4, wave, lane, true);
    __syncthreads();
    acc_to_lds<4, NBT>(gt, A, LD256, 0, ft0, lane);
    __syncthreads();
    gemm_tile_s<4, NBT, 256>(acc, a.wts.W1T, 16, ft0, A, LD256, lane);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < NBT; ++et)
        if (valid[et]) stg4(a.gHn + (size_t)vi[et] * MDX_ND + 16 * (ft0 + ft) + 4 * q, acc[ft][et]);
  }
}

// =================================================================================================
// Bond-predictor decoder: logits = MLP3([He[h]+He[Eh+h] | Hn[l]+Hn[r]])   (bond_predictor.py:155-160)
// =================================================================================================
constexpr int DET = 2;
constexpr int DTE = 16 * DET;
constexpr int D_A = 0;                       // [DTE][72]   edge part
constexpr int D_N = D_A + DTE * LD64;        // [DTE][264]  node part / gradient staging
constexpr int D_S = D_N + DTE * LD256;       // [DTE][72]
constexpr int D_S2 = D_S + DTE * LD64;       // [DTE][72]
constexpr int D_G = D_S2 + DTE * LD64;       // [DTE][24]   padded logits gradient
constexpr int D_RED = D_G + DTE * LD16;
constexpr int D_TOTAL = D_RED + 16 * DTE;

template <bool BWD>
__global__ __launch_bounds__(MDX_WG, 2) void bond_decode_kernel(const BondDecArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[D_TOTAL];
  float* A = smem + D_A;
  float* Nn = smem + D_N;
  float* S = smem + D_S;
  float* S2 = smem + D_S2;
  float* Gl = smem + D_G;
  float* red = smem + D_RED;
  float *red2 = red + 4 * DTE, *red3 = red + 8 * DTE, *red4 = red + 12 * DTE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int h0 = blockIdx.x * DTE, Eh = a.Eh;
  const int f1 = 16 * wave + 4 * q;
  for (int i = tid; i < DTE * 16; i += MDX_WG) {
    const int row = i >> 4, c4 = i & 15;
    const int h = h0 + row;
    f32x4 v = splat4(0.f);
    if (h < Eh) v = ldg4(a.He + (size_t)a.ref2int[h] * 64 + 4 * c4) + ldg4(a.He + (size_t)a.ref2int[Eh + h] * 64 + 4 * c4);
    sts4(A + row * LD64 + 4 * c4, v);
  }
  for (int i = tid; i < DTE * 64; i += MDX_WG) {
    const int row = i >> 6, c4 = i & 63;
    const int h = h0 + row;
    f32x4 v = splat4(0.f);
    if (h < Eh) {
      const int e = a.ref2int[h];
      v = ldg4(a.Hn + (size_t)a.left[e] * MDX_ND + 4 * c4) + ldg4(a.Hn + (size_t)a.right[e] * MDX_ND + 4 * c4);
    }
    sts4(Nn + row * LD256 + 4 * c4, v);
  }
  if (BWD) {
    for (int i = tid; i < DTE * 16; i += MDX_WG) {
      const int row = i >> 4, k = i & 15;
      const int h = h0 + row;
      Gl[row * LD16 + k] = (h < Eh && k < a.Ke) ? a.glogits[(size_t)h * a.Ke + k] : 0.f;
    }
  }
  __syncthreads();
  f32x4 x1[1][DET], x2[1][DET];
  float rs1[DET], rs2[DET];
  acc_bias<1, DET>(x1, a.w.b1, wave, lane);
  gemm_tile<1, DET, 64>(x1, a.w.W1e, 4, wave, A, LD64, lane);
  gemm_tile<1, DET, 256>(x1, a.w.W1n, 4, wave, Nn, LD256, lane);
  ln_xhat<1, DET, 4>(x1, rs1, red, red2, wave, lane, true);
  {
    f32x4 y[1][DET];
    ln_apply_relu<1, DET>(y, x1, a.w.g1, a.w.be1, wave, lane);
    acc_to_lds<1, DET>(y, S, LD64, 0, wave, lane);
  }
  __syncthreads();
  acc_bias<1, DET>(x2, a.w.b2, wave, lane);
  gemm_tile<1, DET, 64>(x2, a.w.W2, 4, wave, S, LD64, lane);
  ln_xhat<1, DET, 4>(x2, rs2, red, red2, wave, lane, true);
  if (!BWD) {
    f32x4 y[1][DET];
    ln_apply_relu<1, DET>(y, x2, a.w.g2, a.w.be2, wave, lane);
    acc_to_lds<1, DET>(y, S2, LD64, 0, wave, lane);
    __syncthreads();
    if (wave == 0) {
      f32x4 o[1][DET];
      acc_bias<1, DET>(o, a.w.b3, 0, lane);
      gemm_tile<1, DET, 64>(o, a.w.W3, 1, 0, S2, LD64, lane);
#pragma unroll
      for (int et = 0; et < DET; ++et) {
        const int h = h0 + 16 * et + c;
        if (h < Eh)
          for (int r = 0; r < 4; ++r)
            if (4 * q + r < a.Ke) a.logits[(size_t)h * a.Ke + 4 * q + r] = o[0][et][r];
      }
    }
    return;
  }
  // ---- backward ----
  f32x4 g2[1][DET];
  acc_zero<1, DET>(g2);
  gemm_tile<1, DET, 16>(g2, a.w.W3T, 4, wave, Gl, LD16, lane);
  ln_relu_bwd<1, DET, 4>(g2, x2, rs2, a.w.g2, a.w.be2, wave, red3, red4, wave, lane, true);
  acc_to_lds<1, DET>(g2, S2, LD64, 0, wave, lane);
  __syncthreads();
  f32x4 g1[1][DET];
  acc_zero<1, DET>(g1);
  gemm_tile<1, DET, 64>(g1, a.w.W2T, 4, wave, S2, LD64, lane);
  ln_relu_bwd<1, DET, 4>(g1, x1, rs1, a.w.g1, a.w.be1, wave, red3, red4, wave, lane, true);
  __syncthreads();
  acc_to_lds<1, DET>(g1, S, LD64, 0, wave, lane);
  __syncthreads();
  {  // dL/d(edge part) -> both directed edges of the pair
    f32x4 ge[1][DET];
    acc_zero<1, DET>(ge);
    gemm_tile<1, DET, 64>(ge, a.w.W1eT, 4, wave, S, LD64, lane);
#pragma unroll
    for (int et = 0; et < DET; ++et) {
      const int h = h0 + 16 * et + c;
      if (h < Eh) {
        stg4(a.gHe + (size_t)a.ref2int[h] * 64 + f1, ge[0][et]);
        stg4(a.gHe + (size_t)a.ref2int[Eh + h] * 64 + f1, ge[0][et]);
      }
    }
  }
  {  // dL/d(node part) per half-edge (summed into the nodes by a CSR pass)
    f32x4 gn[4][DET];
    acc_zero<4, DET>(gn);
    gemm_tile<4, DET, 64>(gn, a.w.W1nT, 16, 4 * wave, S, LD64, lane);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < DET; ++et) {
        const int h = h0 + 16 * et + c;
        if (h < Eh) stg4(a.GBN + (size_t)h * MDX_ND + 16 * (4 * wave + ft) + 4 * q, gn[ft][et]);
      }
  }
}

// =================================================================================================
// segment sum with an output leading dimension (writes straight into a column block of the gradient table)
// =================================================================================================
template <int C>
__device__ __forceinline__ f32x4 seg_sum_ld(const float* __restrict__ src, const int* __restrict__ ptr, const int* __restrict__ eids,
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
  for (; j < j1; ++j) s0 = s0 + ldg4(src + (size_t)(eids ? eids[j] : j) * C + 4 * c4);
  return s0;
}

template <int C>
__global__ __launch_bounds__(MDX_WG) void seg_reduce_ld_kernel(const float* __restrict__ src, const int* __restrict__ ptr,
                                                               const int* __restrict__ eids, float* __restrict__ out,
                                                               int out_ld, int N) {
  constexpr int LPN = C / 4;
  constexpr int NPB = MDX_WG / LPN;
  const int v = blockIdx.x * NPB + threadIdx.x / LPN;
  const int c4 = threadIdx.x % LPN;
  if (v >= N) return;
  stg4(out + (size_t)v * out_ld + 4 * c4, seg_sum_ld<C>(src, ptr, eids, v, c4));
}

// The six payload reductions that follow the backward edge kernel of a block, in one launch (13 waves per four nodes: GH and GGX
// four waves each, the two BondFFN payloads two each, the two gate payloads half a wave each).  Since round 5 the four by-right
// payloads arrive as partial rows summed inside the edge kernel (a node's ~2.5 pieces are added here in order; the launch reads
// ~140 MB instead of 515 MB at 256 molecules), the two by-left ones per edge as before; and the two
// (E,64) reductions that follow the EdgeBlock-tail backward.  Same per-lane loop as seg_reduce_ld_kernel: same bits.
__global__ __launch_bounds__(832) void seg_reduce_bwd_block_kernel(const SegBwdArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int v0 = blockIdx.x * 4, N = a.N;
  if (wave < 8) {  // (parts,256) partial rows of the by-right sums: dL/dH rows and the gate's node part
    const int v = v0 + (wave & 3);
    if (v >= N) return;
    if (wave < 4)
      stg4(a.gH + (size_t)v * MDX_ND + 4 * lane, seg_sum_ld<256>(a.GH, a.pbase_r, nullptr, v, lane));
    else
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_GX + 4 * lane, seg_sum_ld<256>(a.GGX, a.pbase_r, nullptr, v, lane));
  } else if (wave < 12) {  // left FFN (E,128) per edge by left endpoint; right FFN: partial rows
    const int v = v0 + 2 * (wave & 1) + (lane >> 5), c4 = lane & 31;
    if (v >= N) return;
    if (wave < 10)
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_NLL + 4 * c4, seg_sum_ld<128>(a.GNL0, a.row_ptr, nullptr, v, c4));
    else
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_NLR + 4 * c4, seg_sum_ld<128>(a.GNL1, a.pbase_r, nullptr, v, c4));
  } else {  // (.,32)
    const int v = v0 + ((lane & 31) >> 3), c4 = lane & 7;
    if (v >= N) return;
    if (lane < 32)
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_GXL + 4 * c4, seg_sum_ld<32>(a.GGXS0, a.row_ptr, nullptr, v, c4));
    else
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_GXR + 4 * c4, seg_sum_ld<32>(a.GGXS1, a.pbase_r, nullptr, v, c4));
  }
}

__global__ __launch_bounds__(MDX_WG) void seg_reduce_tail_block_kernel(const float* __restrict__ GU, const int* __restrict__ row_ptr,
                                                                       const int* __restrict__ col_ptr,
                                                                       const int* __restrict__ col_eids, float* __restrict__ GNT,
                                                                       int N) {
  // 8 nodes per workgroup: threads 0..127 the by-left sums (-> node_ffn_left columns), 128..255 the by-right sums
  const int half = threadIdx.x >> 7, t = threadIdx.x & 127;
  const int v = blockIdx.x * 8 + (t >> 4), c4 = t & 15;
  if (v >= N) return;
  if (half == 0)
    stg4(GNT + (size_t)v * MDX_NTW + MDX_NT_NFL + 4 * c4, seg_sum_ld<64>(GU, row_ptr, nullptr, v, c4));
  else
    stg4(GNT + (size_t)v * MDX_NTW + MDX_NT_NFR + 4 * c4, seg_sum_ld<64>(GU, col_ptr, col_eids, v, c4));
}

__global__ void dist_force_kernel(const float* __restrict__ gdist, const float* __restrict__ pos, const int* __restrict__ l,
                                  const int* __restrict__ r, float* __restrict__ w, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const float dx = pos[3 * l[e] + 0] - pos[3 * r[e] + 0];
  const float dy = pos[3 * l[e] + 1] - pos[3 * r[e] + 1];
  const float dz = pos[3 * l[e] + 2] - pos[3 * r[e] + 2];
  const float d = sqrtf(dx * dx + dy * dy + dz * dz);
  const float g = gdist[e] / d;
  w[3 * (size_t)e + 0] = g * dx;
  w[3 * (size_t)e + 1] = g * dy;
  w[3 * (size_t)e + 2] = g * dz;
}

__global__ void pos_grad_kernel(const float* __restrict__ w, const int* __restrict__ row_ptr, const int* __restrict__ col_ptr,
                                const int* __restrict__ col_eids, float* __restrict__ gpos, float scale, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * N) return;
  const int v = i / 3, k = i - 3 * v;
  float sl = 0.f, sr = 0.f;
  for (int j = row_ptr[v]; j < row_ptr[v + 1]; ++j) sl += w[3 * (size_t)j + k];
  for (int j = col_ptr[v]; j < col_ptr[v + 1]; ++j) sr += w[3 * (size_t)col_eids[j] + k];
  gpos[i] = scale * (sl - sr);
}

}  // namespace

void launch_node_bwd(const NodeBwdArgs& a, hipStream_t s) {
  if (a.N <= 0) return;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)node_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, N_TOTAL * 4);
    (void)hipFuncSetAttribute((const void*)node_bwd_s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, N_TOTAL * 4);
    attr = true;
  }
  if (a.flags & NB_SPLIT)
    hipLaunchKernelGGL(node_bwd_s_kernel, dim3((a.N + NTN - 1) / NTN), dim3(MDX_WG), N_TOTAL * 4, s, a);
  else
    hipLaunchKernelGGL(node_bwd_kernel, dim3((a.N + NTN - 1) / NTN), dim3(MDX_WG), N_TOTAL * 4, s, a);
}

void launch_bond_decode(const BondDecArgs& a, bool backward, hipStream_t s) {
  if (a.Eh <= 0) return;
  const dim3 grid((a.Eh + DTE - 1) / DTE);
  if (backward)
    hipLaunchKernelGGL(bond_decode_kernel<true>, grid, dim3(MDX_WG), 0, s, a);
  else
    hipLaunchKernelGGL(bond_decode_kernel<false>, grid, dim3(MDX_WG), 0, s, a);
}

void launch_seg_reduce_ld(const float* src, const int* ptr, const int* eids, float* out, int out_ld, int N, int C,
                          hipStream_t s) {
  if (N <= 0) return;
#define MDX_SR(CC)                                                                                                    \
  case CC:                                                                                                            \
    hipLaunchKernelGGL(seg_reduce_ld_kernel<CC>, dim3((N + (MDX_WG / (CC / 4)) - 1) / (MDX_WG / (CC / 4))), dim3(MDX_WG), 0, \
                       s, src, ptr, eids, out, out_ld, N);                                                            \
    break;
  switch (C) {
    MDX_SR(32) MDX_SR(64) MDX_SR(128) MDX_SR(256)
    default: break;
  }
#undef MDX_SR
}

void launch_seg_reduce_bwd_block(const SegBwdArgs& a, hipStream_t s) {
  if (a.N > 0) hipLaunchKernelGGL(seg_reduce_bwd_block_kernel, dim3((a.N + 3) / 4), dim3(832), 0, s, a);
}
void launch_seg_reduce_tail_block(const float* GU, const int* row_ptr, const int* col_ptr, const int* col_eids, float* GNT, int N,
                                  hipStream_t s) {
  if (N > 0) hipLaunchKernelGGL(seg_reduce_tail_block_kernel, dim3((N + 7) / 8), dim3(MDX_WG), 0, s, GU, row_ptr, col_ptr, col_eids, GNT, N);
}

void launch_dist_to_pos(const float* gdist, const float* pos, const int* l, const int* r, const int* row_ptr,
                        const int* col_ptr, const int* col_eids, float* tmpE3, float* tmpN3, float* gpos, float scale, int N,
                        int E, float cutoff, hipStream_t s) {
  (void)tmpN3; (void)cutoff;
  if (E > 0) hipLaunchKernelGGL(dist_force_kernel, dim3((E + 255) / 256), dim3(256), 0, s, gdist, pos, l, r, tmpE3, E);
  if (N > 0)
    hipLaunchKernelGGL(pos_grad_kernel, dim3((3 * N + 255) / 256), dim3(256), 0, s, tmpE3, row_ptr, col_ptr, col_eids, gpos,
                       scale, N);
}
