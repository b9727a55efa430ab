// Per-node kernels + graph reductions + embed/decode (gfx950).
//
//   node kernel  : MID  = NodeBlock tail   x += out_transform(relu(LN(centroid_lin(x) + sum_row m)))   (graph.py:50-54,363)
//                  POSMLP = PosUpdate.left/right_lin_edge(x_new)                                      (graph.py:387-388)
//                  PRE  = everything the next block computes per node: node_net(x) and the hoisted
//                         Linear(x) terms of the gates / BondFFNs / node_ffn (table NT, see mdx_kernels.h)
//   seg_reduce   : deterministic segmented sum over a CSR (the replacement for torch_scatter.scatter_sum,
//                  call sites graph.py:50,279,283,394): one wave-slice per node, sequential in CSR order.
//   embed/decode : MolDiff.forward / BondPredictor.forward ends (model.py:210-213,225-228; bond_predictor.py:135-140)
#include "mdx_kernels.h"
#ifndef MDX_TILE_RING
#define MDX_TILE_RING 4  // weight groups in flight per wave (node kernel at 256 molecules: 0.495 -> 0.444 ms per step against the two-stage
                         // loop; cutting the per-tile chain over three workgroups or 32-node tiles on top of it: no further gain)
#endif
#include "mdx_tile.h"
#include "mdx_node_common.h"

namespace {

__device__ __forceinline__ void mlp_small(const MlpW& w, const float* Hn, float* S, float* red, float* red2, float* out,
                                          int v0, int N, int wave, int lane) {
  // 256 -> 64 (LN, ReLU) -> 64 ; each wave owns one 16-feature tile
  const int c = lane & 15, q = lane >> 4;
  f32x4 t[1][NT_];
  acc_bias<1, NT_>(t, w.b1, wave, lane);
  gemm_tile<1, NT_, 256>(t, w.W1, 4, wave, Hn, LD256, lane);
  layernorm_relu<1, NT_, 4>(t, w.g, w.be, wave, red, red2, wave, lane, true);
  acc_to_lds<1, NT_>(t, S, LD64, 0, wave, lane);
  __syncthreads();
  acc_bias<1, NT_>(t, w.b2, wave, lane);
  gemm_tile<1, NT_, 64>(t, w.W2, 4, wave, S, LD64, lane);
#pragma unroll
  for (int et = 0; et < NT_; ++et) {
    const int v = v0 + 16 * et + c;
    if (v < N) stg4(out + (size_t)v * 64 + 16 * wave + 4 * q, t[0][et]);
  }
  __syncthreads();
}

__global__ __launch_bounds__(MDX_WG, 2) void node_kernel(const NodeArgs a) {
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
  if ((int)blockIdx.x >= ntile) {
    // Fused reduction, second role: the workgroups past the tile range compute the BondFFN sums edge kernel B needs (SR from the
    // partial rows of PR, SL through the by-right index list over FL), one (node, 4 features) per thread.  393 tiles leave 119 of the
    // chip's 512 workgroup slots free, so these run beside the tiles' GEMM chains instead of in front of them (as a prologue of every
    // tile they cost 10 us per launch).
    const int v = ((int)blockIdx.x - ntile) * (MDX_WG / 16) + (tid >> 4), c4 = tid & 15;   // 16 nodes per workgroup, one thread per (node, float4)
    if (v < N) {
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
        const f32x4 ag = a.P ? seg_sum<256>(a.P, a.pbase, nullptr, vi[et], 4 * (ft0 + ft) + q)   // the node's partial rows, in order
                             : ldg4(a.aggr + (size_t)vi[et] * MDX_ND + f);
        if (a.aggr_out && valid[et]) stg4(a.aggr_out + (size_t)vi[et] * MDX_ND + f, ag);
        z[ft][et] = ldg4(a.NTin + (size_t)vi[et] * MDX_NTW + MDX_NT_C + f) + ag;
      }
    layernorm_relu<4, NT_, 4>(z, a.wmid.lng, a.wmid.lnb, ft0, red, red2, wave, lane, true);
    acc_to_lds<4, NT_>(z, X, LD256, 0, ft0, lane);
    __syncthreads();
    acc_bias<4, NT_>(z, a.wmid.bout, ft0, lane);
    gemm_tile<4, NT_, 256>(z, a.wmid.Wout, 16, ft0, X, LD256, lane);
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
    mlp_small(a.wmid.left, Hn, S, red, red2, a.Lf, v0, N, wave, lane);
    mlp_small(a.wmid.right, Hn, S, red, red2, a.Rf, v0, N, wave, lane);
  }

  if (a.flags & ND_PRE) {
    {
      f32x4 t[4][NT_];
      acc_bias<4, NT_>(t, a.wpre.nn.b1, ft0, lane);
      gemm_tile<4, NT_, 256>(t, a.wpre.nn.W1, 16, ft0, Hn, LD256, lane);
      layernorm_relu<4, NT_, 4>(t, a.wpre.nn.g, a.wpre.nn.be, ft0, red, red2, wave, lane, true);
      acc_to_lds<4, NT_>(t, X, LD256, 0, ft0, lane);
      __syncthreads();
      acc_bias<4, NT_>(t, a.wpre.nn.b2, ft0, lane);
      gemm_tile<4, NT_, 256>(t, a.wpre.nn.W2, 16, ft0, X, LD256, lane);
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
      gemm_tile<5, NT_, 256>(t, a.wpre.Wcat, MDX_NTW / 16, f0, Hn, LD256, lane);
#pragma unroll
      for (int ft = 0; ft < 5; ++ft)
#pragma unroll
        for (int et = 0; et < NT_; ++et)
          if (valid[et]) stg4(a.NT + (size_t)vi[et] * MDX_NTW + 16 * (f0 + ft) + 4 * q, t[ft][et]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Segmented sum.  C = 256: one wave per node (64 lanes x float4);  C = 64: 16 lanes per node;
// C = 3: one lane per (node, component).  Sequential in CSR order => bit-reproducible.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(MDX_WG) void seg_reduce_kernel(const float* __restrict__ src, const int* __restrict__ ptr,
                                                            const int* __restrict__ eids, float* __restrict__ out,
                                                            const float* __restrict__ addend, int N) {
  constexpr int LPN = C / 4;              // lanes per node
  constexpr int NPB = MDX_WG / LPN;       // nodes per block
  const int v = blockIdx.x * NPB + threadIdx.x / LPN;
  const int c4 = threadIdx.x % LPN;
  if (v >= N) return;
  f32x4 s0 = seg_sum<C>(src, ptr, eids, v, c4);
  if (addend) s0 = ldg4(addend + (size_t)v * C + 4 * c4) + s0;
  stg4(out + (size_t)v * C + 4 * c4, s0);
}


// The same three sums after an EA_AGG edge kernel A (round 3): the kernel has already summed every node's run inside each 16-row
// unit, so aggr[v] / SR[v] are the in-order sums of v's partial rows pbase[v] .. pbase[v+1] of P (256 wide) / PR (64 wide) -- at most
// ceil(run / 16) + 1 rows instead of the whole run of M.  SL is still the indexed sum over FL.  Eight waves serve eight nodes:
// waves 0-3 two nodes' P rows each (one after the other), waves 4-5 the PR rows of four nodes each (16 lanes per node), waves 6-7 the
// FL runs of four nodes each.
__global__ __launch_bounds__(512) void seg_reduce_block2_kernel(const float* __restrict__ P, const float* __restrict__ PR,
                                                                const float* __restrict__ FL, const int* __restrict__ pbase,
                                                                const int* __restrict__ col_ptr, const int* __restrict__ col_eids,
                                                                float* __restrict__ aggr, float* __restrict__ SL,
                                                                float* __restrict__ SR, int N) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int v0 = blockIdx.x * 8;
  if (wave < 4) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = v0 + 2 * wave + k;
      if (v < N) stg4(aggr + (size_t)v * 256 + 4 * lane, seg_sum<256>(P, pbase, nullptr, v, lane));
    }
  } else {
    const int v = v0 + 4 * (wave & 1) + (lane >> 4), c4 = lane & 15;
    if (v >= N) return;
    if (wave < 6)
      stg4(SR + (size_t)v * 64 + 4 * c4, seg_sum<64>(PR, pbase, nullptr, v, c4));
    else
      stg4(SL + (size_t)v * 64 + 4 * c4, seg_sum<64>(FL, col_ptr, col_eids, v, c4));
  }
}

__global__ __launch_bounds__(MDX_WG) void seg_reduce3_kernel(const float* __restrict__ src, const int* __restrict__ ptr,
                                                             const int* __restrict__ eids, float* __restrict__ out,
                                                             const float* __restrict__ addend, int N) {
  const int i = blockIdx.x * MDX_WG + threadIdx.x;
  if (i >= 3 * N) return;
  const int v = i / 3, k = i - 3 * v;
  // sequential in CSR order like before (same bits), but eight independent loads in flight per step: the loop is latency-bound
  // (a run of ~24 edges used to be 24 dependent L2 round trips = 11.5 us per launch, six launches per step)
  float s = 0.f;
  const int j1 = ptr[v + 1];
  int j = ptr[v];
  for (; j + 8 <= j1; j += 8) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = src[3 * (size_t)(eids ? eids[j + u] : j + u) + k];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u];
  }
  for (; j < j1; ++j) s += src[3 * (size_t)(eids ? eids[j] : j) + k];
  out[i] = addend ? addend[i] + s : s;
}

// ------------------------------------------------------------------------------------------------
// Embedding: h_node = [x_n W_n^T | smear(t)],  h_edge = [x_e W_e^T | smear(t)]  (internal edge order),
// plus the per-row time arrays t/T used by the gates.
// ------------------------------------------------------------------------------------------------
// Node rows: 64 threads per atom, four features each (one 16-byte store per thread); the embedder's weight sits transposed in LDS
// ([k][feature]); each feature is the same fmaf chain over k as the one-thread-per-element version, so the result has the same bits.
constexpr int EMB_NODE_REPS = 4;
__device__ __forceinline__ void embed_node_block(const EmbedArgs& a, int blk, float* WT) {
  const int tid = threadIdx.x;
  const int Kn = a.Kn;
  for (int i = tid; i < Kn * MDX_ND; i += MDX_WG) {
    const int k = i / MDX_ND, f = i - k * MDX_ND;
    WT[i] = f < a.nd_emb ? a.Wn[f * Kn + k] : 0.f;
  }
  __syncthreads();
  const int f0 = 4 * (tid & 63);
#pragma unroll 1
  for (int rep = 0; rep < EMB_NODE_REPS; ++rep) {   // 4 atoms per pass: the staged weight serves 16 atoms per workgroup
    const int v = (blk * EMB_NODE_REPS + rep) * (MDX_WG / 64) + (tid >> 6);
    if (v >= a.N) return;
    const int64_t t = a.zero_time ? 0 : a.t[a.node_graph[v]];
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = k < Kn ? a.xn[(size_t)v * Kn + k] : 0.f;
    f32x4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = f0 + j;
      if (f < a.nd_emb) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < Kn) s = fmaf(x[k], WT[k * MDX_ND + f], s);
        out[j] = s;
      } else {
        const int k = f - a.nd_emb;
        const float u = fminf(fmaxf((float)t, 0.f), (float)a.T) - a.toff[k];
        out[j] = expf(a.tcoef[k] * (u * u));
      }
    }
    stg4(a.Hn + (size_t)v * MDX_ND + f0, out);
    if (f0 == 0) a.tn[v] = (float)t / (float)a.T;
  }
}

// Edge rows: 16 threads per edge, four features each (one 16-byte store per thread; round 2's one-thread-per-element version
// spent 95 us per step on 40 MB of output).  The embedder's weight sits transposed in LDS ([k][feature]); each feature is the
// same fmaf chain over k as before, so the result has the same bits.
constexpr int EMB_KMAX = 16;  // Ke (MolDiff: 6) or 2 Kn (bond predictor: 16); the node embedder has Kn <= 8
__device__ __forceinline__ void embed_edge_block(const EmbedArgs& a, int blk, float* WT) {
  const int tid = threadIdx.x;
  const int K = a.xe ? a.Ke : 2 * a.Kn;
  for (int i = tid; i < K * MDX_ED; i += MDX_WG) {
    const int k = i / MDX_ED, f = i - k * MDX_ED;
    WT[i] = f < a.ed_emb ? a.We[f * K + k] : 0.f;
  }
  __syncthreads();
  const int e = blk * (MDX_WG / 16) + (tid >> 4), f0 = 4 * (tid & 15);
  if (e >= a.E) return;
  const int nl = a.l[e], nr = a.r[e];
  const int64_t t = a.zero_time ? 0 : a.t[a.node_graph[nl]];
  float x[EMB_KMAX];
  if (a.xe) {
    int ref = a.int2ref ? a.int2ref[e] : e;
    if (a.half_rows > 0 && ref >= a.half_rows) ref -= a.half_rows;  // both directions share the half-edge row
#pragma unroll
    for (int k = 0; k < EMB_KMAX; ++k) x[k] = k < K ? a.xe[(size_t)ref * a.Ke + k] : 0.f;
  } else {  // bond predictor: cat[x_n[left], x_n[right]]
#pragma unroll
    for (int k = 0; k < EMB_KMAX; ++k)
      x[k] = k < a.Kn ? a.xn[(size_t)nl * a.Kn + k] : k < K ? a.xn[(size_t)nr * a.Kn + (k - a.Kn)] : 0.f;
  }
  f32x4 out;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int f = f0 + j;
    if (f < a.ed_emb) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < EMB_KMAX; ++k)
        if (k < K) s = fmaf(x[k], WT[k * MDX_ED + f], s);
      out[j] = s;
    } else {
      const int k = f - a.ed_emb;
      const float u = fminf(fmaxf((float)t, 0.f), (float)a.T) - a.toff[k];
      out[j] = expf(a.tcoef[k] * (u * u));
    }
  }
  stg4(a.He + (size_t)e * MDX_ED + f0, out);
  if (f0 == 0) a.te[e] = (float)t / (float)a.T;
}

// one launch for both: the first nb_node workgroups embed atoms (4 per workgroup), the rest edges (16 per workgroup)
__global__ __launch_bounds__(MDX_WG) void embed_kernel(const EmbedArgs a, const int nb_node) {
  __shared__ float WT[8 * MDX_ND];   // >= EMB_KMAX * MDX_ED
  static_assert(8 * MDX_ND >= EMB_KMAX * MDX_ED, "one staging buffer for both embedders");
  if ((int)blockIdx.x < nb_node)
    embed_node_block(a, blockIdx.x, WT);
  else
    embed_edge_block(a, blockIdx.x - nb_node, WT);
}

// ------------------------------------------------------------------------------------------------
// Decoders: node_decoder MLP(256 -> 256 -> Kn), edge_decoder MLP(64 -> 64 -> Ke) on He[h] + He[Eh + h].
// Second layers are zero-padded to 16 outputs on the host.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_node_body(const DecodeArgs& a, float* smem, const int block) {
  float* Hn = smem + OFF_HN;
  float* X = smem + OFF_X;
  float* red = smem + OFF_RED;
  float* red2 = smem + OFF_RED2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int v0 = block * TN, N = a.N, ft0 = 4 * wave;
  for (int i = tid; i < TN * 64; i += MDX_WG) {
    const int row = i >> 6, c4 = i & 63;
    const int v = v0 + row;
    sts4(Hn + row * LD256 + 4 * c4, v < N ? ldg4(a.Hn + (size_t)v * MDX_ND + 4 * c4) : splat4(0.f));
  }
  __syncthreads();
  f32x4 t[4][NT_];
  acc_bias<4, NT_>(t, a.nodedec.b1, ft0, lane);
  gemm_tile<4, NT_, 256>(t, a.nodedec.W1, 16, ft0, Hn, LD256, lane);
  layernorm_relu<4, NT_, 4>(t, a.nodedec.g, a.nodedec.be, ft0, red, red2, wave, lane, true);
  acc_to_lds<4, NT_>(t, X, LD256, 0, ft0, lane);
  __syncthreads();
  if (wave == 0) {
    f32x4 o[1][NT_];
    acc_bias<1, NT_>(o, a.nodedec.b2, 0, lane);
    gemm_tile<1, NT_, 256>(o, a.nodedec.W2, 1, 0, X, LD256, lane);
#pragma unroll
    for (int et = 0; et < NT_; ++et) {
      const int v = v0 + 16 * et + c;
      if (v < N)
        for (int r = 0; r < 4; ++r)
          if (4 * q + r < a.Kn) a.pred_node[(size_t)v * a.Kn + 4 * q + r] = o[0][et][r];
    }
  }
}

constexpr int DET = MDX_ET;
constexpr int DTE = 16 * DET;
constexpr int DOFF_A = 0;
constexpr int DOFF_B = DOFF_A + DTE * LD64;
constexpr int DOFF_RED = DOFF_B + DTE * LD64;
constexpr int DEC_EDGE_LDS_FLOATS = DOFF_RED + 8 * DTE;

__device__ __forceinline__ void decode_edge_body(const DecodeArgs& a, float* smem, const int block) {
  float* A = smem + DOFF_A;
  float* B = smem + DOFF_B;
  float* red = smem + DOFF_RED;
  float* red2 = red + 4 * DTE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int h0 = block * DTE, Eh = a.Eh;
  for (int i = tid; i < DTE * 16; i += MDX_WG) {
    const int row = i >> 4, c4 = i & 15;
    const int h = h0 + row;
    f32x4 v = splat4(0.f);
    if (h < Eh)
      v = ldg4(a.He + (size_t)a.ref2int[h] * 64 + 4 * c4) + ldg4(a.He + (size_t)a.ref2int[Eh + h] * 64 + 4 * c4);
    sts4(A + row * LD64 + 4 * c4, v);
  }
  __syncthreads();
  f32x4 t[1][DET];
  acc_bias<1, DET>(t, a.edgedec.b1, wave, lane);
  gemm_tile<1, DET, 64>(t, a.edgedec.W1, 4, wave, A, LD64, lane);
  layernorm_relu<1, DET, 4>(t, a.edgedec.g, a.edgedec.be, wave, red, red2, wave, lane, true);
  acc_to_lds<1, DET>(t, B, LD64, 0, wave, lane);
  __syncthreads();
  if (wave == 0) {
    f32x4 o[1][DET];
    acc_bias<1, DET>(o, a.edgedec.b2, 0, lane);
    gemm_tile<1, DET, 64>(o, a.edgedec.W2, 1, 0, B, LD64, lane);
#pragma unroll
    for (int et = 0; et < DET; ++et) {
      const int h = h0 + 16 * et + c;
      if (h < Eh)
        for (int r = 0; r < 4; ++r)
          if (4 * q + r < a.Ke) a.pred_halfedge[(size_t)h * a.Ke + 4 * q + r] = o[0][et][r];
    }
  }
}

// Both decoders in ONE launch (round 3): the first nb_node workgroups decode atoms, the rest half-edges -- the two are independent and
// each too small to fill the chip (16 + 20 us one after the other).
constexpr int DEC_LDS_FLOATS = NODE_LDS_FLOATS > DEC_EDGE_LDS_FLOATS ? NODE_LDS_FLOATS : DEC_EDGE_LDS_FLOATS;
__global__ __launch_bounds__(MDX_WG, 2) void decode_kernel(const DecodeArgs a, const int nb_node) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < nb_node)
    decode_node_body(a, smem, blockIdx.x);
  else
    decode_edge_body(a, smem, blockIdx.x - nb_node);
}

}  // namespace

void launch_node(const NodeArgs& a, hipStream_t s) {
  if (a.N <= 0) return;
  if (a.flags & ND_SPLIT) return launch_node_s(a, s);
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)node_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NODE_LDS_FLOATS * 4);
    attr = true;
  }
  const int ntile = (a.N + TN - 1) / TN;
  // fused reduction: a second set of workgroups (ids >= ntile) computes the BondFFN sums beside the tiles
  const int grid = ((a.flags & ND_MID) && a.P) ? ntile + (a.N + MDX_WG / 16 - 1) / (MDX_WG / 16) : ntile;
  hipLaunchKernelGGL(node_kernel, dim3(grid), dim3(MDX_WG), NODE_LDS_FLOATS * 4, s, a);
}

void launch_seg_reduce(const float* src, const int* ptr, const int* eids, float* out, const float* addend, int N, int C,
                       hipStream_t s) {
  if (N <= 0) return;
  if (C == 256)
    hipLaunchKernelGGL(seg_reduce_kernel<256>, dim3((N + 3) / 4), dim3(MDX_WG), 0, s, src, ptr, eids, out, addend, N);
  else if (C == 64)
    hipLaunchKernelGGL(seg_reduce_kernel<64>, dim3((N + 15) / 16), dim3(MDX_WG), 0, s, src, ptr, eids, out, addend, N);
  else
    hipLaunchKernelGGL(seg_reduce3_kernel, dim3((3 * N + MDX_WG - 1) / MDX_WG), dim3(MDX_WG), 0, s, src, ptr, eids, out,
                       addend, N);
}

void launch_seg_reduce_block2(const float* P, const float* PR, const float* FL, const int* pbase, const int* col_ptr,
                              const int* col_eids, float* aggr, float* SL, float* SR, int N, hipStream_t s) {
  if (N <= 0) return;
  hipLaunchKernelGGL(seg_reduce_block2_kernel, dim3((N + 7) / 8), dim3(512), 0, s, P, PR, FL, pbase, col_ptr, col_eids, aggr, SL, SR,
                     N);
}

void launch_embed(const EmbedArgs& a, hipStream_t s) {
  const int apb = (MDX_WG / 64) * EMB_NODE_REPS;   // atoms per workgroup
  const int nbn = a.N > 0 ? (int)((a.N + apb - 1) / apb) : 0;
  const int nbe = a.E > 0 ? (int)((a.E + MDX_WG / 16 - 1) / (MDX_WG / 16)) : 0;  // 16 edges per workgroup
  if (nbn + nbe > 0) hipLaunchKernelGGL(embed_kernel, dim3(nbn + nbe), dim3(MDX_WG), 0, s, a, nbn);
}

void launch_decode(const DecodeArgs& a, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DEC_LDS_FLOATS * 4);
    attr = true;
  }
  const int nbn = (a.N > 0 && a.pred_node) ? (a.N + TN - 1) / TN : 0;
  const int nbe = (a.Eh > 0 && a.pred_halfedge) ? (a.Eh + DTE - 1) / DTE : 0;
  if (nbn + nbe > 0) hipLaunchKernelGGL(decode_kernel, dim3(nbn + nbe), dim3(MDX_WG), DEC_LDS_FLOATS * 4, s, a, nbn);
}
