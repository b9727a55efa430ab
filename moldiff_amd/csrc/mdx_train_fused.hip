// Row-owner FUSED training kernels (round 6): one forward and one backward launch for a whole BondFFN of the EdgeBlock
// (reference models/graph.py:122-141 inside :268-283) in the float16 autocast arithmetic of the training path -- the per-operator
// path of mdx_train.hip runs the same chain as 11 forward and ~28 backward launches with every intermediate making a round trip
// through HBM.
//
//   bf    = bond_linear(X)                          (E,64) -> (E,128), no bias
//   prod  = bf * NL[idx]                            NL = node_linear(h_node), hoisted to N rows by the caller
//   inter = W_i2 relu(LN(W_i1 prod + b_i1)) + b_i2  common.MLP 128 -> 128 -> 64
//   gate  = W_g2 relu(LN(W_g1e X + b_g1 + GN[idx] + t w_t)) + b_g2      common.MLP (64 + 256 + 1) -> 32 -> 64, node part hoisted
//   out   = inter * sigmoid(gate)
//
// Design (the float16 row-owner scheme of hgemm_nt_rows_kernel, chained): a persistent workgroup of 8 waves per CU keeps EVERY
// weight of the chain in LDS as float16 (80 KB forward, 99 KB backward); a wave owns 16-row tiles and needs nobody else: the first
// layers take their activation operand straight from global memory (8 consecutive k of one row per lane = one 16-byte load), and
// every later layer takes it from the previous layer's ACCUMULATORS: lane (q, c) of v_mfma_f32_16x16x32_f16's D holds features
// 16 ft + 4 q .. + 3 of row c, the B operand of k-step ks wants 8 k-values of row c per lane -- tiles 2 ks and 2 ks + 1, rounded to
// float16 (the value autocast's Linear would read anyway), with the k permutation  pos(k) = 32 (t / 2) + 8 q + 4 (t % 2) + s  for
// k = 16 t + 4 q + s  baked into the LDS copy of the weight.  No LDS tile, no barrier in the loop; LayerNorm is wave-local.
// Rounding points are those of the per-operator path (every Linear result, every product of two Linear results, the sigmoid).
// Only what the weight gradients and the backward need is written out (float16): prod, the two pre-/post-LayerNorm pairs, inter, gate, out.
//
// Backward: dL/d(scatter_sum(out)) is gathered by the output index in the kernel; the data-gradient chain runs in registers; what
// leaves are the row gradients the weight-gradient GEMMs contract (g_inter, g_gate, g_pre1, g_bf, g_gpre), dL/dX, the per-edge
// dL/dNL rows (summed per node by the caller's segment sum) and one partial row of LayerNorm-parameter gradients per workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/moldiff_hip.h"
#include "mdx_tile.h"

int mdx_set_error(int code, const char* msg);  // mdx_api.hip

namespace {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));

#ifndef MDX_BF_FWD_THREADS
#define MDX_BF_FWD_THREADS 512   // forward: 8 waves per CU (12 measured 100 us against 88 us: the third wave per SIMD thrashes the LDS weight reads)
#endif
constexpr int BF_THREADS = 512, BF_WAVES = 8;
constexpr int BF_FWD_THREADS = MDX_BF_FWD_THREADS, BF_FWD_WAVES = BF_FWD_THREADS / 64;
constexpr int KB = 64, KI = 128, KO = 64, KG = 32;   // bond width, inter width, output width, gate hidden width
constexpr int LDB = KB + 8, LDI = KI + 8, LDG = KG + 8, LDO = KO + 8;

__device__ __forceinline__ float rh(float a) { return (float)(_Float16)a; }
__device__ __forceinline__ f32x4 rh4(f32x4 v) {
  f32x4 r = {rh(v[0]), rh(v[1]), rh(v[2]), rh(v[3])};
  return r;
}
__device__ __forceinline__ uint2 pack4(f32x4 v) {   // four floats -> four float16 (RNE)
  const f16x4_t h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  return __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ f32x4 unpack4(uint2 u) {
  const f16x4_t h = __builtin_bit_cast(f16x4_t, u);
  f32x4 r = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
  return r;
}
__device__ __forceinline__ f16x8_t pair8(uint2 a, uint2 b) {   // B operand of one k-step from two packed feature tiles
  const uint4 u = {a.x, a.y, b.x, b.y};
  return __builtin_bit_cast(f16x8_t, u);
}
__device__ __forceinline__ f16x8_t lds8(const uint16_t* p) { return *reinterpret_cast<const f16x8_t*>(p); }
__device__ __forceinline__ f32x4 ldh4(const _Float16* p) { return unpack4(*reinterpret_cast<const uint2*>(p)); }
__device__ __forceinline__ void sth4(_Float16* p, uint2 v) { *reinterpret_cast<uint2*>(p) = v; }
__device__ __forceinline__ float sumq(float v) {   // sum over the four lanes c, c + 16, c + 32, c + 48 (every lane gets it)
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// position of natural input feature k in the LDS copy of a weight whose B operand is the previous layer's accumulators
__device__ __forceinline__ int kperm(int k) {
  const int t = k >> 4, q = (k & 15) >> 2, s = k & 3;
  return 32 * (t >> 1) + 8 * q + 4 * (t & 1) + s;
}
// W (N outputs x K inputs, fp32, row stride ld) -> dst [N][K + 8] float16.  PERM: accumulator-fed layer (see kperm).
// Two adjacent feature tiles of a 16-row tile as ONE 16-byte store per lane (round 6).  A lane (row c, q) holds columns 16 ft + 4 q .. + 3
// of every tile: 8 bytes in float16, so a tile's store instruction wrote 16 rows x 32 bytes -- and with them the forward of this
// kernel took 355 us against 148 us without its stores (0.7 GB: 2.3 x the time HBM needs for them).  v_permlane16_swap exchanges the
// chunks of lane rows q and q ^ 1: even rows end up with columns 4 q .. 4 q + 7 of tile `a`, odd rows with columns 4 (q - 1) .. + 7 of tile
// `b` -- one instruction then writes 16 rows x 64 contiguous bytes.  Every lane of the wave must call it (the store itself is predicated).
__device__ __forceinline__ void sth8_pair(_Float16* row_base, int g2, uint2 ha, uint2 hb, int q, bool ok) {
  const auto sx = __builtin_amdgcn_permlane16_swap(ha.x, hb.x, false, false);
  const auto sy = __builtin_amdgcn_permlane16_swap(ha.y, hb.y, false, false);
  const uint4 v = {sx[0], sy[0], sx[1], sy[1]};
  const int col = (q & 1) ? 32 * g2 + 16 + 4 * (q - 1) : 32 * g2 + 4 * q;
#ifdef MDX_TRAIN_NT
  typedef unsigned int nt_u32x4_t __attribute__((__vector_size__(4 * sizeof(unsigned int))));
  if (ok) __builtin_nontemporal_store(__builtin_bit_cast(nt_u32x4_t, v), reinterpret_cast<nt_u32x4_t*>(row_base + col));
#else
  if (ok) *reinterpret_cast<uint4*>(row_base + col) = v;
#endif
}

// FT (even) packed tiles of one row block: FT / 2 paired stores
template <int FT>
__device__ __forceinline__ void st_tiles(_Float16* row_base, const uint2 (&pk)[FT], int q, bool ok) {
#pragma unroll
  for (int g2 = 0; g2 < FT / 2; ++g2) sth8_pair(row_base, g2, pk[2 * g2], pk[2 * g2 + 1], q, ok);
}

template <int N, int K, bool PERM>
__device__ __forceinline__ void stage_w(uint16_t* dst, const float* __restrict__ W, int ld, int tid, int nt = BF_THREADS) {
  for (int i = tid; i < N * K; i += nt) {
    const int n = i / K, k = i % K;
    dst[n * (K + 8) + (PERM ? kperm(k) : k)] = __builtin_bit_cast(uint16_t, (_Float16)W[(size_t)n * ld + k]);
  }
}
// the TRANSPOSE of W (N x K): dst [K outputs][N + 8], for the data gradient  g_in[k] = sum_n g_out[n] W[n][k]
template <int N, int K, bool PERM>
__device__ __forceinline__ void stage_wt(uint16_t* dst, const float* __restrict__ W, int ld, int tid, int nt = BF_THREADS) {
  for (int i = tid; i < N * K; i += nt) {
    const int n = i / K, k = i % K;
    dst[k * (N + 8) + (PERM ? kperm(n) : n)] = __builtin_bit_cast(uint16_t, (_Float16)W[(size_t)n * ld + k]);
  }
}
template <int N>
__device__ __forceinline__ void stage_v(float* dst, const float* __restrict__ v, int stride, int tid, bool round_half = false, int nt = BF_THREADS) {
  for (int i = tid; i < N; i += nt) dst[i] = v ? (round_half ? rh(v[(size_t)i * stride]) : v[(size_t)i * stride]) : 0.f;
}

// y[ft] += W[16 ft + .][.] x   for FT output tiles over KS k-steps of 32; w = LDS weight + c * LD + 8 q; x[ks] the B operands
template <int FT, int KS, int LD>
__device__ __forceinline__ void mm(f32x4 (&y)[FT], const uint16_t* w, const f16x8_t (&x)[KS]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(lds8(w + 16 * ft * LD + 32 * ks), x[ks], y[ft], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);   // one k-step's weight fragments at a time (hoisting them all spills)
  }
}
template <int FT>
__device__ __forceinline__ void zero(f32x4 (&y)[FT]) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) y[ft] = splat4(0.f);
}

// LayerNorm statistics of a row held as FT tiles of four features per lane (two-pass, biased variance: nn.LayerNorm)
template <int FT>
__device__ __forceinline__ void ln_stats(const f32x4 (&v)[FT], float& mean, float& rstd) {
  constexpr float inv_n = 1.0f / (16 * FT);
  float s = 0.f;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) s += (v[ft][0] + v[ft][1]) + (v[ft][2] + v[ft][3]);
  mean = sumq(s) * inv_n;
  float d2 = 0.f;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float d = v[ft][r] - mean;
      d2 = fmaf(d, d, d2);
    }
  rstd = 1.0f / sqrtf(sumq(d2) * inv_n + MDX_LN_EPS);
}

struct FwdLds {   // offsets in uint16 units
  static constexpr int WB = 0, WI1 = WB + KI * LDB, WI2 = WI1 + KI * LDI, WG1 = WI2 + KO * LDI, WG2 = WG1 + KG * LDB,
                       END = WG2 + KO * LDG;
  // fp32 constants behind the weights: bi1 g1 be1 (128 each) bi2 (64) bg1 gg gbe wt (32 each) bg2 (64)
  static constexpr int C_BI1 = 0, C_G1 = 128, C_BE1 = 256, C_BI2 = 384, C_BG1 = 448, C_GG = 480, C_GBE = 512, C_WT = 544, C_BG2 = 576,
                       C_END = 640;
  static constexpr int BYTES = END * 2 + C_END * 4;
};

__global__ __launch_bounds__(BF_FWD_THREADS) void bondffn_fwd_kernel(const mdx_bondffn_args a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t bf_smem[];
  uint16_t* S = bf_smem;
  float* C = reinterpret_cast<float*>(bf_smem + FwdLds::END);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  stage_w<KI, KB, false>(S + FwdLds::WB, a.Wb, (int)a.ldwb, tid, BF_FWD_THREADS);
  stage_w<KI, KI, true>(S + FwdLds::WI1, a.Wi1, (int)a.ldwi1, tid, BF_FWD_THREADS);
  stage_w<KO, KI, true>(S + FwdLds::WI2, a.Wi2, (int)a.ldwi2, tid, BF_FWD_THREADS);
  stage_w<KG, KB, false>(S + FwdLds::WG1, a.Wg1, (int)a.ldwg1, tid, BF_FWD_THREADS);
  stage_w<KO, KG, true>(S + FwdLds::WG2, a.Wg2, (int)a.ldwg2, tid, BF_FWD_THREADS);
  stage_v<128>(C + FwdLds::C_BI1, a.bi1, 1, tid, false, BF_FWD_THREADS); stage_v<128>(C + FwdLds::C_G1, a.g1, 1, tid, false, BF_FWD_THREADS); stage_v<128>(C + FwdLds::C_BE1, a.be1, 1, tid, false, BF_FWD_THREADS);
  stage_v<64>(C + FwdLds::C_BI2, a.bi2, 1, tid, false, BF_FWD_THREADS); stage_v<32>(C + FwdLds::C_BG1, a.bg1, 1, tid, false, BF_FWD_THREADS); stage_v<32>(C + FwdLds::C_GG, a.gg, 1, tid, false, BF_FWD_THREADS);
  stage_v<32>(C + FwdLds::C_GBE, a.gbe, 1, tid, false, BF_FWD_THREADS); stage_v<32>(C + FwdLds::C_WT, a.Wt, (int)a.ldwt, tid, true, BF_FWD_THREADS);
  stage_v<64>(C + FwdLds::C_BG2, a.bg2, 1, tid, false, BF_FWD_THREADS);
  __syncthreads();   // the only barrier

  const int E = (int)a.E, ntiles = (E + 15) >> 4, nw = gridDim.x * BF_FWD_WAVES;
  const _Float16* X = reinterpret_cast<const _Float16*>(a.X);
  const _Float16* NL = reinterpret_cast<const _Float16*>(a.NL);
  const uint16_t* wb = S + FwdLds::WB + c * LDB + 8 * q;
  const uint16_t* wi1 = S + FwdLds::WI1 + c * LDI + 8 * q;
  const uint16_t* wi2 = S + FwdLds::WI2 + c * LDI + 8 * q;
  const uint16_t* wg1 = S + FwdLds::WG1 + c * LDB + 8 * q;
  const uint16_t* wg2 = S + FwdLds::WG2 + c * LDG + 8 * q;
  _Float16* o_prod = reinterpret_cast<_Float16*>(a.prod);
  _Float16* o_pre1 = reinterpret_cast<_Float16*>(a.pre1);
  _Float16* o_post1 = reinterpret_cast<_Float16*>(a.post1);
  _Float16* o_inter = reinterpret_cast<_Float16*>(a.inter);
  _Float16* o_gpre = reinterpret_cast<_Float16*>(a.gpre);
  _Float16* o_gpost = reinterpret_cast<_Float16*>(a.gpost);
  _Float16* o_gate = reinterpret_cast<_Float16*>(a.gate);
  _Float16* o_out = reinterpret_cast<_Float16*>(a.out);

  int tile = blockIdx.x * BF_FWD_WAVES + wave;
  uint4 xr[2], xn[2];
  int64_t ni = 0, nin = 0;
  float te = 0.f, ten = 0.f;
  auto load = [&](uint4 (&d)[2], int64_t& n_, float& t_, int t) {
    const int row = min(16 * t + c, E - 1);   // clamped: loads stay inside the matrices, stores are predicated
    const _Float16* p = X + (size_t)row * a.ldx + 8 * q;
    d[0] = *reinterpret_cast<const uint4*>(p);
    d[1] = *reinterpret_cast<const uint4*>(p + 32);
    n_ = a.idx[row];
    t_ = a.te[row];
  };
  if (tile < ntiles) load(xr, ni, te, tile);
#pragma unroll 1
  for (; tile < ntiles; tile += nw) {
    if (tile + nw < ntiles) load(xn, nin, ten, tile + nw);
    const int row = 16 * tile + c;
    const bool ok = row < E;
    const size_t r = (size_t)min(row, E - 1);
    const f16x8_t xb[2] = {__builtin_bit_cast(f16x8_t, xr[0]), __builtin_bit_cast(f16x8_t, xr[1])};
    // ---- bond_linear, product with the node row
    uint2 pk[8];
    {
      f32x4 nl[8];
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) nl[ft] = ldh4(NL + (size_t)ni * a.ldnl + 16 * ft + 4 * q);
      f32x4 y[8];
      zero<8>(y);
      mm<8, 2, LDB>(y, wb, xb);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        pk[ft] = pack4(rh4(y[ft]) * nl[ft]);
      }
      st_tiles<8>(o_prod + r * KI, pk, q, ok);
    }
    // ---- inter module: Linear -> LayerNorm -> ReLU -> Linear
    f32x4 inter[4];
    {
      const f16x8_t pb[4] = {pair8(pk[0], pk[1]), pair8(pk[2], pk[3]), pair8(pk[4], pk[5]), pair8(pk[6], pk[7])};
      f32x4 y[8];
      zero<8>(y);
      mm<8, 4, LDI>(y, wi1, pb);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        y[ft] = rh4(y[ft] + lds4(C + FwdLds::C_BI1 + 16 * ft + 4 * q));
        pk[ft] = pack4(y[ft]);
      }
      st_tiles<8>(o_pre1 + r * KI, pk, q, ok);
      float mean, rstd;
      ln_stats<8>(y, mean, rstd);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        const f32x4 v = relu4((y[ft] - splat4(mean)) * splat4(rstd) * lds4(C + FwdLds::C_G1 + 16 * ft + 4 * q) +
                              lds4(C + FwdLds::C_BE1 + 16 * ft + 4 * q));
        pk[ft] = pack4(v);
      }
      st_tiles<8>(o_post1 + r * KI, pk, q, ok);
      const f16x8_t hb[4] = {pair8(pk[0], pk[1]), pair8(pk[2], pk[3]), pair8(pk[4], pk[5]), pair8(pk[6], pk[7])};
      zero<4>(inter);
      mm<4, 4, LDI>(inter, wi2, hb);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        inter[ft] = rh4(inter[ft] + lds4(C + FwdLds::C_BI2 + 16 * ft + 4 * q));
      }
      {
        const uint2 pi[4] = {pack4(inter[0]), pack4(inter[1]), pack4(inter[2]), pack4(inter[3])};
        st_tiles<4>(o_inter + r * KO, pi, q, ok);
      }
    }
    // ---- gate: Linear([X | node | t]) -> LayerNorm -> ReLU -> Linear; the node part arrives hoisted (GN rows), the time column is w_t t
    {
      f32x4 y[2];
      zero<2>(y);
      mm<2, 2, LDB>(y, wg1, xb);
      const float th = rh(te);
      uint2 pg[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const f32x4 ad = ldg4(a.GN + (size_t)ni * a.ldgn + 16 * ft + 4 * q) + splat4(th) * lds4(C + FwdLds::C_WT + 16 * ft + 4 * q);
        y[ft] = rh4((y[ft] + lds4(C + FwdLds::C_BG1 + 16 * ft + 4 * q)) + ad);
        pg[ft] = pack4(y[ft]);
      }
      st_tiles<2>(o_gpre + r * KG, pg, q, ok);
      float mean, rstd;
      ln_stats<2>(y, mean, rstd);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const f32x4 v = relu4((y[ft] - splat4(mean)) * splat4(rstd) * lds4(C + FwdLds::C_GG + 16 * ft + 4 * q) +
                              lds4(C + FwdLds::C_GBE + 16 * ft + 4 * q));
        pg[ft] = pack4(v);
      }
      st_tiles<2>(o_gpost + r * KG, pg, q, ok);
      const f16x8_t gb[1] = {pair8(pg[0], pg[1])};
      f32x4 g2[4];
      zero<4>(g2);
      mm<4, 1, LDG>(g2, wg2, gb);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        g2[ft] = rh4(g2[ft] + lds4(C + FwdLds::C_BG2 + 16 * ft + 4 * q));
      }
      {
        const uint2 pgt[4] = {pack4(g2[0]), pack4(g2[1]), pack4(g2[2]), pack4(g2[3])};
        st_tiles<4>(o_gate + r * KO, pgt, q, ok);
        const uint2 po[4] = {pack4(inter[0] * rh4(sigmoid4(g2[0]))), pack4(inter[1] * rh4(sigmoid4(g2[1]))),
                             pack4(inter[2] * rh4(sigmoid4(g2[2]))), pack4(inter[3] * rh4(sigmoid4(g2[3])))};
        st_tiles<4>(o_out + r * KO, po, q, ok);
      }
    }
    xr[0] = xn[0]; xr[1] = xn[1]; ni = nin; te = ten;
  }
}

struct BwdLds {
  static constexpr int WI2T = 0, WI1T = WI2T + KI * LDO, WB = WI1T + KI * LDI, WBT = WB + KI * LDB, WG2T = WBT + KB * LDI,
                       WG1T = WG2T + KG * LDO, END = WG1T + KB * LDG;
  // fp32 constants: g1 be1 (128 each) gg gbe (32 each)
  static constexpr int C_G1 = 0, C_BE1 = 128, C_GG = 256, C_GBE = 288, C_END = 320;
  static constexpr int BYTES = END * 2 + C_END * 4;
};
constexpr int BF_LNP = 320;   // floats per partial row of LayerNorm-parameter gradients: dg1 | dbe1 (128 each) | dgg | dgbe (32 each)

// g (dL/d relu(LN(x))) -> dL/dx in place; x holds the pre-LayerNorm values (float16-representable).  The row's LayerNorm-parameter
// gradients are added to dgam / dbet (per lane: the lane's four features of each tile, summed over the wave's rows at the end).
template <int FT>
__device__ __forceinline__ void ln_relu_bwd(f32x4 (&g)[FT], const uint2 (&xp)[FT], const float* gam, const float* bet, int q, bool ok,
                                            f32x4 (&dgam)[FT], f32x4 (&dbet)[FT]) {
  // x arrives PACKED (float16, as stored): 2 registers per tile instead of 4 across the two reductions -- the kernel sits at the
  // 256-register limit of two waves per SIMD, and the float16 MFMA co-executes with the extra conversions
  constexpr float inv_n = 1.0f / (16 * FT);
  float mean, rstd;
  {
    float sm = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      const f32x4 v = unpack4(xp[ft]);
      sm += (v[0] + v[1]) + (v[2] + v[3]);
    }
    mean = sumq(sm) * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      const f32x4 v = unpack4(xp[ft]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = v[r] - mean;
        d2 = fmaf(d, d, d2);
      }
    }
    rstd = 1.0f / sqrtf(sumq(d2) * inv_n + MDX_LN_EPS);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const f32x4 gm = lds4(gam + 16 * ft + 4 * q), bt = lds4(bet + 16 * ft + 4 * q);
    const f32x4 xh = (unpack4(xp[ft]) - splat4(mean)) * splat4(rstd);
    const f32x4 y = xh * gm + bt;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float go = (ok && y[r] > 0.f) ? g[ft][r] : 0.f;
      dgam[ft][r] = fmaf(go, xh[r], dgam[ft][r]);
      dbet[ft][r] += go;
      const float gh = go * gm[r];
      g[ft][r] = gh;
      s1 += gh;
      s2 = fmaf(gh, xh[r], s2);
    }
  }
  const float m1 = sumq(s1) * inv_n, m2 = sumq(s2) * inv_n;
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) {
    const f32x4 xh = (unpack4(xp[ft]) - splat4(mean)) * splat4(rstd);
    g[ft] = (g[ft] - splat4(m1) - xh * splat4(m2)) * splat4(rstd);
  }
}

// ---- column sums over the 16 rows of a tile by DPP (round 6, third session; the 256-wide version and the reasoning: colsum16 below)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
// one butterfly step on a pair of tiles: keep the tile of this lane's side, add the partner lane's value of it
template <int CTRL>
__device__ __forceinline__ f32x4 rs_comb(f32x4 lo, f32x4 hi, bool bit) {
  f32x4 out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float send = bit ? lo[r] : hi[r], keep = bit ? hi[r] : lo[r];
    out[r] = keep + dpp_row<CTRL>(send);
  }
  return out;
}
// 8 tiles (128 features) -> TWO floats per lane: three tile-halving steps (c ^ 8, c ^ 7, c ^ 2) leave tile b1 + 2 b2 + 4 b3 of this lane's q,
// the fourth (c ^ 1) splits its four values: lane (c, q) ends with the sums over the 16 rows of features
// 16 (c >> 1) + 4 q + 2 (c & 1) + {0, 1}.  Two registers per LayerNorm vector instead of the 32 per-lane partials the kernel carried
// through its tile loop (80 registers of accumulators = its 38 spilled ones).
template <class F>
__device__ __forceinline__ void colsum8x2(F&& val, int c, float& o0, float& o1) {
  constexpr int ROR8 = 0x128, HALF_MIRROR = 0x141, QP_X2 = 0x4E, QP_X1 = 0xB1;
  const bool b3 = (c & 8) != 0, b2 = (c & 4) != 0, b1 = (c & 2) != 0, b0 = (c & 1) != 0;
  auto k4 = [&](int j) { return rs_comb<ROR8>(val(j), val(j + 4), b3); };                 // tile j + 4 b3
  auto k2 = [&](int j) {                                                                    // tile j + 2 b2 + 4 b3
    const f32x4 a = k4(j);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 b = k4(j + 2);
    __builtin_amdgcn_sched_barrier(0);
    return rs_comb<HALF_MIRROR>(a, b, b2);
  };
  const f32x4 a = k2(0);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 b = k2(1);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 t = rs_comb<QP_X2>(a, b, b1);
  o0 = (b0 ? t[2] : t[0]) + dpp_row<QP_X1>(b0 ? t[0] : t[2]);
  o1 = (b0 ? t[3] : t[1]) + dpp_row<QP_X1>(b0 ? t[1] : t[3]);
}
// ln_relu_bwd<8> with the LayerNorm-parameter sums of the tile folded over its rows at once (acc: d gamma x 2, d beta x 2 of this lane's
// two features, see colsum8x2)
__device__ __forceinline__ void ln_relu_bwd8_rs(f32x4 (&g)[8], const uint2 (&xp)[8], const float* gam, const float* bet, int q, int c, bool ok,
                                                float (&acc)[4]) {
  constexpr float inv_n = 1.0f / 128;
  float mean, rstd;
  {
    float sm = 0.f;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      const f32x4 v = unpack4(xp[ft]);
      sm += (v[0] + v[1]) + (v[2] + v[3]);
    }
    mean = sumq(sm) * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int ft = 0; ft < 8; ++ft) {
      const f32x4 v = unpack4(xp[ft]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = v[r] - mean;
        d2 = fmaf(d, d, d2);
      }
    }
    rstd = 1.0f / sqrtf(sumq(d2) * inv_n + MDX_LN_EPS);
  }
#pragma unroll
  for (int ft = 0; ft < 8; ++ft) {   // go = g masked by the ReLU (zero for rows past the end)
    const f32x4 gm = lds4(gam + 16 * ft + 4 * q), bt = lds4(bet + 16 * ft + 4 * q);
    const f32x4 y = (unpack4(xp[ft]) - splat4(mean)) * splat4(rstd) * gm + bt;
#pragma unroll
    for (int r = 0; r < 4; ++r) g[ft][r] = (ok && y[r] > 0.f) ? g[ft][r] : 0.f;
  }
  {
    float s0, s1_;
    colsum8x2([&](int ft) { return g[ft]; }, c, s0, s1_);
    acc[2] += s0, acc[3] += s1_;
    __builtin_amdgcn_sched_barrier(0);
    colsum8x2([&](int ft) { return g[ft] * ((unpack4(xp[ft]) - splat4(mean)) * splat4(rstd)); }, c, s0, s1_);
    acc[0] += s0, acc[1] += s1_;
    __builtin_amdgcn_sched_barrier(0);
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int ft = 0; ft < 8; ++ft) {
    const f32x4 gm = lds4(gam + 16 * ft + 4 * q);
    const f32x4 xh = (unpack4(xp[ft]) - splat4(mean)) * splat4(rstd);
    g[ft] = g[ft] * gm;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1 += g[ft][r];
      s2 = fmaf(g[ft][r], xh[r], s2);
    }
  }
  const float m1 = sumq(s1) * inv_n, m2 = sumq(s2) * inv_n;
#pragma unroll
  for (int ft = 0; ft < 8; ++ft) {
    const f32x4 xh = (unpack4(xp[ft]) - splat4(mean)) * splat4(rstd);
    g[ft] = (g[ft] - splat4(m1) - xh * splat4(m2)) * splat4(rstd);
  }
}

// sum over the 16 lanes of a DPP row (the wave's 16 rows c = 0..15 of one q); result in every lane of the row
__device__ __forceinline__ float sum_c(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

__global__ __launch_bounds__(BF_THREADS) void bondffn_bwd_kernel(const mdx_bondffn_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t bf_smem[];
  uint16_t* S = bf_smem;
  float* C = reinterpret_cast<float*>(bf_smem + BwdLds::END);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  stage_wt<KO, KI, true>(S + BwdLds::WI2T, a.f.Wi2, (int)a.f.ldwi2, tid);    // (64 x 128)^T: 128 outputs over 64 inputs
  stage_wt<KI, KI, true>(S + BwdLds::WI1T, a.f.Wi1, (int)a.f.ldwi1, tid);
  stage_w<KI, KB, false>(S + BwdLds::WB, a.f.Wb, (int)a.f.ldwb, tid);         // recompute of bond_linear(X)
  stage_wt<KI, KB, true>(S + BwdLds::WBT, a.f.Wb, (int)a.f.ldwb, tid);        // (128 x 64)^T: 64 outputs over 128 inputs
  stage_wt<KO, KG, true>(S + BwdLds::WG2T, a.f.Wg2, (int)a.f.ldwg2, tid);     // (64 x 32)^T: 32 outputs over 64 inputs
  stage_wt<KG, KB, true>(S + BwdLds::WG1T, a.f.Wg1, (int)a.f.ldwg1, tid);     // (32 x 64)^T: 64 outputs over 32 inputs
  stage_v<128>(C + BwdLds::C_G1, a.f.g1, 1, tid); stage_v<128>(C + BwdLds::C_BE1, a.f.be1, 1, tid);
  stage_v<32>(C + BwdLds::C_GG, a.f.gg, 1, tid); stage_v<32>(C + BwdLds::C_GBE, a.f.gbe, 1, tid);
  __syncthreads();

  const int E = (int)a.f.E, ntiles = (E + 15) >> 4, nw = gridDim.x * BF_WAVES;
  const _Float16* X = reinterpret_cast<const _Float16*>(a.f.X);
  const _Float16* NL = reinterpret_cast<const _Float16*>(a.f.NL);
  const _Float16* s_pre1 = reinterpret_cast<const _Float16*>(a.f.pre1);
  const _Float16* s_inter = reinterpret_cast<const _Float16*>(a.f.inter);
  const _Float16* s_gpre = reinterpret_cast<const _Float16*>(a.f.gpre);
  const _Float16* s_gate = reinterpret_cast<const _Float16*>(a.f.gate);
  _Float16* o_ginter = reinterpret_cast<_Float16*>(a.g_inter);
  _Float16* o_ggate = reinterpret_cast<_Float16*>(a.g_gate);
  _Float16* o_gpre1 = reinterpret_cast<_Float16*>(a.g_pre1);
  _Float16* o_gbf = reinterpret_cast<_Float16*>(a.g_bf);
  _Float16* o_gnl = reinterpret_cast<_Float16*>(a.g_nl);
  _Float16* o_ggpre = reinterpret_cast<_Float16*>(a.g_gpre);
  _Float16* o_gx = reinterpret_cast<_Float16*>(a.g_x);
  const uint16_t* wi2t = S + BwdLds::WI2T + c * LDO + 8 * q;
  const uint16_t* wi1t = S + BwdLds::WI1T + c * LDI + 8 * q;
  const uint16_t* wb = S + BwdLds::WB + c * LDB + 8 * q;
  const uint16_t* wbt = S + BwdLds::WBT + c * LDI + 8 * q;
  const uint16_t* wg2t = S + BwdLds::WG2T + c * LDO + 8 * q;
  const uint16_t* wg1t = S + BwdLds::WG1T + c * LDG + 8 * q;

#ifndef MDX_BF_LNP_DPP
#define MDX_BF_LNP_DPP 1
#endif
#if MDX_BF_LNP_DPP
  float acc1[4] = {0.f, 0.f, 0.f, 0.f};   // inter LayerNorm: d gamma, d beta of features 16 (c >> 1) + 4 q + 2 (c & 1) + {0, 1} (colsum8x2)
  f32x4 dgg[2], dgb[2];
  zero<2>(dgg); zero<2>(dgb);
#else
  f32x4 dg1[8], db1[8], dgg[2], dgb[2];
  zero<8>(dg1); zero<8>(db1); zero<2>(dgg); zero<2>(dgb);
#endif

#pragma unroll 1
  for (int tile = blockIdx.x * BF_WAVES + wave; tile < ntiles; tile += nw) {
    const int row = 16 * tile + c;
    const bool ok = row < E;
    const size_t r = (size_t)min(row, E - 1);
    const int64_t ni = a.f.idx[r], no = a.oidx[r];
    uint4 xr[2];
    {
      const _Float16* p = X + r * a.f.ldx + 8 * q;
      xr[0] = *reinterpret_cast<const uint4*>(p);
      xr[1] = *reinterpret_cast<const uint4*>(p + 32);
    }
    // ---- out = inter * sigmoid(gate): the incoming gradient is dL/d(sum over the output node's rows), a float16 row per edge
    uint2 pgi[4], pgg[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x4 g = rh4(ldg4(a.gS + (size_t)no * a.ldgs + 16 * ft + 4 * q));
      const f32x4 it = ldh4(s_inter + r * KO + 16 * ft + 4 * q), sg = sigmoid4(ldh4(s_gate + r * KO + 16 * ft + 4 * q));
      pgi[ft] = pack4(g * sg);
      pgg[ft] = pack4(g * it * sg * (splat4(1.f) - sg));
    }
    st_tiles<4>(o_ginter + r * KO, pgi, q, ok);
    st_tiles<4>(o_ggate + r * KO, pgg, q, ok);
    __builtin_amdgcn_sched_barrier(0);
    // ---- inter module backward
    uint2 pgb[8];
    {
      f32x4 g[8];
      uint2 xp[8];
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) xp[ft] = *reinterpret_cast<const uint2*>(s_pre1 + r * KI + 16 * ft + 4 * q);
      const f16x8_t b2[2] = {pair8(pgi[0], pgi[1]), pair8(pgi[2], pgi[3])};
      zero<8>(g);
      mm<8, 2, LDO>(g, wi2t, b2);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) g[ft] = rh4(g[ft]);
#if MDX_BF_LNP_DPP
      ln_relu_bwd8_rs(g, xp, C + BwdLds::C_G1, C + BwdLds::C_BE1, q, c, ok, acc1);
#else
      ln_relu_bwd<8>(g, xp, C + BwdLds::C_G1, C + BwdLds::C_BE1, q, ok, dg1, db1);
#endif
      uint2 pg[8];
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        pg[ft] = pack4(g[ft]);
      }
      st_tiles<8>(o_gpre1 + r * KI, pg, q, ok);
      const f16x8_t b1[4] = {pair8(pg[0], pg[1]), pair8(pg[2], pg[3]), pair8(pg[4], pg[5]), pair8(pg[6], pg[7])};
      zero<8>(g);
      mm<8, 4, LDI>(g, wi1t, b1);     // dL/d prod
      __builtin_amdgcn_sched_barrier(0);
      // prod = bf * nl:  d bf = gp * nl,  d nl = gp * bf  (bf = bond_linear(X) recomputed: 16 MFMAs against a 256-byte row of HBM traffic each way)
      const f16x8_t xb[2] = {__builtin_bit_cast(f16x8_t, xr[0]), __builtin_bit_cast(f16x8_t, xr[1])};
      f32x4 x[8];
      zero<8>(x);
      mm<8, 2, LDB>(x, wb, xb);
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2) {
        uint2 pn[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ft = 2 * g2 + j;
          const f32x4 gp = rh4(g[ft]), nl = ldh4(NL + (size_t)ni * a.f.ldnl + 16 * ft + 4 * q);
          pgb[ft] = pack4(gp * nl);
          pn[j] = pack4(gp * rh4(x[ft]));
        }
        sth8_pair(o_gbf + r * KI, g2, pgb[2 * g2], pgb[2 * g2 + 1], q, ok);
        sth8_pair(o_gnl + r * KI, g2, pn[0], pn[1], q, ok);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- gate backward
    uint2 pgp[2];
    {
      f32x4 g[2];
      uint2 x[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) x[ft] = *reinterpret_cast<const uint2*>(s_gpre + r * KG + 16 * ft + 4 * q);
      const f16x8_t b2[2] = {pair8(pgg[0], pgg[1]), pair8(pgg[2], pgg[3])};
      zero<2>(g);
      mm<2, 2, LDO>(g, wg2t, b2);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) g[ft] = rh4(g[ft]);
      ln_relu_bwd<2>(g, x, C + BwdLds::C_GG, C + BwdLds::C_GBE, q, ok, dgg, dgb);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        pgp[ft] = pack4(g[ft]);
      }
      st_tiles<2>(o_ggpre + r * KG, pgp, q, ok);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dL/dX = bond_linear^T d bf + W_g1e^T d gate_pre  (two Linear data gradients, each a float16 tensor, summed in float16)
    {
      f32x4 g1[4], g2[4];
      const f16x8_t bb[4] = {pair8(pgb[0], pgb[1]), pair8(pgb[2], pgb[3]), pair8(pgb[4], pgb[5]), pair8(pgb[6], pgb[7])};
      zero<4>(g1);
      mm<4, 4, LDI>(g1, wbt, bb);
      const f16x8_t bg[1] = {pair8(pgp[0], pgp[1])};
      zero<4>(g2);
      mm<4, 1, LDG>(g2, wg1t, bg);
      {
        const uint2 px[4] = {pack4(rh4(g1[0]) + rh4(g2[0])), pack4(rh4(g1[1]) + rh4(g2[1])), pack4(rh4(g1[2]) + rh4(g2[2])),
                             pack4(rh4(g1[3]) + rh4(g2[3]))};
        st_tiles<4>(o_gx + r * KB, px, q, ok);
      }
    }
  }
  // ---- LayerNorm-parameter gradients: lanes -> rows of the wave (DPP row sums) -> the workgroup's 8 waves (LDS, fixed order) -> one
  // partial row per workgroup; the caller's deferred reduction sums the gridDim.x rows
  __syncthreads();   // every wave is done with the weights: the LDS area is free
  float* R = reinterpret_cast<float*>(bf_smem);
#if MDX_BF_LNP_DPP
  {
    const int f = 16 * (c >> 1) + 4 * q + 2 * (c & 1);
    R[wave * BF_LNP + f] = acc1[0], R[wave * BF_LNP + f + 1] = acc1[1];
    R[wave * BF_LNP + 128 + f] = acc1[2], R[wave * BF_LNP + 128 + f + 1] = acc1[3];
  }
#else
#pragma unroll
  for (int ft = 0; ft < 8; ++ft)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float v1 = sum_c(dg1[ft][s]), v2 = sum_c(db1[ft][s]);
      if (c == 0) {
        R[wave * BF_LNP + 16 * ft + 4 * q + s] = v1;
        R[wave * BF_LNP + 128 + 16 * ft + 4 * q + s] = v2;
      }
    }
#endif
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float v1 = sum_c(dgg[ft][s]), v2 = sum_c(dgb[ft][s]);
      if (c == 0) {
        R[wave * BF_LNP + 256 + 16 * ft + 4 * q + s] = v1;
        R[wave * BF_LNP + 288 + 16 * ft + 4 * q + s] = v2;
      }
    }
  __syncthreads();
  for (int i = tid; i < BF_LNP; i += BF_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < BF_WAVES; ++w) s += R[w * BF_LNP + i];
    a.lnp[(size_t)blockIdx.x * BF_LNP + i] = s;
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// EdgeBlock tail + residual (reference models/graph.py:281-294 and the `h_edge = h_edge + ...` of :360): per row
//   pre = self_ffn(h) + (BL[l] + BR[r]);  new = h + out_transform(relu(LN(pre)))
// with BL = S_L + node_ffn_left(h_node), BR = S_R + node_ffn_right(h_node) computed per NODE by the caller.  One launch instead of
// gather, gather, add, Linear+LayerNorm, Linear, add; backward: one launch + the two weight-gradient contractions + two segment sums.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int ET_THREADS = 512, ET_WAVES = 8, ET_LNP = 128;
struct EtLds {
  static constexpr int WS = 0, WO = WS + KB * LDB, END = WO + KB * LDB;
  static constexpr int C_BS = 0, C_G = 64, C_B = 128, C_BO = 192, C_END = 256;
  static constexpr int BYTES = END * 2 + C_END * 4;
};

__global__ __launch_bounds__(ET_THREADS) void edge_tail_fwd_kernel(const mdx_edge_tail_args a) {
  __shared__ __attribute__((aligned(16))) uint16_t S[EtLds::END];
  __shared__ __attribute__((aligned(16))) float C[EtLds::C_END];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  stage_w<KB, KB, false>(S + EtLds::WS, a.Ws, (int)a.ldws, tid);
  stage_w<KB, KB, true>(S + EtLds::WO, a.Wo, (int)a.ldwo, tid);
  stage_v<64>(C + EtLds::C_BS, a.bs, 1, tid); stage_v<64>(C + EtLds::C_G, a.lng, 1, tid); stage_v<64>(C + EtLds::C_B, a.lnb, 1, tid);
  stage_v<64>(C + EtLds::C_BO, a.bo, 1, tid);
  __syncthreads();
  const int E = (int)a.E, ntiles = (E + 15) >> 4, nw = gridDim.x * ET_WAVES;
  const _Float16* H = reinterpret_cast<const _Float16*>(a.H);
  const _Float16* BL = reinterpret_cast<const _Float16*>(a.BL);
  const _Float16* BR = reinterpret_cast<const _Float16*>(a.BR);
  _Float16* o_pre = reinterpret_cast<_Float16*>(a.pre);
  _Float16* o_post = reinterpret_cast<_Float16*>(a.post);
  _Float16* o_out = reinterpret_cast<_Float16*>(a.out);
  const uint16_t* ws = S + EtLds::WS + c * LDB + 8 * q;
  const uint16_t* wo = S + EtLds::WO + c * LDB + 8 * q;
#pragma unroll 1
  for (int tile = blockIdx.x * ET_WAVES + wave; tile < ntiles; tile += nw) {
    const int row = 16 * tile + c;
    const bool ok = row < E;
    const size_t r = (size_t)min(row, E - 1);
    const _Float16* p = H + r * a.ldh + 8 * q;
    const uint4 x0 = *reinterpret_cast<const uint4*>(p), x1 = *reinterpret_cast<const uint4*>(p + 32);
    const int64_t il = a.il[r], ir = a.ir[r];
    f32x4 ad[4], hres[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      ad[ft] = rh4(ldh4(BL + (size_t)il * a.ldbl + 16 * ft + 4 * q) + ldh4(BR + (size_t)ir * a.ldbr + 16 * ft + 4 * q));
      hres[ft] = ldh4(H + r * a.ldh + 16 * ft + 4 * q);
    }
    const f16x8_t xb[2] = {__builtin_bit_cast(f16x8_t, x0), __builtin_bit_cast(f16x8_t, x1)};
    f32x4 y[4];
    zero<4>(y);
    mm<4, 2, LDB>(y, ws, xb);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      y[ft] = rh4((y[ft] + lds4(C + EtLds::C_BS + 16 * ft + 4 * q)) + ad[ft]);
    }
    {
      const uint2 pp[4] = {pack4(y[0]), pack4(y[1]), pack4(y[2]), pack4(y[3])};
      st_tiles<4>(o_pre + r * KB, pp, q, ok);
    }
    float mean, rstd;
    ln_stats<4>(y, mean, rstd);
    uint2 pk[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      pk[ft] = pack4(relu4((y[ft] - splat4(mean)) * splat4(rstd) * lds4(C + EtLds::C_G + 16 * ft + 4 * q) + lds4(C + EtLds::C_B + 16 * ft + 4 * q)));
    }
    st_tiles<4>(o_post + r * KB, pk, q, ok);
    const f16x8_t hb[2] = {pair8(pk[0], pk[1]), pair8(pk[2], pk[3])};
    zero<4>(y);
    mm<4, 2, LDB>(y, wo, hb);
    {
      uint2 po[4];
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) po[ft] = pack4(hres[ft] + rh4(y[ft] + lds4(C + EtLds::C_BO + 16 * ft + 4 * q)));
      st_tiles<4>(o_out + r * KB, po, q, ok);
    }
  }
}

__global__ __launch_bounds__(ET_THREADS) void edge_tail_bwd_kernel(const mdx_edge_tail_bwd_args a) {
  __shared__ __attribute__((aligned(16))) uint16_t S[EtLds::END];
  __shared__ __attribute__((aligned(16))) float C[ET_WAVES * ET_LNP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  stage_wt<KB, KB, true>(S + EtLds::WS, a.f.Ws, (int)a.f.ldws, tid);   // self_ffn^T
  stage_wt<KB, KB, true>(S + EtLds::WO, a.f.Wo, (int)a.f.ldwo, tid);   // out_transform^T
  stage_v<64>(C + 0, a.f.lng, 1, tid); stage_v<64>(C + 64, a.f.lnb, 1, tid);
  __syncthreads();
  const int E = (int)a.f.E, ntiles = (E + 15) >> 4, nw = gridDim.x * ET_WAVES;
  const _Float16* G = reinterpret_cast<const _Float16*>(a.g_out);
  const _Float16* s_pre = reinterpret_cast<const _Float16*>(a.f.pre);
  _Float16* o_gpre = reinterpret_cast<_Float16*>(a.g_pre);
  _Float16* o_gh = reinterpret_cast<_Float16*>(a.g_h);
  const uint16_t* wst = S + EtLds::WS + c * LDB + 8 * q;
  const uint16_t* wot = S + EtLds::WO + c * LDB + 8 * q;
  f32x4 dg[4], db[4];
  zero<4>(dg); zero<4>(db);
#pragma unroll 1
  for (int tile = blockIdx.x * ET_WAVES + wave; tile < ntiles; tile += nw) {
    const int row = 16 * tile + c;
    const bool ok = row < E;
    const size_t r = (size_t)min(row, E - 1);
    uint2 pg[4], x[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      pg[ft] = *reinterpret_cast<const uint2*>(G + r * a.ldg + 16 * ft + 4 * q);
      x[ft] = *reinterpret_cast<const uint2*>(s_pre + r * KB + 16 * ft + 4 * q);
    }
    const f16x8_t gb[2] = {pair8(pg[0], pg[1]), pair8(pg[2], pg[3])};
    f32x4 g[4];
    zero<4>(g);
    mm<4, 2, LDB>(g, wot, gb);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) g[ft] = rh4(g[ft]);
    ln_relu_bwd<4>(g, x, C + 0, C + 64, q, ok, dg, db);
    uint2 pp[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      pp[ft] = pack4(g[ft]);
    }
    st_tiles<4>(o_gpre + r * KB, pp, q, ok);
    const f16x8_t pb[2] = {pair8(pp[0], pp[1]), pair8(pp[2], pp[3])};
    zero<4>(g);
    mm<4, 2, LDB>(g, wst, pb);
    {
      uint2 ph[4];
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) ph[ft] = pack4(unpack4(pg[ft]) + rh4(g[ft]));   // residual + self_ffn^T
      st_tiles<4>(o_gh + r * KB, ph, q, ok);
    }
  }
  __syncthreads();
#pragma unroll
  for (int ft = 0; ft < 4; ++ft)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float v1 = sum_c(dg[ft][s]), v2 = sum_c(db[ft][s]);
      if (c == 0) {
        C[wave * ET_LNP + 16 * ft + 4 * q + s] = v1;
        C[wave * ET_LNP + 64 + 16 * ft + 4 * q + s] = v2;
      }
    }
  __syncthreads();
  for (int i = tid; i < ET_LNP; i += ET_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < ET_WAVES; ++w) s += C[w * ET_LNP + i];
    a.lnp[(size_t)blockIdx.x * ET_LNP + i] = s;
  }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// PosUpdate, front of its BondFFN (reference models/graph.py:384-392 with :133-141, bond 64 / node 64 / inter 256 / out 1):
//   a    = left_lin_edge(h)[l] * right_lin_edge(h)[r]                      (the two per-node MLPs are hoisted by the caller: LF, RF)
//   prod = bond_linear(X) * node_linear(a)                                 (E,256): input of the inter MLP (its 256 x 256 Linear + LayerNorm
//                                                                           stays the per-operator launch: the weight does not fit LDS beside these)
//   gate = W_g2 relu(LN(W_g1x X + W_g1a a + t w_t + b)) + b_g2             (E,1)
// One launch instead of gather, mul_gather, 2 Linears, mul, 2 partial Linears, Linear+LayerNorm, Linear; backward: one launch + six
// weight-gradient contractions + two segment sums.
// ------------------------------------------------------------------------------------------------------------------------------------
constexpr int PF_THREADS = 512, PF_WAVES = 8, PF_LNP = 64, KW = 256, LDW = KW + 8;
struct PfLds {   // forward: Wb, Wn (256 x 64), Wg1x, Wg1a (32 x 64)
  static constexpr int WB = 0, WN = WB + KW * LDB, WGX = WN + KW * LDB, WGA = WGX + KG * LDB, END = WGA + KG * LDB;
  static constexpr int C_BG1 = 0, C_GG = 32, C_GBE = 64, C_WT = 96, C_WG2 = 128, C_END = 160;
  static constexpr int BYTES = END * 2 + C_END * 4;
};
struct PbLds {   // backward: Wb, Wn (recompute), their transposes (64 x 256), the gate's first-layer transposes (64 x 32)
  static constexpr int WB = 0, WN = WB + KW * LDB, WBT = WN + KW * LDB, WNT = WBT + KB * LDW, WGXT = WNT + KB * LDW, WGAT = WGXT + KB * LDG,
                       END = WGAT + KB * LDG;
  static constexpr int C_GG = 0, C_GBE = 32, C_WG2 = 64, C_END = 96;
  static constexpr int BYTES = END * 2 + C_END * 4 + PF_WAVES * PF_LNP * 4;
};

// a = LF[il] * RF[ir] as the B operand of the Linears that read it (8 consecutive features of row c per lane and k-step), float16
__device__ __forceinline__ void load_a(f16x8_t (&ab)[2], uint4 (&araw)[2], const _Float16* LF, size_t ol, const _Float16* RF, size_t orr, int q) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const f16x8_t l = *reinterpret_cast<const f16x8_t*>(LF + ol + 32 * ks + 8 * q), r = *reinterpret_cast<const f16x8_t*>(RF + orr + 32 * ks + 8 * q);
    f16x8_t p;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = (_Float16)((float)l[j] * (float)r[j]);
    ab[ks] = p;
    araw[ks] = __builtin_bit_cast(uint4, p);
  }
}

__global__ __launch_bounds__(PF_THREADS) void posffn_fwd_kernel(const mdx_posffn_args a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t bf_smem[];
  uint16_t* S = bf_smem;
  float* C = reinterpret_cast<float*>(bf_smem + PfLds::END);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  stage_w<KW, KB, false>(S + PfLds::WB, a.Wb, (int)a.ldwb, tid);
  stage_w<KW, KB, false>(S + PfLds::WN, a.Wn, (int)a.ldwn, tid);
  stage_w<KG, KB, false>(S + PfLds::WGX, a.Wg1x, (int)a.ldwg1x, tid);
  stage_w<KG, KB, false>(S + PfLds::WGA, a.Wg1a, (int)a.ldwg1a, tid);
  stage_v<32>(C + PfLds::C_BG1, a.bg1, 1, tid); stage_v<32>(C + PfLds::C_GG, a.gg, 1, tid); stage_v<32>(C + PfLds::C_GBE, a.gbe, 1, tid);
  stage_v<32>(C + PfLds::C_WT, a.Wt, (int)a.ldwt, tid, true); stage_v<32>(C + PfLds::C_WG2, a.Wg2, 1, tid, true);
  __syncthreads();
  const int E = (int)a.E, ntiles = (E + 15) >> 4, nw = gridDim.x * PF_WAVES;
  const _Float16* X = reinterpret_cast<const _Float16*>(a.X);
  const _Float16* LF = reinterpret_cast<const _Float16*>(a.LF);
  const _Float16* RF = reinterpret_cast<const _Float16*>(a.RF);
  _Float16* o_a = reinterpret_cast<_Float16*>(a.a);
  _Float16* o_prod = reinterpret_cast<_Float16*>(a.prod);
  _Float16* o_gpre = reinterpret_cast<_Float16*>(a.gpre);
  _Float16* o_gpost = reinterpret_cast<_Float16*>(a.gpost);
  _Float16* o_gate = reinterpret_cast<_Float16*>(a.gate);
  const uint16_t* wb = S + PfLds::WB + c * LDB + 8 * q;
  const uint16_t* wn = S + PfLds::WN + c * LDB + 8 * q;
  const uint16_t* wgx = S + PfLds::WGX + c * LDB + 8 * q;
  const uint16_t* wga = S + PfLds::WGA + c * LDB + 8 * q;
  const float bg2 = a.bg2 ? a.bg2[0] : 0.f;
#pragma unroll 1
  for (int tile = blockIdx.x * PF_WAVES + wave; tile < ntiles; tile += nw) {
    const int row = 16 * tile + c;
    const bool ok = row < E;
    const size_t r = (size_t)min(row, E - 1);
    const _Float16* px = X + r * a.ldx + 8 * q;
    const f16x8_t xb[2] = {*reinterpret_cast<const f16x8_t*>(px), *reinterpret_cast<const f16x8_t*>(px + 32)};
    f16x8_t ab[2];
    uint4 araw[2];
    load_a(ab, araw, LF, (size_t)a.il[r] * a.ldlf, RF, (size_t)a.ir[r] * a.ldrf, q);
    const float th = rh(a.te[r]);
    if (ok) {
      *reinterpret_cast<uint4*>(o_a + r * KB + 8 * q) = araw[0];
      *reinterpret_cast<uint4*>(o_a + r * KB + 32 + 8 * q) = araw[1];
    }
    // ---- prod = bond_linear(X) * node_linear(a), 128 features at a time
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 yb[8], yn[8];
      zero<8>(yb); zero<8>(yn);
      mm<8, 2, LDB>(yb, wb + 128 * h * LDB, xb);
      mm<8, 2, LDB>(yn, wn + 128 * h * LDB, ab);
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2)
        sth8_pair(o_prod + r * KW + 128 * h, g2, pack4(rh4(yb[2 * g2]) * rh4(yn[2 * g2])), pack4(rh4(yb[2 * g2 + 1]) * rh4(yn[2 * g2 + 1])), q, ok);
    }
    // ---- gate: the node (a) and time columns of its first Linear are an fp32 addend of the bond columns' product, like the per-operator path
    {
      f32x4 gx[2], ga[2];
      zero<2>(gx); zero<2>(ga);
      mm<2, 2, LDB>(gx, wgx, xb);
      mm<2, 2, LDB>(ga, wga, ab);
      uint2 pg[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const f32x4 ad = splat4(th) * lds4(C + PfLds::C_WT + 16 * ft + 4 * q) + ga[ft];
        gx[ft] = rh4((gx[ft] + lds4(C + PfLds::C_BG1 + 16 * ft + 4 * q)) + ad);
      }
      sth8_pair(o_gpre + r * KG, 0, pack4(gx[0]), pack4(gx[1]), q, ok);
      float mean, rstd;
      ln_stats<2>(gx, mean, rstd);
      float dot = 0.f;
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const f32x4 v = rh4(relu4((gx[ft] - splat4(mean)) * splat4(rstd) * lds4(C + PfLds::C_GG + 16 * ft + 4 * q) + lds4(C + PfLds::C_GBE + 16 * ft + 4 * q)));
        pg[ft] = pack4(v);
        const f32x4 w2 = lds4(C + PfLds::C_WG2 + 16 * ft + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; ++s) dot = fmaf(v[s], w2[s], dot);
      }
      sth8_pair(o_gpost + r * KG, 0, pg[0], pg[1], q, ok);
      dot = sumq(dot);
      if (ok && q == 0) o_gate[r] = (_Float16)(dot + bg2);
    }
  }
}

__global__ __launch_bounds__(PF_THREADS) void posffn_bwd_kernel(const mdx_posffn_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t bf_smem[];
  uint16_t* S = bf_smem;
  float* C = reinterpret_cast<float*>(bf_smem + PbLds::END);
  float* R = C + PbLds::C_END;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  stage_w<KW, KB, false>(S + PbLds::WB, a.f.Wb, (int)a.f.ldwb, tid);
  stage_w<KW, KB, false>(S + PbLds::WN, a.f.Wn, (int)a.f.ldwn, tid);
  stage_wt<KW, KB, true>(S + PbLds::WBT, a.f.Wb, (int)a.f.ldwb, tid);        // (256 x 64)^T: 64 outputs over 256 inputs
  stage_wt<KW, KB, true>(S + PbLds::WNT, a.f.Wn, (int)a.f.ldwn, tid);
  stage_wt<KG, KB, true>(S + PbLds::WGXT, a.f.Wg1x, (int)a.f.ldwg1x, tid);   // (32 x 64)^T: 64 outputs over 32 inputs
  stage_wt<KG, KB, true>(S + PbLds::WGAT, a.f.Wg1a, (int)a.f.ldwg1a, tid);
  stage_v<32>(C + PbLds::C_GG, a.f.gg, 1, tid); stage_v<32>(C + PbLds::C_GBE, a.f.gbe, 1, tid); stage_v<32>(C + PbLds::C_WG2, a.f.Wg2, 1, tid, true);
  __syncthreads();
  const int E = (int)a.f.E, ntiles = (E + 15) >> 4, nw = gridDim.x * PF_WAVES;
  const _Float16* X = reinterpret_cast<const _Float16*>(a.f.X);
  const _Float16* LF = reinterpret_cast<const _Float16*>(a.f.LF);
  const _Float16* RF = reinterpret_cast<const _Float16*>(a.f.RF);
  const _Float16* GP = reinterpret_cast<const _Float16*>(a.g_prod);
  const _Float16* GG = reinterpret_cast<const _Float16*>(a.g_gate);
  const _Float16* s_gpre = reinterpret_cast<const _Float16*>(a.f.gpre);
  _Float16* o_gbf = reinterpret_cast<_Float16*>(a.g_bf);
  _Float16* o_gnf = reinterpret_cast<_Float16*>(a.g_nf);
  _Float16* o_ggpre = reinterpret_cast<_Float16*>(a.g_gpre);
  _Float16* o_gx = reinterpret_cast<_Float16*>(a.g_x);
  _Float16* o_glf = reinterpret_cast<_Float16*>(a.g_lf);
  _Float16* o_grf = reinterpret_cast<_Float16*>(a.g_rf);
  const uint16_t* wb = S + PbLds::WB + c * LDB + 8 * q;
  const uint16_t* wn = S + PbLds::WN + c * LDB + 8 * q;
  const uint16_t* wbt = S + PbLds::WBT + c * LDW + 8 * q;
  const uint16_t* wnt = S + PbLds::WNT + c * LDW + 8 * q;
  const uint16_t* wgxt = S + PbLds::WGXT + c * LDG + 8 * q;
  const uint16_t* wgat = S + PbLds::WGAT + c * LDG + 8 * q;
  f32x4 dgg[2], dgb[2];
  zero<2>(dgg); zero<2>(dgb);
#pragma unroll 1
  for (int tile = blockIdx.x * PF_WAVES + wave; tile < ntiles; tile += nw) {
    const int row = 16 * tile + c;
    const bool ok = row < E;
    const size_t r = (size_t)min(row, E - 1);
    const size_t ol = (size_t)a.f.il[r] * a.f.ldlf, orr = (size_t)a.f.ir[r] * a.f.ldrf;
    const _Float16* px = X + r * a.f.ldx + 8 * q;
    const f16x8_t xb[2] = {*reinterpret_cast<const f16x8_t*>(px), *reinterpret_cast<const f16x8_t*>(px + 32)};
    f16x8_t ab[2];
    uint4 araw[2];
    load_a(ab, araw, LF, ol, RF, orr, q);
    // ---- prod = bf * nf:  d bf = g * nf,  d nf = g * bf  (both Linears recomputed: 64 MFMAs against 2 KB per edge of HBM traffic)
    uint2 pgb[16], pgn[16];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 yb[8], yn[8];
      zero<8>(yb); zero<8>(yn);
      mm<8, 2, LDB>(yb, wb + 128 * h * LDB, xb);
      mm<8, 2, LDB>(yn, wn + 128 * h * LDB, ab);
#pragma unroll
      for (int ft = 0; ft < 8; ++ft) {
        const f32x4 g = ldh4(GP + r * a.ldgp + 128 * h + 16 * ft + 4 * q);
        pgb[8 * h + ft] = pack4(g * rh4(yn[ft]));
        pgn[8 * h + ft] = pack4(g * rh4(yb[ft]));
      }
#pragma unroll
      for (int g2 = 0; g2 < 4; ++g2) {
        sth8_pair(o_gbf + r * KW + 128 * h, g2, pgb[8 * h + 2 * g2], pgb[8 * h + 2 * g2 + 1], q, ok);
        sth8_pair(o_gnf + r * KW + 128 * h, g2, pgn[8 * h + 2 * g2], pgn[8 * h + 2 * g2 + 1], q, ok);
      }
    }
    f32x4 gx1[4], ga1[4];
    {
      f16x8_t bb[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bb[ks] = pair8(pgb[2 * ks], pgb[2 * ks + 1]);
      zero<4>(gx1);
      mm<4, 8, LDW>(gx1, wbt, bb);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bb[ks] = pair8(pgn[2 * ks], pgn[2 * ks + 1]);
      zero<4>(ga1);
      mm<4, 8, LDW>(ga1, wnt, bb);
    }
    // ---- gate backward: gate = w_g2 . relu(LN(gpre)) + b
    uint2 pgp[2];
    {
      const float gg = (float)GG[r];
      f32x4 g[2];
      uint2 xp[2];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        xp[ft] = *reinterpret_cast<const uint2*>(s_gpre + r * KG + 16 * ft + 4 * q);
        g[ft] = rh4(splat4(gg) * lds4(C + PbLds::C_WG2 + 16 * ft + 4 * q));
      }
      ln_relu_bwd<2>(g, xp, C + PbLds::C_GG, C + PbLds::C_GBE, q, ok, dgg, dgb);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        pgp[ft] = pack4(g[ft]);
      }
      sth8_pair(o_ggpre + r * KG, 0, pgp[0], pgp[1], q, ok);
    }
    {
      const f16x8_t bg[1] = {pair8(pgp[0], pgp[1])};
      f32x4 gx2[4], ga2[4];
      zero<4>(gx2); zero<4>(ga2);
      mm<4, 1, LDG>(gx2, wgxt, bg);
      mm<4, 1, LDG>(ga2, wgat, bg);
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        uint2 px[2], pl[2], pr[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ft = 2 * g2 + j;
          px[j] = pack4(rh4(gx1[ft]) + rh4(gx2[ft]));
          const f32x4 ga = rh4(rh4(ga1[ft]) + rh4(ga2[ft]));
          pl[j] = pack4(ga * ldh4(RF + orr + 16 * ft + 4 * q));
          pr[j] = pack4(ga * ldh4(LF + ol + 16 * ft + 4 * q));
        }
        sth8_pair(o_gx + r * KB, g2, px[0], px[1], q, ok);
        sth8_pair(o_glf + r * KB, g2, pl[0], pl[1], q, ok);
        sth8_pair(o_grf + r * KB, g2, pr[0], pr[1], q, ok);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float v1 = sum_c(dgg[ft][s]), v2 = sum_c(dgb[ft][s]);
      if (c == 0) {
        R[wave * PF_LNP + 16 * ft + 4 * q + s] = v1;
        R[wave * PF_LNP + 32 + 16 * ft + 4 * q + s] = v2;
      }
    }
  __syncthreads();
  for (int i = tid; i < PF_LNP; i += PF_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < PF_WAVES; ++w) s += R[w * PF_LNP + i];
    a.lnp[(size_t)blockIdx.x * PF_LNP + i] = s;
  }
}

int g_ncu = 0;
int ncus() {
  if (!g_ncu) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_ncu = p.multiProcessorCount;
    if (g_ncu <= 0) g_ncu = 256;
  }
  return g_ncu;
}
bool g_attr_fwd = false, g_attr_bwd = false;

int check_fwd(const mdx_bondffn_args& a, const char* who) {
  if (a.E < 0) return mdx_set_error(MDX_ERR_ARG, "bondffn: negative row count");
  if (!a.X || !a.Wb || !a.Wi1 || !a.Wi2 || !a.Wg1 || !a.Wg2 || !a.Wt || !a.NL || !a.GN || !a.idx || !a.te || !a.bi1 || !a.g1 || !a.be1 ||
      !a.bi2 || !a.bg1 || !a.gg || !a.gbe || !a.bg2)
    return mdx_set_error(MDX_ERR_ARG, "bondffn: null operand");
  if ((a.ldx & 7) || (reinterpret_cast<uintptr_t>(a.X) & 15)) return mdx_set_error(MDX_ERR_ARG, "bondffn: X rows must be 16-byte aligned");
  if ((a.ldnl & 3) || (reinterpret_cast<uintptr_t>(a.NL) & 7)) return mdx_set_error(MDX_ERR_ARG, "bondffn: NL rows must be 8-byte aligned");
  if ((a.ldgn & 3) || (reinterpret_cast<uintptr_t>(a.GN) & 15)) return mdx_set_error(MDX_ERR_ARG, "bondffn: GN rows must be 16-byte aligned");
  (void)who;
  return MDX_OK;
}

int check_tail(const mdx_edge_tail_args& a) {
  if (a.E < 0) return mdx_set_error(MDX_ERR_ARG, "edge_tail: negative row count");
  if (!a.H || !a.BL || !a.BR || !a.il || !a.ir || !a.Ws || !a.bs || !a.lng || !a.lnb || !a.Wo || !a.bo) return mdx_set_error(MDX_ERR_ARG, "edge_tail: null operand");
  if ((a.ldh & 7) || (reinterpret_cast<uintptr_t>(a.H) & 15)) return mdx_set_error(MDX_ERR_ARG, "edge_tail: H rows must be 16-byte aligned");
  if ((a.ldbl & 3) || (a.ldbr & 3) || (reinterpret_cast<uintptr_t>(a.BL) & 7) || (reinterpret_cast<uintptr_t>(a.BR) & 7))
    return mdx_set_error(MDX_ERR_ARG, "edge_tail: node rows must be 8-byte aligned");
  return MDX_OK;
}

}  // namespace

extern "C" int mdx_op_bondffn_workgroups(void) { return ncus(); }
extern "C" int mdx_op_bondffn_lnp_floats(void) { return BF_LNP; }

extern "C" int mdx_op_bondffn_fwd(const mdx_bondffn_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "bondffn_fwd: null argument block");
  if (a->E == 0) return MDX_OK;
  if (int rc = check_fwd(*a, "fwd")) return rc;
  if (!a->prod || !a->pre1 || !a->post1 || !a->inter || !a->gpre || !a->gpost || !a->gate || !a->out)
    return mdx_set_error(MDX_ERR_ARG, "bondffn_fwd: null output");
  if (!g_attr_fwd) {
    if (hipFuncSetAttribute((const void*)bondffn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FwdLds::BYTES) != hipSuccess)
      return mdx_set_error(MDX_ERR_HIP, "bondffn_fwd: cannot reserve LDS");
    g_attr_fwd = true;
  }
  const int ntiles = (int)((a->E + 15) / 16);
  const int grid = std::max(1, std::min(ncus(), (ntiles + BF_FWD_WAVES - 1) / BF_FWD_WAVES));
  hipLaunchKernelGGL(bondffn_fwd_kernel, dim3(grid), dim3(BF_FWD_THREADS), FwdLds::BYTES, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "bondffn_fwd: launch failed");
}

extern "C" int mdx_op_bondffn_bwd(const mdx_bondffn_bwd_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "bondffn_bwd: null argument block");
  if (int rc = check_fwd(a->f, "bwd")) return rc;
  if (!a->f.pre1 || !a->f.inter || !a->f.gpre || !a->f.gate || !a->gS || !a->oidx || !a->g_inter || !a->g_gate || !a->g_pre1 || !a->g_bf ||
      !a->g_nl || !a->g_gpre || !a->g_x || !a->lnp)
    return mdx_set_error(MDX_ERR_ARG, "bondffn_bwd: null operand");
  if ((a->ldgs & 3) || (reinterpret_cast<uintptr_t>(a->gS) & 15)) return mdx_set_error(MDX_ERR_ARG, "bondffn_bwd: gS rows must be 16-byte aligned");
  if (!g_attr_bwd) {
    if (hipFuncSetAttribute((const void*)bondffn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, BwdLds::BYTES) != hipSuccess)
      return mdx_set_error(MDX_ERR_HIP, "bondffn_bwd: cannot reserve LDS");
    g_attr_bwd = true;
  }
  // ALWAYS the full grid: the caller's reduction of the LayerNorm-parameter partials reads mdx_op_bondffn_workgroups() rows
  // (workgroups without tiles write zero rows)
  hipLaunchKernelGGL(bondffn_bwd_kernel, dim3(ncus()), dim3(BF_THREADS), BwdLds::BYTES, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "bondffn_bwd: launch failed");
}

extern "C" int mdx_op_edge_tail_lnp_floats(void) { return ET_LNP; }

extern "C" int mdx_op_edge_tail_fwd(const mdx_edge_tail_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "edge_tail_fwd: null argument block");
  if (a->E == 0) return MDX_OK;
  if (int rc = check_tail(*a)) return rc;
  if (!a->pre || !a->post || !a->out) return mdx_set_error(MDX_ERR_ARG, "edge_tail_fwd: null output");
  const int ntiles = (int)((a->E + 15) / 16);
  const int grid = std::max(1, std::min(2 * ncus(), (ntiles + ET_WAVES - 1) / ET_WAVES));
  hipLaunchKernelGGL(edge_tail_fwd_kernel, dim3(grid), dim3(ET_THREADS), 0, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "edge_tail_fwd: launch failed");
}

/* always mdx_op_bondffn_workgroups() workgroups: that many partial rows of LayerNorm-parameter gradients (128 floats: d gamma | d beta) */
extern "C" int mdx_op_edge_tail_bwd(const mdx_edge_tail_bwd_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "edge_tail_bwd: null argument block");
  if (int rc = check_tail(a->f)) return rc;
  if (!a->f.pre || !a->g_out || !a->g_pre || !a->g_h || !a->lnp) return mdx_set_error(MDX_ERR_ARG, "edge_tail_bwd: null operand");
  if ((a->ldg & 3) || (reinterpret_cast<uintptr_t>(a->g_out) & 7)) return mdx_set_error(MDX_ERR_ARG, "edge_tail_bwd: gradient rows must be 8-byte aligned");
  hipLaunchKernelGGL(edge_tail_bwd_kernel, dim3(ncus()), dim3(ET_THREADS), 0, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "edge_tail_bwd: launch failed");
}

extern "C" int mdx_op_posffn_lnp_floats(void) { return PF_LNP; }

static int check_posffn(const mdx_posffn_args& a) {
  if (a.E < 0) return mdx_set_error(MDX_ERR_ARG, "posffn: negative row count");
  if (!a.X || !a.LF || !a.RF || !a.il || !a.ir || !a.te || !a.Wb || !a.Wn || !a.Wg1x || !a.Wg1a || !a.Wt || !a.bg1 || !a.gg || !a.gbe || !a.Wg2)
    return mdx_set_error(MDX_ERR_ARG, "posffn: null operand");
  if ((a.ldx & 7) || (a.ldlf & 7) || (a.ldrf & 7) || (reinterpret_cast<uintptr_t>(a.X) & 15) || (reinterpret_cast<uintptr_t>(a.LF) & 15) ||
      (reinterpret_cast<uintptr_t>(a.RF) & 15))
    return mdx_set_error(MDX_ERR_ARG, "posffn: X / LF / RF rows must be 16-byte aligned");
  return MDX_OK;
}
static bool g_attr_pf = false, g_attr_pb = false;

extern "C" int mdx_op_posffn_fwd(const mdx_posffn_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "posffn_fwd: null argument block");
  if (a->E == 0) return MDX_OK;
  if (int rc = check_posffn(*a)) return rc;
  if (!a->a || !a->prod || !a->gpre || !a->gpost || !a->gate) return mdx_set_error(MDX_ERR_ARG, "posffn_fwd: null output");
  if (!g_attr_pf) {
    if (hipFuncSetAttribute((const void*)posffn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PfLds::BYTES) != hipSuccess)
      return mdx_set_error(MDX_ERR_HIP, "posffn_fwd: cannot reserve LDS");
    g_attr_pf = true;
  }
  const int ntiles = (int)((a->E + 15) / 16);
  const int grid = std::max(1, std::min(ncus(), (ntiles + PF_WAVES - 1) / PF_WAVES));
  hipLaunchKernelGGL(posffn_fwd_kernel, dim3(grid), dim3(PF_THREADS), PfLds::BYTES, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "posffn_fwd: launch failed");
}

extern "C" int mdx_op_posffn_bwd(const mdx_posffn_bwd_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "posffn_bwd: null argument block");
  if (int rc = check_posffn(a->f)) return rc;
  if (!a->f.gpre || !a->g_prod || !a->g_gate || !a->g_bf || !a->g_nf || !a->g_gpre || !a->g_x || !a->g_lf || !a->g_rf || !a->lnp)
    return mdx_set_error(MDX_ERR_ARG, "posffn_bwd: null operand");
  if ((a->ldgp & 3) || (reinterpret_cast<uintptr_t>(a->g_prod) & 7)) return mdx_set_error(MDX_ERR_ARG, "posffn_bwd: gradient rows must be 8-byte aligned");
  if (!g_attr_pb) {
    if (hipFuncSetAttribute((const void*)posffn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PbLds::BYTES) != hipSuccess)
      return mdx_set_error(MDX_ERR_HIP, "posffn_bwd: cannot reserve LDS");
    g_attr_pb = true;
  }
  hipLaunchKernelGGL(posffn_bwd_kernel, dim3(ncus()), dim3(PF_THREADS), PbLds::BYTES, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "posffn_bwd: launch failed");
}

// ------------------------------------------------------------------------------------------------------------------------------------
// NodeBlock message path (reference models/graph.py:40-50: edge_net, the product with node_net(x)[col], msg_net, the gate MLP on
// [edge | x[col] | t[col]] and the sigmoid product) as ONE forward and ONE backward launch.  Its three 256 x 256 weights (128 KiB each as
// float16) do not fit LDS, so here -- unlike the kernels above -- the weights STREAM: a per-call pack kernel writes every weight of the
// chain as float16 MFMA A-operand fragments ([k-step][feature tile][lane][8 halves], the k permutation of the accumulator-fed layers
// baked in, transposes for the backward), and a wave fetches the 1-KiB fragment of each v_mfma_f32_16x16x32_f16 straight from L2 with
// one 16-byte load per lane, eight fragments in flight ahead of the MFMAs that consume them.  The chain itself is the design of the
// kernels above: a wave owns 16 rows, every later layer's B operand is the previous layer's accumulators, LayerNorm is wave-local.
// LDS is free, so the backward's 256-wide LayerNorm-parameter gradients (they would cost 2 x 128 accumulator registers per lane) are
// formed by transposing each tile through a wave-private 16.25-KiB LDS area: a lane then owns four FEATURES and adds the 16 rows in
// order.
// ------------------------------------------------------------------------------------------------------------------------------------
#ifndef MDX_NM_FWD_THREADS
#define MDX_NM_FWD_THREADS 512   // (768 = three waves per SIMD measured 369 us against 362 us: the LDS pipe, not occupancy, is what the tile loop waits on)
#endif
constexpr int NM_THREADS = 512, NM_WAVES = 8, NM_LNP = 1024;
constexpr int NMF_THREADS = MDX_NM_FWD_THREADS, NMF_WAVES = NMF_THREADS / 64;
// workgroups per CU the forward is launched for: with 256 threads TWO independent workgroups share a CU (one wave per SIMD each, 76 KiB
// of LDS each) -- a workgroup's waves run in barrier lockstep on the shared weight buffer, so whatever stalls one wave (its epilogue's
// store burst: gfx9 counts stores on vmcnt, a wave's next weight fetch is usable only after them) stalls all of them; two groups stall
// independently
constexpr int NMF_WG_PER_CU = NMF_THREADS <= 256 ? 2 : 1;
constexpr int NMF_WPS = NMF_THREADS <= 512 ? 2 : 3;   // waves per SIMD the forward is compiled for   // forward: 156 VGPRs, three waves per SIMD fit   // partial row: d gamma_e | d beta_e | d gamma_g | d beta_g (256 each)

__global__ void pack_a_kernel(const mdx_pack_jobs a) {
  const mdx_pack_job& jb = a.job[blockIdx.y];
  const int FT = jb.n_out / 16;
  const size_t total = (size_t)jb.n_out * jb.n_in;
  for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    const int j8 = (int)(o & 7), lane = (int)((o >> 3) & 63);
    const size_t rest = o >> 9;
    const int ft = (int)(rest % FT), ks = (int)(rest / FT);
    const int c = lane & 15, q = lane >> 4;
    const int n = 16 * ft + c;
    const int k = jb.perm ? 16 * (2 * ks + (j8 >> 2)) + 4 * q + (j8 & 3) : 32 * ks + 8 * q + j8;
    const float v = jb.trans ? jb.W[(size_t)k * jb.ld + n] : jb.W[(size_t)n * jb.ld + k];
    reinterpret_cast<_Float16*>(jb.out)[o] = (_Float16)v;
  }
}

// y[ft] += W x over KS k-steps, A fragments from a global pack ([ks][ft][lane][8]); fragments of the next half-step are requested
// before the MFMAs of the current one (8 x 16 bytes per lane in flight)
template <int FT, int KS>
__device__ __forceinline__ void mmg(f32x4 (&y)[FT], const _Float16* __restrict__ pack, int lane, const f16x8_t (&x)[KS]) {
  constexpr int G = 4;                       // fragments per group (two groups = 8 x 16 bytes per lane in flight)
  constexpr int NG = KS * (FT / G);
  // buffer loads: scalar base + the lane's byte offset + an immediate fragment offset -- no per-fragment vector address arithmetic
  // (with flat loads the 128 hoisted 64-bit lane addresses of a 256 x 256 layer were what spilled)
  asm volatile("" : "+s"(pack));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(pack), 0, -1, 0x00020000);
  const unsigned lo = 16u * (unsigned)lane;
  auto frag = [&](int i) { return __builtin_bit_cast(f16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rs, lo, i * 1024, 0)); };
  f16x8_t a[2][G];
#pragma unroll
  for (int i = 0; i < G; ++i) a[0][i] = frag(i);
  static_for<0, NG>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    constexpr int ks = g / (FT / G), f0 = (g % (FT / G)) * G;
    if constexpr (g + 1 < NG) {
#pragma unroll
      for (int i = 0; i < G; ++i) a[(g + 1) & 1][i] = frag((g + 1) * G + i);
    }
#pragma unroll
    for (int i = 0; i < G; ++i) y[f0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[g & 1][i], x[ks], y[f0 + i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  });
}

// The same product with the weight fragments shared by the workgroup (round 6, second cut): the 8 waves of a workgroup run the same layer
// at the same time, so a k-step's FT fragments (FT KiB) are fetched from L2 ONCE per workgroup -- every thread copies 16 or 32 bytes
// into one of two LDS buffers -- and each wave reads its A operands from LDS.  With private streams (mmg) every wave pulled the whole
// 448 KiB of a tile's weights through L2 by itself: 4.3 GB per launch, ~10 TB/s, and the kernels ran at that rate (428 / 540 us).
// One barrier per k-step: the buffer a step writes was last read two steps earlier, and no wave can be more than one barrier ahead.
// `par` (the buffer parity) runs through consecutive calls.  Every wave of the workgroup must make the same calls.
// D = k-steps of fragments in flight per thread (round 6, third cut): with one, every k-step waited for its own L2 round trip -- 28
// k-steps of ~2.5 us per 16-row tile WAS the kernel's time (pipes 26 % busy, profiles/HISTORY.md); the LDS double buffer stays (a buffer
// is rewritten two barriers after its last read), only the registers between L2 and LDS get deeper.
#ifndef MDX_NM_DEPTH_F
#define MDX_NM_DEPTH_F 4
#endif
#ifndef MDX_NM_DEPTH_B
#define MDX_NM_DEPTH_B 2
#endif
template <int FT, int KS, int NT = NM_THREADS, int D = MDX_NM_DEPTH_B>
__device__ __forceinline__ void mmw(f32x4 (&y)[FT], const _Float16* __restrict__ pack, uint16_t* wbuf, int& par, int tid, int lane,
                                    const f16x8_t (&x)[KS]) {
  constexpr int CH = FT * 1024;                                      // bytes per k-step
  constexpr int NL = (CH + NT * 16 - 1) / (NT * 16);                  // 16-byte copies per thread and k-step
  asm volatile("" : "+s"(pack));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(pack), 0, -1, 0x00020000);
  uint4 st[D][NL];
  auto fetch = [&](auto slot, int ks) {
    constexpr int sl = decltype(slot)::value;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const unsigned o = (unsigned)(tid + i * NT) * 16u;
      if (CH % (NT * 16) == 0 || o < (unsigned)CH) st[sl][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, ks * CH, 0));
    }
  };
  static_for<0, (D < KS ? D : KS)>([&](auto dc) { fetch(dc, decltype(dc)::value); });
  static_for<0, KS>([&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    char* buf = reinterpret_cast<char*>(wbuf) + ((par + ks) & 1) * 16384;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const unsigned o = (unsigned)(tid + i * NT) * 16u;
      if (CH % (NT * 16) == 0 || o < (unsigned)CH) *reinterpret_cast<uint4*>(buf + o) = st[ks % D][i];
    }
    if constexpr (ks + D < KS) fetch(std::integral_constant<int, ks % D>{}, ks + D);
    // raw barrier: __syncthreads() would also wait for vmcnt(0), i.e. for the fetch just issued and for every global store of the
    // previous epilogue (gfx9 counts stores on vmcnt) -- only the LDS writes have to be complete here
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* ab = buf + lane * 16;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) {
      y[ft] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const f16x8_t*>(ab + ft * 1024), x[ks], y[ft], 0, 0, 0);
      if (ft % 8 == 7) __builtin_amdgcn_sched_barrier(0);   // at most eight fragments' reads ahead of their MFMAs (32 registers)
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  par = (par + KS) & 1;
}

template <int N>
__device__ __forceinline__ void pairs(f16x8_t (&b)[N / 2], const uint2 (&pk)[N]) {
#pragma unroll
  for (int i = 0; i < N / 2; ++i) b[i] = pair8(pk[2 * i], pk[2 * i + 1]);
}

// Whole 16 x 256 float16 tiles through a wave-private LDS area (round 6): chunks go in in accumulator layout (lane (c, q): row c, columns
// 16 ft + 4 q .. + 3), full rows come out -- lanes 0..31 read row 2 i, lanes 32..63 row 2 i + 1, 16 bytes each, so a store instruction writes
// two complete 512-byte rows (1 KiB contiguous) instead of 16 rows x 32 or 64 bytes.  Row stride 528 bytes (132 dwords): the 8-byte writes
// of a half wave and the 16-byte reads of a quarter wave each cover the 64 banks once.  Only the owning wave touches its area (LDS
// operations of a wave execute in order: no barrier).
constexpr int TS_LD = 264;                       // halves per row of the area
constexpr int TS_BYTES = 16 * TS_LD * 2;         // 8,448 bytes per wave
__device__ __forceinline__ void ts_put(uint16_t* T, int ft, uint2 h, int c, int q) { *reinterpret_cast<uint2*>(T + c * TS_LD + 16 * ft + 4 * q) = h; }
__device__ __forceinline__ void ts_put2(uint16_t* T, int g2, f16x8_t b, int c, int q) {   // the two tiles of a B operand (pair8)
  const uint4 u = __builtin_bit_cast(uint4, b);
  ts_put(T, 2 * g2, uint2{u.x, u.y}, c, q);
  ts_put(T, 2 * g2 + 1, uint2{u.z, u.w}, c, q);
}
// The rows leave through BUFFER stores: a resource per output (scalar base, size = the matrix: rows past the end are dropped by the
// bounds check, no predicate), the tile's byte offset as the scalar offset, one lane offset for all eight stores, the row pair as the
// immediate.  (With flat stores the compiler kept eight 64-bit addresses per output alive: 274 spilled registers in the backward.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ts_rsrc(_Float16* out, int E) {
  return __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((unsigned)E * (unsigned)(KW * 2)), 0x00020000);
}
template <int B>
__device__ __forceinline__ void ts_flush_b(const uint16_t* T, __amdgpu_buffer_rsrc_t rs, int tile, int lane) {
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(tile) * (unsigned)(16 * KW * 2);
  const unsigned vo = (unsigned)(lane >> 5) * (unsigned)(KW * 2) + 16u * (unsigned)(lane & 31);
  const uint16_t* src = T + (lane >> 5) * TS_LD + 8 * (lane & 31);
#pragma unroll
  for (int i0 = 0; i0 < 8; i0 += B) {
    uint4 v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) v[i] = *reinterpret_cast<const uint4*>(src + 2 * (i0 + i) * TS_LD);
#pragma unroll
    for (int i = 0; i < B; ++i) {
      const int ii = i0 + i;
      typedef unsigned int u32x4_t __attribute__((__vector_size__(4 * sizeof(unsigned int))));
#ifdef MDX_TRAIN_NT
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[i]), rs, vo, so + (unsigned)ii * 1024u, 2);   // nt
#else
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v[i]), rs, vo, so + (unsigned)ii * 1024u, 0);
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void ts_flush(const uint16_t* T, __amdgpu_buffer_rsrc_t rs, int tile, int lane) { ts_flush_b<8>(T, rs, tile, lane); }

__global__ __launch_bounds__(NMF_THREADS, NMF_WPS) void nodemsg_fwd_kernel(const mdx_nodemsg_args a) {
  __shared__ __attribute__((aligned(16))) float C[9 * 256];
  __shared__ __attribute__((aligned(16))) uint16_t wbuf[2 * 8192];     // two k-steps of weight fragments (mmw)
  extern __shared__ __attribute__((aligned(16))) uint16_t nm_ts[];     // NMF_WAVES transposition areas (ts_put / ts_flush)
  int par = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint16_t* T = nm_ts + (size_t)wave * (TS_BYTES / 2);
  const int c = lane & 15, q = lane >> 4;
  {
    const float* src[9] = {a.b1e, a.lng_e, a.lnb_e, a.b2e, a.bm, a.bg1, a.lng_g, a.lnb_g, a.bg2};
    for (int i = tid; i < 9 * 256; i += NMF_THREADS) C[i] = src[i >> 8][i & 255];
  }
  __syncthreads();
  const float *c_b1e = C, *c_ge = C + 256, *c_be = C + 512, *c_b2e = C + 768, *c_bm = C + 1024, *c_bg1 = C + 1280, *c_gg = C + 1536,
              *c_gb = C + 1792, *c_bg2 = C + 2048;
  const int E = (int)a.E, ntiles = (E + 15) >> 4, nw = gridDim.x * NMF_WAVES;
  const int iters = (ntiles + nw - 1) / nw;     // the same for every wave of the grid: the weight copies are workgroup-cooperative
  const _Float16* X = reinterpret_cast<const _Float16*>(a.X);
  const _Float16* HN = reinterpret_cast<const _Float16*>(a.HN);
  const _Float16 *w1e = reinterpret_cast<const _Float16*>(a.pk_w1e), *w2e = reinterpret_cast<const _Float16*>(a.pk_w2e),
                 *wm = reinterpret_cast<const _Float16*>(a.pk_wm), *wg1 = reinterpret_cast<const _Float16*>(a.pk_wg1),
                 *wg2 = reinterpret_cast<const _Float16*>(a.pk_wg2);
  _Float16 *o_hepre = reinterpret_cast<_Float16*>(a.he_pre), *o_hepost = reinterpret_cast<_Float16*>(a.he_post),
           *o_he = reinterpret_cast<_Float16*>(a.he), *o_p = reinterpret_cast<_Float16*>(a.p), *o_m0 = reinterpret_cast<_Float16*>(a.m0),
           *o_gpre = reinterpret_cast<_Float16*>(a.g_pre), *o_gpost = reinterpret_cast<_Float16*>(a.g_post),
           *o_gt = reinterpret_cast<_Float16*>(a.gt), *o_msg = reinterpret_cast<_Float16*>(a.msg);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int tile = it * nw + blockIdx.x * NMF_WAVES + wave;
    const int row = 16 * tile + c;
#ifdef MDX_NM_NOSTORE            // ablation (timing only, results are garbage): no epilogue stores -- profiles/r6_ab_nodemsg_stores.txt
    const bool ok = row < E && a.E < 0;
#else
    const bool ok = row < E;      // (a wave past the last tile computes on the clamped last row and stores nothing)
#endif
    const size_t r = (size_t)min(row, E - 1), ro = r * KW + 4 * q, rb = r * KW;
    const int64_t nc = a.col[r];
    const _Float16* px = X + r * a.ldx + 8 * q;
    const f16x8_t xb[2] = {*reinterpret_cast<const f16x8_t*>(px), *reinterpret_cast<const f16x8_t*>(px + 32)};
    f32x4 y[16];
    f16x8_t b8[8];
    // B operand of the next layer from the tiles as they are produced (two tiles = one k-step); SB: keep the per-tile loads next to
    // their use (hoisted together, 16 x 4 registers of LayerNorm parameters / gathered rows spill)
#define NM_SB(ft) if ((ft) % 4 == 3) __builtin_amdgcn_sched_barrier(0)
    // ---- edge_net: Linear -> LayerNorm -> ReLU -> Linear, then the product with node_net(x)[col]
    zero<16>(y);
    mmw<16, 2, NMF_THREADS, MDX_NM_DEPTH_F>(y, w1e, wbuf, par, tid, lane, xb);
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      y[2 * g2] = rh4(y[2 * g2] + lds4(c_b1e + 32 * g2 + 4 * q));
      y[2 * g2 + 1] = rh4(y[2 * g2 + 1] + lds4(c_b1e + 32 * g2 + 16 + 4 * q));
      ts_put(T, 2 * g2, pack4(y[2 * g2]), c, q);
      ts_put(T, 2 * g2 + 1, pack4(y[2 * g2 + 1]), c, q);
      NM_SB(2 * g2 + 1);
    }
    ts_flush(T, ts_rsrc(o_hepre, E), tile, lane);
    {
      float mean, rstd;
      ln_stats<16>(y, mean, rstd);
#pragma unroll
      for (int g2 = 0; g2 < 8; ++g2) {
        uint2 h[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ft = 2 * g2 + j;
          h[j] = pack4(relu4((y[ft] - splat4(mean)) * splat4(rstd) * lds4(c_ge + 16 * ft + 4 * q) + lds4(c_be + 16 * ft + 4 * q)));
        }
        b8[g2] = pair8(h[0], h[1]);
        ts_put2(T, g2, b8[g2], c, q);
        NM_SB(2 * g2 + 1);
      }
    }
    ts_flush(T, ts_rsrc(o_hepost, E), tile, lane);
    zero<16>(y);
    mmw<16, 8, NMF_THREADS, MDX_NM_DEPTH_F>(y, w2e, wbuf, par, tid, lane, b8);
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      uint2 h[2], hh[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ft = 2 * g2 + j;
        const f32x4 he = rh4(y[ft] + lds4(c_b2e + 16 * ft + 4 * q));
        hh[j] = pack4(he);
        h[j] = pack4(he * ldh4(HN + (size_t)nc * a.ldhn + 16 * ft + 4 * q));
      }
      ts_put(T, 2 * g2, hh[0], c, q);
      ts_put(T, 2 * g2 + 1, hh[1], c, q);
      b8[g2] = pair8(h[0], h[1]);
      NM_SB(2 * g2 + 1);
    }
    ts_flush(T, ts_rsrc(o_he, E), tile, lane);
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) ts_put2(T, g2, b8[g2], c, q);
    ts_flush(T, ts_rsrc(o_p, E), tile, lane);
    // ---- msg_net (m0 is read back at the end: L2-hot, and 32 registers cheaper than holding it across the gate chain)
    zero<16>(y);
    mmw<16, 8, NMF_THREADS, MDX_NM_DEPTH_F>(y, wm, wbuf, par, tid, lane, b8);
    if (ok) {
#pragma unroll
      for (int ft = 0; ft < 16; ++ft) {
        sth4(o_m0 + ro + 16 * ft, pack4(y[ft] + lds4(c_bm + 16 * ft + 4 * q)));
        NM_SB(ft);
      }
    }
    // ---- gate: Linear([edge | x[col] | t[col]]) with the node / time columns as the hoisted fp32 addend PN[col]
    zero<16>(y);
    mmw<16, 2, NMF_THREADS, MDX_NM_DEPTH_F>(y, wg1, wbuf, par, tid, lane, xb);
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ft = 2 * g2 + j;
        y[ft] = rh4((y[ft] + lds4(c_bg1 + 16 * ft + 4 * q)) + ldg4(a.PN + (size_t)nc * a.ldpn + 16 * ft + 4 * q));
      }
      ts_put(T, 2 * g2, pack4(y[2 * g2]), c, q);
      ts_put(T, 2 * g2 + 1, pack4(y[2 * g2 + 1]), c, q);
      NM_SB(2 * g2 + 1);
    }
    ts_flush(T, ts_rsrc(o_gpre, E), tile, lane);
    {
      float mean, rstd;
      ln_stats<16>(y, mean, rstd);
#pragma unroll
      for (int g2 = 0; g2 < 8; ++g2) {
        uint2 h[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ft = 2 * g2 + j;
          h[j] = pack4(relu4((y[ft] - splat4(mean)) * splat4(rstd) * lds4(c_gg + 16 * ft + 4 * q) + lds4(c_gb + 16 * ft + 4 * q)));
        }
        b8[g2] = pair8(h[0], h[1]);
        ts_put2(T, g2, b8[g2], c, q);
        NM_SB(2 * g2 + 1);
      }
    }
    ts_flush(T, ts_rsrc(o_gpost, E), tile, lane);
    zero<16>(y);
    mmw<16, 8, NMF_THREADS, MDX_NM_DEPTH_F>(y, wg2, wbuf, par, tid, lane, b8);
    // (m0 keeps the per-lane 8-byte pattern above: every lane reads back the bytes it wrote itself)
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      uint2 hg[2], hm[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ft = 2 * g2 + j;
        const f32x4 g = rh4(y[ft] + lds4(c_bg2 + 16 * ft + 4 * q));
        hg[j] = pack4(g);
        hm[j] = pack4(ldh4(o_m0 + ro + 16 * ft) * rh4(sigmoid4(g)));
      }
      ts_put(T, 2 * g2, hg[0], c, q);
      ts_put(T, 2 * g2 + 1, hg[1], c, q);
      sth8_pair(o_msg + rb, g2, hm[0], hm[1], q, ok);
      NM_SB(2 * g2 + 1);
    }
    ts_flush(T, ts_rsrc(o_gt, E), tile, lane);
  }
}

// Column sums over the 16 rows of a tile (fp32, accumulator layout: lane (q, c) holds features 16 ft + 4 q .. of row c), added to acc
// (lane L owns features 4 L .. 4 L + 3): the tile goes through the wave's LDS area T [16][NM_TLD] one feature tile at a time (cs_put),
// then every lane adds its four features of rows 0..15 in order (cs_sum).
constexpr int NM_TLD = 132;   // half a row (128 features) + 4: the 256 columns go through in two halves (LDS: 8.25 KiB per wave)
__device__ __forceinline__ void cs_put(float* T, int ft8, f32x4 v, int c, int q) { *reinterpret_cast<f32x4*>(T + c * NM_TLD + 16 * ft8 + 4 * q) = v; }
// lanes 32 hf .. 32 hf + 31 own the features of half hf: lane L adds features 4 L .. 4 L + 3 of rows 0..15 in order
__device__ __forceinline__ void cs_sum(f32x4& acc, const float* T, int lane, int hf) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {
    // every lane reads (lanes of the other half read the same words and add zero: no divergent branch around the sums)
    const float keep = ((lane >> 5) == hf) ? 1.0f : 0.0f;
    f32x4 part = splat4(0.f);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {   // four rows in flight (all 16 at once cost 64 registers where the kernel has none to spare):
      part = part + *reinterpret_cast<const volatile f32x4*>(T + rr * NM_TLD + 4 * (lane & 31));
      if (rr % 4 == 3) asm volatile("" : "+v"(part));   // pinned every four rows, so the reads cannot all be hoisted above the adds
    }
    acc = acc + part * splat4(keep);
  }
  __builtin_amdgcn_wave_barrier();   // the area is rewritten only after every lane has read it
}

// Column sums over the 16 rows of a tile WITHOUT the LDS transposition (round 6, third session): a reduce-scatter butterfly over the 16
// lanes c of a lane row (= one DPP row; q, the feature quad, is the same in all of them).  Step 1 pairs c with c ^ 8 (row_ror:8): a lane
// keeps the eight tiles whose bit 3 equals its own and adds the partner's values of them; step 2 (row_half_mirror: c ^ 7, same bit 3,
// other bit 2) halves again, steps 3 / 4 (quad_perm: c ^ 2, c ^ 1) end with ONE tile per lane: lane (c, q) holds the sum over the 16
// rows of features 16 c + 4 q .. + 3.  60 DPP adds + 120 selects per call instead of 8 LDS writes, a wave barrier and 16 dependent LDS
// reads per half (the LDS version cost the backward 71 us of 458 per launch and the registers that made it spill).  Deterministic; the
// order of additions is the butterfly's (pairs of rows), not rows 0..15 in sequence.
#ifndef MDX_NM_LNP_DPP
#define MDX_NM_LNP_DPP 1
#endif
// depth first (a tile pair is folded as soon as both halves exist): at most ~6 tiles of temporaries alive instead of the 8 + 4 of a
// step-by-step butterfly -- the kernel has no registers to spare
template <class F>
__device__ __forceinline__ f32x4 colsum16(F&& val, int c) {
  constexpr int ROR8 = 0x128, HALF_MIRROR = 0x141, QP_X2 = 0x4E, QP_X1 = 0xB1;
  const bool b3 = (c & 8) != 0, b2 = (c & 4) != 0, b1 = (c & 2) != 0, b0 = (c & 1) != 0;
  auto k8 = [&](int j) { return rs_comb<ROR8>(val(j), val(j + 8), b3); };                 // tile j + 8 b3
  auto k4 = [&](int j) {                                                                    // tile j + 4 b2 + 8 b3
    const f32x4 a = k8(j);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 b = k8(j + 4);
    __builtin_amdgcn_sched_barrier(0);
    return rs_comb<HALF_MIRROR>(a, b, b2);
  };
  auto k2 = [&](int j) {
    const f32x4 a = k4(j);
    __builtin_amdgcn_sched_barrier(0);
    const f32x4 b = k4(j + 2);
    __builtin_amdgcn_sched_barrier(0);
    return rs_comb<QP_X2>(a, b, b1);
  };
  const f32x4 a = k2(0);
  __builtin_amdgcn_sched_barrier(0);
  const f32x4 b = k2(1);
  __builtin_amdgcn_sched_barrier(0);
  return rs_comb<QP_X1>(a, b, b0);
}

// A 16 x 256 float16 tile of a stored tensor into the wave's LDS area, the inverse of ts_flush (round 6, third session): eight buffer loads
// of two complete 512-byte rows each (lanes 0..31 row 2 i, lanes 32..63 row 2 i + 1, 16 bytes per lane; rows past the end read as zero
// through the bounds check), written as they are; ts_get then returns the accumulator-layout chunk (row c, columns 16 ft + 4 q .. + 3)
// the direct 8-byte loads fetched -- 16 rows x 32 bytes per instruction, which cost the backward 48 us per launch (profiles/HISTORY.md:
// row-0 ablation).  The tile STAYS in LDS while the LayerNorm backward reads it six times: 32 registers less than holding it.
typedef unsigned int tsu4_t __attribute__((ext_vector_type(4)));
typedef unsigned int tsu2_t __attribute__((ext_vector_type(2)));
template <int B>
__device__ __forceinline__ void ts_load(uint16_t* T, const _Float16* src, int E, int tile, int lane) {
  const __amdgpu_buffer_rsrc_t rs = ts_rsrc(const_cast<_Float16*>(src), E);
  const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(tile) * (unsigned)(16 * KW * 2);
  const unsigned vo = (unsigned)(lane >> 5) * (unsigned)(KW * 2) + 16u * (unsigned)(lane & 31);
  uint16_t* dst = T + (lane >> 5) * TS_LD + 8 * (lane & 31);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // every lane has finished with the area's previous content
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i0 = 0; i0 < 8; i0 += B) {
    uint4 v[B];
#pragma unroll
    for (int i = 0; i < B; ++i) v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so + (unsigned)(i0 + i) * 1024u, 0));
#pragma unroll
    for (int i = 0; i < B; ++i) *reinterpret_cast<volatile tsu4_t*>(dst + 2 * (i0 + i) * TS_LD) = __builtin_bit_cast(tsu4_t, v[i]);
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint2 ts_get(const uint16_t* T, int ft, int c, int q) {
  const tsu2_t v = *reinterpret_cast<const volatile tsu2_t*>(T + c * TS_LD + 16 * ft + 4 * q);
  return uint2{v.x, v.y};
}
// Measured (third session): with the running sums in LDS the kernel does not spill either way (228 VGPRs direct, 250 staged), and the
// staged tile's LDS traffic and wave barriers cost more than its load pattern saves: 392.6 us direct against 418.7 us staged on the same
// box.  The direct 8-byte loads are the default; 1 keeps the staged form.
#ifndef MDX_NM_TSLOAD
#define MDX_NM_TSLOAD 0
#endif
static_assert(!MDX_NM_TSLOAD || MDX_NM_LNP_DPP, "the staged tile lives in the area the LDS column sums would use");

// 256-wide LayerNorm + ReLU backward in place (g: dL/d output -> dL/d pre-activation); xsrc = the stored pre-activation (E x 256
// float16), read through the wave's LDS area (MDX_NM_TSLOAD) or directly (xrow = this lane's row + 4 q); the rows' contributions to
// d gamma / d beta are summed over the tile by the DPP butterfly (or the LDS column sums)
__device__ __forceinline__ void ln256_relu_bwd(f32x4 (&g)[16], const _Float16* xsrc, const _Float16* xrow, int E, int tile, const float* gam,
                                               const float* bet, int q, bool ok, float* pgam, float* pbet, float* T, int lane) {
  constexpr float inv_n = 1.0f / 256;
  const int c = lane & 15;
#if MDX_NM_TSLOAD
  const uint16_t* Th = reinterpret_cast<const uint16_t*>(T);
  ts_load<4>(reinterpret_cast<uint16_t*>(T), xsrc, E, tile, lane);
#define XP(ft) ts_get(Th, (ft), c, q)
#else
  uint2 xp[16];
#pragma unroll
  for (int ft = 0; ft < 16; ++ft) xp[ft] = *reinterpret_cast<const uint2*>(xrow + 16 * ft);
#define XP(ft) xp[ft]
#endif
  float mean, rstd;
  {
    float sm = 0.f;
#pragma unroll
    for (int ft = 0; ft < 16; ++ft) {
      const f32x4 v = unpack4(XP(ft));
      sm += (v[0] + v[1]) + (v[2] + v[3]);
    }
    mean = sumq(sm) * inv_n;
    float d2 = 0.f;
#pragma unroll
    for (int ft = 0; ft < 16; ++ft) {
      const f32x4 v = unpack4(XP(ft));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float d = v[s] - mean;
        d2 = fmaf(d, d, d2);
      }
    }
    rstd = 1.0f / sqrtf(sumq(d2) * inv_n + MDX_LN_EPS);
  }
  // go = g masked by the ReLU (zero for rows past the end): d beta += go, d gamma += go * x_hat
#pragma unroll
  for (int ft = 0; ft < 16; ++ft) {
    const f32x4 gm = lds4(gam + 16 * ft + 4 * q), bt = lds4(bet + 16 * ft + 4 * q);
    const f32x4 xh = (unpack4(XP(ft)) - splat4(mean)) * splat4(rstd);
    const f32x4 yv = xh * gm + bt;
#pragma unroll
    for (int s = 0; s < 4; ++s) g[ft][s] = (ok && yv[s] > 0.f) ? g[ft][s] : 0.f;
    if (ft % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
#if defined(MDX_NM_NOLNP)         // (ablation, timing only: without the LayerNorm-parameter column sums)
#elif MDX_NM_LNP_DPP              // lane (c, q) accumulates features 16 c + 4 q .. + 3
  // the running sums live in LDS (this lane's own four floats per vector: no other lane touches them): as registers they were alive
  // across the whole tile loop -- 16 of them were the difference between no spills and ~60
  *reinterpret_cast<f32x4*>(pbet) = *reinterpret_cast<const f32x4*>(pbet) + colsum16([&](int ft) { return g[ft]; }, c);
  __builtin_amdgcn_sched_barrier(0);
  *reinterpret_cast<f32x4*>(pgam) = *reinterpret_cast<const f32x4*>(pgam) +
                                    colsum16([&](int ft) { return g[ft] * ((unpack4(XP(ft)) - splat4(mean)) * splat4(rstd)); }, c);
  __builtin_amdgcn_sched_barrier(0);
#else                             // lane L accumulates features 4 L .. 4 L + 3 (LDS column sums)
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
    for (int f8 = 0; f8 < 8; ++f8) cs_put(T, f8, g[8 * hf + f8], c, q);
    f32x4 acc = *reinterpret_cast<const f32x4*>(pbet);
    cs_sum(acc, T, lane, hf);
    *reinterpret_cast<f32x4*>(pbet) = acc;
#pragma unroll
    for (int f8 = 0; f8 < 8; ++f8) cs_put(T, f8, g[8 * hf + f8] * ((unpack4(XP(8 * hf + f8)) - splat4(mean)) * splat4(rstd)), c, q);
    acc = *reinterpret_cast<const f32x4*>(pgam);
    cs_sum(acc, T, lane, hf);
    *reinterpret_cast<f32x4*>(pgam) = acc;
  }
#endif
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int ft = 0; ft < 16; ++ft) {
    const f32x4 gm = lds4(gam + 16 * ft + 4 * q);
    const f32x4 xh = (unpack4(XP(ft)) - splat4(mean)) * splat4(rstd);
    g[ft] = g[ft] * gm;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      s1 += g[ft][s];
      s2 = fmaf(g[ft][s], xh[s], s2);
    }
    if (ft % 4 == 3) __builtin_amdgcn_sched_barrier(0);
  }
  const float m1 = sumq(s1) * inv_n, m2 = sumq(s2) * inv_n;
#pragma unroll
  for (int ft = 0; ft < 16; ++ft) {
    const f32x4 xh = (unpack4(XP(ft)) - splat4(mean)) * splat4(rstd);
    g[ft] = (g[ft] - splat4(m1) - xh * splat4(m2)) * splat4(rstd);
  }
}
#undef XP

__global__ __launch_bounds__(NM_THREADS) void nodemsg_bwd_kernel(const mdx_nodemsg_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) uint16_t bf_smem[];
  uint16_t* wbuf = bf_smem;                                  // two k-steps of weight fragments (mmw), 32 KiB
  float* C = reinterpret_cast<float*>(bf_smem + 2 * 8192);    // 4 x 256 LayerNorm parameters
  float* Tall = C + 1024;                                     // NM_WAVES areas of 16 x NM_TLD floats (>= NM_WAVES x NM_LNP floats for the end)
  int par = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  {
    const float* src[4] = {a.f.lng_e, a.f.lnb_e, a.f.lng_g, a.f.lnb_g};
    for (int i = tid; i < 1024; i += NM_THREADS) C[i] = src[i >> 8][i & 255];
  }
  __syncthreads();
  float* T = Tall + (size_t)wave * 16 * NM_TLD;
  const int E = (int)a.f.E, ntiles = (E + 15) >> 4, nw = gridDim.x * NM_WAVES;
  const int iters = (ntiles + nw - 1) / nw;
  const _Float16* HN = reinterpret_cast<const _Float16*>(a.f.HN);
  const _Float16 *s_hepre = reinterpret_cast<const _Float16*>(a.f.he_pre), *s_he = reinterpret_cast<const _Float16*>(a.f.he),
                 *s_m0 = reinterpret_cast<const _Float16*>(a.f.m0), *s_gpre = reinterpret_cast<const _Float16*>(a.f.g_pre),
                 *s_gt = reinterpret_cast<const _Float16*>(a.f.gt);
  const _Float16 *wg2t = reinterpret_cast<const _Float16*>(a.pk_wg2t), *wg1t = reinterpret_cast<const _Float16*>(a.pk_wg1t),
                 *wmt = reinterpret_cast<const _Float16*>(a.pk_wmt), *w2et = reinterpret_cast<const _Float16*>(a.pk_w2et),
                 *w1et = reinterpret_cast<const _Float16*>(a.pk_w1et);
  _Float16 *o_gm0 = reinterpret_cast<_Float16*>(a.g_m0), *o_ggt = reinterpret_cast<_Float16*>(a.g_gt),
           *o_ggpre = reinterpret_cast<_Float16*>(a.g_gpre), *o_ghne = reinterpret_cast<_Float16*>(a.g_hne),
           *o_ghe = reinterpret_cast<_Float16*>(a.g_he), *o_gpre = reinterpret_cast<_Float16*>(a.g_pre), *o_gx = reinterpret_cast<_Float16*>(a.g_x);
  // this lane's running LayerNorm-parameter sums: four floats in each of the wave's four vectors of R (d gamma_e | d beta_e | d gamma_g | d beta_g)
  float* R = Tall + (size_t)NM_WAVES * 16 * NM_TLD;   // [NM_WAVES][NM_LNP], behind the waves' tile areas
  float* racc = R + wave * NM_LNP + (MDX_NM_LNP_DPP ? 16 * c + 4 * q : 4 * lane);
#pragma unroll
  for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(racc + 256 * k) = splat4(0.f);
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const int tile = it * nw + blockIdx.x * NM_WAVES + wave;
    const int row = 16 * tile + c;
#ifdef MDX_NM_NOSTORE_B
    const bool ok = row < E && a.f.E < 0;
#else
    const bool ok = row < E;
#endif
    const size_t r = (size_t)min(row, E - 1), ro = r * KW + 4 * q;
    const size_t rb = r * KW;
#ifdef MDX_NM_ROW0_B             // ablation (timing only): every tile reads the tape rows of tile 0 (1: cache-hot, same pattern) or row 0 (2)
    const size_t rl = MDX_NM_ROW0_B == 1 ? (size_t)c : 0, rol = rl * KW + 4 * q;
#else
    const size_t rl = r, rol = ro;
#endif
    const int64_t nc = a.f.col[rl], nr = a.row[rl];
    f32x4 y[16];
    f16x8_t b8[8];
    f32x4 gx1[4];
#define NM_SB(ft) if ((ft) % 4 == 3) __builtin_amdgcn_sched_barrier(0)
    // ---- msg = m0 * sigmoid(gt); the incoming gradient is dL/d(sum over the left node's rows), a float16 row per edge
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      uint2 h[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ft = 2 * g2 + j;
        const f32x4 g = rh4(ldg4(a.gA + (size_t)nr * a.ldga + 16 * ft + 4 * q));
        const f32x4 m0 = ldh4(s_m0 + rol + 16 * ft), sg = sigmoid4(ldh4(s_gt + rol + 16 * ft));
        h[j] = pack4(g * m0 * sg * (splat4(1.f) - sg));      // d gt
        if (ok) {
          sth4(o_gm0 + ro + 16 * ft, pack4(g * sg));        // d m0 (formed again below: the registers go to the gate chain first)
        }
      }
      sth8_pair(o_ggt + rb, g2, h[0], h[1], q, ok);         // (the backward has no registers for the LDS transposition of the forward)
      b8[g2] = pair8(h[0], h[1]);
      NM_SB(2 * g2 + 1);
    }
    // ---- gate backward
    zero<16>(y);
    mmw<16, 8>(y, wg2t, wbuf, par, tid, lane, b8);
#pragma unroll
    for (int ft = 0; ft < 16; ++ft) y[ft] = rh4(y[ft]);
    ln256_relu_bwd(y, s_gpre, s_gpre + rol, E, tile, C + 512, C + 768, q, ok, racc + 512, racc + 768, T, lane);
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      const uint2 h0 = pack4(y[2 * g2]), h1 = pack4(y[2 * g2 + 1]);
      sth8_pair(o_ggpre + rb, g2, h0, h1, q, ok);
      b8[g2] = pair8(h0, h1);
    }
    zero<4>(gx1);
    mmw<4, 8>(gx1, wg1t, wbuf, par, tid, lane, b8);
    // park the gate chain's dL/dX in its output rows (float16 values: lossless) until the edge chain's half arrives: 16 registers less
    // across the msg_net / edge_net backward, which spills
    if (ok) {
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) sth4(o_gx + r * KB + 16 * ft + 4 * q, pack4(rh4(gx1[ft])));
    }
    // ---- msg_net and the product p = he * hn[col]
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      uint2 h[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ft = 2 * g2 + j;
        h[j] = *reinterpret_cast<const uint2*>(o_gm0 + ro + 16 * ft);     // d m0, written above by this lane (rows past the end: unused)
      }
      b8[g2] = pair8(h[0], h[1]);
      NM_SB(2 * g2 + 1);
    }
    zero<16>(y);
    mmw<16, 8>(y, wmt, wbuf, par, tid, lane, b8);
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      uint2 h[2], hn[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ft = 2 * g2 + j;
        const f32x4 gp = rh4(y[ft]);
        h[j] = pack4(gp * ldh4(HN + (size_t)nc * a.f.ldhn + 16 * ft + 4 * q));       // d he
        hn[j] = pack4(gp * ldh4(s_he + rol + 16 * ft));                               // per-edge d hn[col]
      }
      sth8_pair(o_ghne + rb, g2, hn[0], hn[1], q, ok);
      sth8_pair(o_ghe + rb, g2, h[0], h[1], q, ok);
      b8[g2] = pair8(h[0], h[1]);
      NM_SB(2 * g2 + 1);
    }
    // ---- edge_net backward
    zero<16>(y);
    mmw<16, 8>(y, w2et, wbuf, par, tid, lane, b8);
#pragma unroll
    for (int ft = 0; ft < 16; ++ft) y[ft] = rh4(y[ft]);
    ln256_relu_bwd(y, s_hepre, s_hepre + rol, E, tile, C + 0, C + 256, q, ok, racc, racc + 256, T, lane);
#pragma unroll
    for (int g2 = 0; g2 < 8; ++g2) {
      const uint2 h0 = pack4(y[2 * g2]), h1 = pack4(y[2 * g2 + 1]);
      sth8_pair(o_gpre + rb, g2, h0, h1, q, ok);
      b8[g2] = pair8(h0, h1);
    }
    f32x4 gx2[4];
    zero<4>(gx2);
    mmw<4, 8>(gx2, w1et, wbuf, par, tid, lane, b8);
    if (ok) {
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) sth4(o_gx + r * KB + 16 * ft + 4 * q, pack4(rh4(gx2[ft]) + ldh4(o_gx + r * KB + 16 * ft + 4 * q)));
    }
  }
  // ---- LayerNorm-parameter gradients: every wave's row of R is complete; waves are added in fixed order
  __syncthreads();
  for (int i = tid; i < NM_LNP; i += NM_THREADS) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NM_WAVES; ++w) s += R[w * NM_LNP + i];
    a.lnp[(size_t)blockIdx.x * NM_LNP + i] = s;
  }
}

constexpr int NM_BWD_LDS = 2 * 16384 + (1024 + NM_WAVES * 16 * NM_TLD + NM_WAVES * NM_LNP) * 4;   // weights, constants, tile areas, parameter sums: 135.6 KiB
static bool g_attr_nb = false;

extern "C" int mdx_op_nodemsg_lnp_floats(void) { return NM_LNP; }

extern "C" int mdx_op_pack_a(const mdx_pack_jobs* jobs, void* stream) {
  if (!jobs || jobs->n < 0 || jobs->n > 10) return mdx_set_error(MDX_ERR_ARG, "pack_a: bad job table");
  if (jobs->n == 0) return MDX_OK;
  for (int i = 0; i < jobs->n; ++i) {
    const mdx_pack_job& j = jobs->job[i];
    if (!j.W || !j.out || j.n_out % 16 || j.n_in % 32 || j.n_out <= 0 || j.n_in <= 0) return mdx_set_error(MDX_ERR_ARG, "pack_a: widths must be multiples of 16 / 32");
  }
  hipLaunchKernelGGL(pack_a_kernel, dim3(64, jobs->n), dim3(256), 0, (hipStream_t)stream, *jobs);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "pack_a: launch failed");
}

static int check_nodemsg(const mdx_nodemsg_args& a) {
  if (a.E < 0) return mdx_set_error(MDX_ERR_ARG, "nodemsg: negative row count");
  if (a.E >= ((int64_t)1 << 23)) return mdx_set_error(MDX_ERR_UNSUPPORTED, "nodemsg: more than 2^23 rows in one launch (32-bit buffer offsets)");
  if (!a.X || !a.HN || !a.PN || !a.col || !a.pk_w1e || !a.pk_w2e || !a.pk_wm || !a.pk_wg1 || !a.pk_wg2 || !a.b1e || !a.lng_e || !a.lnb_e || !a.b2e ||
      !a.bm || !a.bg1 || !a.lng_g || !a.lnb_g || !a.bg2)
    return mdx_set_error(MDX_ERR_ARG, "nodemsg: null operand");
  if ((a.ldx & 7) || (reinterpret_cast<uintptr_t>(a.X) & 15)) return mdx_set_error(MDX_ERR_ARG, "nodemsg: X rows must be 16-byte aligned");
  if ((a.ldhn & 3) || (reinterpret_cast<uintptr_t>(a.HN) & 7) || (a.ldpn & 3) || (reinterpret_cast<uintptr_t>(a.PN) & 15))
    return mdx_set_error(MDX_ERR_ARG, "nodemsg: node rows must be vector-aligned");
  return MDX_OK;
}

extern "C" int mdx_op_nodemsg_fwd(const mdx_nodemsg_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "nodemsg_fwd: null argument block");
  if (a->E == 0) return MDX_OK;
  if (int rc = check_nodemsg(*a)) return rc;
  if (!a->he_pre || !a->he_post || !a->he || !a->p || !a->m0 || !a->g_pre || !a->g_post || !a->gt || !a->msg) return mdx_set_error(MDX_ERR_ARG, "nodemsg_fwd: null output");
  const int ntiles = (int)((a->E + 15) / 16);
  const int grid = std::max(1, std::min(NMF_WG_PER_CU * ncus(), (ntiles + NMF_WAVES - 1) / NMF_WAVES));
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)nodemsg_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMF_WAVES * TS_BYTES) != hipSuccess)
      return mdx_set_error(MDX_ERR_HIP, "nodemsg_fwd: cannot reserve LDS");
    attr = true;
  }
  hipLaunchKernelGGL(nodemsg_fwd_kernel, dim3(grid), dim3(NMF_THREADS), NMF_WAVES * TS_BYTES, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "nodemsg_fwd: launch failed");
}

extern "C" int mdx_op_nodemsg_bwd(const mdx_nodemsg_bwd_args* a, void* stream) {
  if (!a) return mdx_set_error(MDX_ERR_ARG, "nodemsg_bwd: null argument block");
  if (int rc = check_nodemsg(a->f)) return rc;
  if (!a->f.he_pre || !a->f.he || !a->f.m0 || !a->f.g_pre || !a->f.gt || !a->gA || !a->row || !a->pk_wg2t || !a->pk_wg1t || !a->pk_wmt || !a->pk_w2et ||
      !a->pk_w1et || !a->g_m0 || !a->g_gt || !a->g_gpre || !a->g_hne || !a->g_he || !a->g_pre || !a->g_x || !a->lnp)
    return mdx_set_error(MDX_ERR_ARG, "nodemsg_bwd: null operand");
  if ((a->ldga & 3) || (reinterpret_cast<uintptr_t>(a->gA) & 15)) return mdx_set_error(MDX_ERR_ARG, "nodemsg_bwd: gA rows must be 16-byte aligned");
  if (!g_attr_nb) {
    if (hipFuncSetAttribute((const void*)nodemsg_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NM_BWD_LDS) != hipSuccess)
      return mdx_set_error(MDX_ERR_HIP, "nodemsg_bwd: cannot reserve LDS");
    g_attr_nb = true;
  }
  hipLaunchKernelGGL(nodemsg_bwd_kernel, dim3(ncus()), dim3(NM_THREADS), NM_BWD_LDS, (hipStream_t)stream, *a);
  return hipGetLastError() == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, "nodemsg_bwd: launch failed");
}
