// Fused per-edge kernels of one NodeEdgeNet block (gfx950).
//
//   edge kernel A  (reference models/graph.py:352-357 edge_embs, :42-47 NodeBlock message path,
//                   :133-141/:278,:282 the two EdgeBlock BondFFNs):
//       He' = edge_embs([He | smear(|pos_l - pos_r|)])
//       M   = msg_net(edge_net(He') * h[r]) * sigmoid(gate([He' | x[r] | t]))          -> (E,256)
//       F_s = inter_s((W_bl He') * nl_s[idx_s]) * sigmoid(gate_s([He' | x[idx_s] | t]))  -> (E,64), s = left,right
//   edge kernel B  (models/graph.py:286-294 EdgeBlock tail, :384-393 PosUpdate):
//       He'' = He' + out_transform(relu(LN(SL[l] + SR[r] + nfl[l] + nfr[r] + self_ffn(He'))))
//       w    = inter((W_bl He'') * (W_nl a)) * sigmoid(gate([He'' | a | t])),  a = Lf[l] * Rf[r]
//       Fe   = w * rel / d / (d + 1)
//
// One workgroup (4 waves) owns 16*MDX_ET consecutive edges of the (left,right)-sorted edge list; all
// activations of the tile stay in LDS between layers, weights stream L2 -> VGPR in packed fragment
// order, all contractions run on v_mfma_f32_16x16x4_f32.  Per-node terms (everything the reference
// computes as Linear(h_node[idx])) are gathered from the hoisted node tables H / NT.
#include "mdx_kernels.h"
#include "mdx_tile.h"

// Phase trace (tools/trace_edge_a.py; build with `make EXTRA=-DMDX_TRACE`): thread 0 of every workgroup stamps the
// shader clock at each phase boundary of edge_a into a 32-slot record.  Compiled out of the shipped library.
#ifdef MDX_TRACE
__device__ unsigned long long* mdx_trace_buf = nullptr;
extern "C" int mdx_debug_set_trace(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace_buf), &p, sizeof(p));
}
#define MDX_STAMP(i)                                                                            \
  do {                                                                                          \
    if (threadIdx.x == 0 && mdx_trace_buf) mdx_trace_buf[(size_t)blockIdx.x * 32 + (i)] = clock64(); \
  } while (0)
#else
#define MDX_STAMP(i) ((void)0)
#endif

namespace {

constexpr int ET = MDX_ET;
constexpr int TE = 16 * ET;
constexpr int LD64 = mdx_ld(64);    // 72
constexpr int LD80 = mdx_ld(80);    // 88
constexpr int LD256 = mdx_ld(256);  // 264
constexpr int LD32 = mdx_ld(32);    // 40

// LDS carve (floats)
constexpr int OFF_HEP = 0;
constexpr int OFF_X = OFF_HEP + TE * LD64;
constexpr int OFF_GG = OFF_X + TE * LD256;
constexpr int OFF_RED = OFF_GG + TE * LD64;  // GG: both BondFFN gate hiddens side by side (2 x 32)
constexpr int OFF_RED2 = OFF_RED + 4 * TE;
constexpr int OFF_RED3 = OFF_RED2 + 4 * TE;  // dot_rows scratch (must not alias the LayerNorm buffers)
constexpr int LDS_FLOATS = OFF_RED3 + 4 * TE;

__device__ __forceinline__ void tile_indices(const int* __restrict__ l, const int* __restrict__ r,
                                             const float* __restrict__ te, int e0, int E, int lane, int (&li)[ET],
                                             int (&ri)[ET], float (&tt)[ET], bool (&valid)[ET]) {
  const int c = lane & 15;
#pragma unroll
  for (int et = 0; et < ET; ++et) {
    const int e = e0 + 16 * et + c;
    valid[et] = e < E;
    li[et] = valid[et] ? l[e] : 0;
    ri[et] = valid[et] ? r[e] : 0;
    tt[et] = valid[et] ? te[e] : 0.f;
  }
}

// rows of a (rows x 64) global array -> LDS tile (ld), zero fill past E
__device__ __forceinline__ void load_rows64(const float* __restrict__ src, int e0, int E, float* dst, int ld, int tid) {
  for (int i = tid; i < TE * 16; i += MDX_WG) {
    const int row = i >> 4, c4 = i & 15;
    const int e = e0 + row;
    f32x4 v = (e < E) ? ldg4(src + (size_t)e * 64 + 4 * c4) : splat4(0.f);
    sts4(dst + row * ld + 4 * c4, v);
  }
}

__global__ __launch_bounds__(MDX_WG, MDX_EWPS) void edge_a_kernel(const EdgeAArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hep = smem + OFF_HEP;
  float* X = smem + OFF_X;
  float* GG = smem + OFF_GG;
  float* red = smem + OFF_RED;
  float* red2 = smem + OFF_RED2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int e0 = tile * TE;
  const int E = a.E;
  const bool do_node = a.flags & EA_NODE;
  const bool do_ffn = a.flags & EA_FFN;

  int li[ET], ri[ET];
  float tt[ET];
  bool valid[ET];
  MDX_STAMP(0);
#ifdef MDX_TRACE
  if (threadIdx.x == 0 && mdx_trace_buf) {
    mdx_trace_buf[(size_t)blockIdx.x * 32 + 30] = wall_clock64();
    mdx_trace_buf[(size_t)blockIdx.x * 32 + 31] =
        ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
  tile_indices(a.l, a.r, a.te, e0, E, lane, li, ri, tt, valid);

  // ---- phase A/B: He' ---------------------------------------------------------------------
  if (a.flags & EA_EMB) {
    load_rows64(a.He_in, e0, E, X, LD80, tid);
    {  // Gaussian smearing of the edge length: thread -> (row = tid/16 + 16 j, gaussian k = tid%16)
      static_assert(MDX_NG == 16 && MDX_WG == 256, "smearing thread map");
      const int k = tid & 15;
      const float off = a.soff[k], coef = a.scoef[k];
#pragma unroll
      for (int j = 0; j < ET; ++j) {
        const int row = (tid >> 4) + 16 * j, e = e0 + row;
        float d = 0.f;
        if (e < E) {
          if (a.dist_in) {
            d = a.dist_in[e];
          } else {
            const int nl = a.l[e], nr = a.r[e];
            const float dx = a.pos[3 * nl + 0] - a.pos[3 * nr + 0];
            const float dy = a.pos[3 * nl + 1] - a.pos[3 * nr + 1];
            const float dz = a.pos[3 * nl + 2] - a.pos[3 * nr + 2];
            d = sqrtf(dx * dx + dy * dy + dz * dz);
          }
        }
        const float u = fminf(fmaxf(d, 0.f), a.cutoff) - off;
        X[row * LD80 + 64 + k] = expf(coef * (u * u));
      }
    }
    __syncthreads();
    MDX_STAMP(1);
    f32x4 acc[1][ET];
    acc_bias<1, ET>(acc, a.w.bemb, wave, lane);
    gemm_tile<1, ET, 80>(acc, a.w.Wemb, 4, wave, X, LD80, lane);
    acc_to_lds<1, ET>(acc, Hep, LD64, 0, wave, lane);
#pragma unroll
    for (int et = 0; et < ET; ++et)
      if (valid[et]) stg4(a.He_out + (size_t)(e0 + 16 * et + c) * 64 + 16 * wave + 4 * q, acc[0][et]);
  } else {
    load_rows64(a.He_in, e0, E, Hep, LD64, tid);
  }
  __syncthreads();
  MDX_STAMP(2);

  // (Measured, round 1: alternating the order of the two sections between co-resident workgroups to break a
  // suspected lock-step did not help -- 100.5 vs 101.7 TFLOP/s -- so the sections run in program order.)
  // ---- NodeBlock message path -----------------------------------------------------------------
  if (do_node) {
    const int ft0 = 4 * wave;
    float tg[ET];  // the NodeBlock gate is fed node_time[col], the BondFFN gates edge_time
#pragma unroll
    for (int et = 0; et < ET; ++et) tg[et] = (a.tn_r && valid[et]) ? a.tn_r[e0 + 16 * et + c] : tt[et];
    f32x4 sg[4][ET];
    {  // gate: sigmoid(W2 relu(LN(W1e He' + gx[r] + t*wt + b1)) + b2)
      f32x4 acc[4][ET];
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const int f = 16 * (ft0 + ft) + 4 * q;
        const f32x4 b = ldg4(a.w.bg1 + f), wt = ldg4(a.w.wtg1 + f);
#pragma unroll
        for (int et = 0; et < ET; ++et)
          acc[ft][et] = b + ldg4(a.NT + (size_t)ri[et] * MDX_NTW + MDX_NT_GX + f) + splat4(tg[et]) * wt;
      }
      MDX_STAMP(3);
      gemm_tile<4, ET, 64>(acc, a.w.Wg1e, 16, ft0, Hep, LD64, lane);
      MDX_STAMP(4);
      layernorm_relu<4, ET, 4>(acc, a.w.gg, a.w.gb, ft0, red, red2, wave, lane, true);
      acc_to_lds<4, ET>(acc, X, LD256, 0, ft0, lane);
      __syncthreads();
      MDX_STAMP(5);
      acc_bias<4, ET>(acc, a.w.bg2, ft0, lane);
      gemm_tile<4, ET, 256>(acc, a.w.Wg2, 16, ft0, X, LD256, lane);
      MDX_STAMP(6);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et) sg[ft][et] = sigmoid4(acc[ft][et]);
      // no barrier: X (read by the GEMM above) is next written after the barrier inside edge_net's LayerNorm,
      // which every wave reaches only after it has left this GEMM
      MDX_STAMP(7);
    }
    {  // edge_net, * h[r], msg_net
      f32x4 acc[4][ET];
      acc_bias<4, ET>(acc, a.w.en.b1, ft0, lane);
      gemm_tile<4, ET, 64>(acc, a.w.en.W1, 16, ft0, Hep, LD64, lane);
      MDX_STAMP(8);
      layernorm_relu<4, ET, 4>(acc, a.w.en.g, a.w.en.be, ft0, red, red2, wave, lane, true);
      acc_to_lds<4, ET>(acc, X, LD256, 0, ft0, lane);
      __syncthreads();
      MDX_STAMP(9);
      acc_bias<4, ET>(acc, a.w.en.b2, ft0, lane);
      gemm_tile<4, ET, 256>(acc, a.w.en.W2, 16, ft0, X, LD256, lane);
      MDX_STAMP(10);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et)
          acc[ft][et] = acc[ft][et] * ldg4(a.H + (size_t)ri[et] * MDX_ND + 16 * (ft0 + ft) + 4 * q);
      __syncthreads();
      acc_to_lds<4, ET>(acc, X, LD256, 0, ft0, lane);
      __syncthreads();
      MDX_STAMP(11);
      acc_bias<4, ET>(acc, a.w.bm, ft0, lane);
      gemm_tile<4, ET, 256>(acc, a.w.Wm, 16, ft0, X, LD256, lane);
      MDX_STAMP(12);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et)
          if (valid[et])
            stg4(a.M + (size_t)(e0 + 16 * et + c) * MDX_ND + 16 * (ft0 + ft) + 4 * q, acc[ft][et] * sg[ft][et]);
      MDX_STAMP(13);  // no barrier: the BondFFN section writes X only after the barrier inside its gate LayerNorm
    }
  }

  // ---- EdgeBlock BondFFNs, both sides at once: waves 0,1 own the left FFN (node = l), waves 2,3 the right (node = r) ----
  //   GEMM A  He' -> [bond_linear_s (64 of 128 features) | gate layer 1 (16 of 32)] per wave, one fused 64 -> 320 pack
  //   GEMM B  inter layer 1 (128 -> 128 per side, 64 per wave), GEMM C inter layer 2 (128 -> 64), GEMM D gate layer 2
  if (do_ffn) {
    const int s = __builtin_amdgcn_readfirstlane(wave >> 1), wh = __builtin_amdgcn_readfirstlane(wave & 1);
    const FfnW& w = a.w.ffn[s];
    const int nlcol = (s ? MDX_NT_NLR : MDX_NT_NLL) + 64 * wh;
    const int gxcol = (s ? MDX_NT_GXR : MDX_NT_GXL) + 16 * wh + 4 * q;
    int idx[ET];
#pragma unroll
    for (int et = 0; et < ET; ++et) idx[et] = s ? ri[et] : li[et];
    {
      f32x4 acc[5][ET], nlv[4][ET], gxv[ET];
      const f32x4 bg = ldg4(w.bg1 + 16 * wh + 4 * q), wt = ldg4(w.wtg1 + 16 * wh + 4 * q);
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        gxv[et] = ldg4(a.NT + (size_t)idx[et] * MDX_NTW + gxcol);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) nlv[ft][et] = ldg4(a.NT + (size_t)idx[et] * MDX_NTW + nlcol + 16 * ft + 4 * q);
      }
      acc_zero<5, ET>(acc);
      gemm_tile<5, ET, 64>(acc, a.w.Wffa, 20, 5 * wave, Hep, LD64, lane);
      MDX_STAMP(14);
      f32x4 g1[1][ET];
#pragma unroll
      for (int et = 0; et < ET; ++et) g1[0][et] = ((acc[4][et] + bg) + gxv[et]) + splat4(tt[et]) * wt;
      // (the barrier inside also orders the X writes below after every wave's msg_net GEMM reads)
      layernorm_relu<1, ET, 2>(g1, w.gg, w.gb, wh, red, red2, wave, lane, true, true, 2 * s);
      acc_to_lds<1, ET>(g1, GG, LD64, 0, wave, lane);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et)
          sts4(X + (16 * et + c) * LD256 + 64 * wave + 16 * ft + 4 * q, acc[ft][et] * nlv[ft][et]);
    }
    __syncthreads();
    MDX_STAMP(15);
    {
      f32x4 h[4][ET];
      acc_bias<4, ET>(h, w.inter.b1, 4 * wh, lane);
      gemm_tile<4, ET, 128>(h, w.inter.W1, 8, 4 * wh, X + 128 * s, LD256, lane);
      MDX_STAMP(16);
      layernorm_relu<4, ET, 2>(h, w.inter.g, w.inter.be, 4 * wh, red, red2, wave, lane, true, true, 2 * s);
      acc_to_lds<4, ET>(h, X, LD256, 128 * s, 4 * wh, lane);  // in place: every wave is past GEMM B (LN barrier)
    }
    __syncthreads();
    MDX_STAMP(17);
    {
      f32x4 o[2][ET], g2[2][ET];
      acc_bias<2, ET>(o, w.inter.b2, 2 * wh, lane);
      gemm_tile<2, ET, 128>(o, w.inter.W2, 4, 2 * wh, X + 128 * s, LD256, lane);
      MDX_STAMP(18);
      acc_bias<2, ET>(g2, w.bg2, 2 * wh, lane);
      gemm_tile<2, ET, 32>(g2, w.Wg2, 4, 2 * wh, GG + 32 * s, LD64, lane);
      MDX_STAMP(19);
      float* F = a.F[s];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et)
          if (valid[et])
            stg4(F + (size_t)(e0 + 16 * et + c) * 64 + 32 * wh + 16 * ft + 4 * q, o[ft][et] * sigmoid4(g2[ft][et]));
    }
    MDX_STAMP(20);
  }
#ifdef MDX_TRACE
  if (threadIdx.x == 0 && mdx_trace_buf) mdx_trace_buf[(size_t)blockIdx.x * 32 + 29] = wall_clock64();
#endif
}

__global__ __launch_bounds__(MDX_WG, MDX_EWPS) void edge_b_kernel(const EdgeBArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hep = smem + OFF_HEP;
  float* X = smem + OFF_X;   // U (ld 72) / A (ld 72) / inter (ld 264)
  float* GG = smem + OFF_GG;
  float* red = smem + OFF_RED;
  float* red2 = smem + OFF_RED2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int e0 = tile * TE;
  const int E = a.E;

  int li[ET], ri[ET];
  float tt[ET];
  bool valid[ET];
  tile_indices(a.l, a.r, a.te, e0, E, lane, li, ri, tt, valid);
  load_rows64(a.Hep, e0, E, Hep, LD64, tid);
  __syncthreads();

  if (a.flags & EB_EDGE) {
    const int f = 16 * wave + 4 * q;
    f32x4 u[1][ET];
    const f32x4 bs = ldg4(a.w.bself + f);
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      f32x4 v = ldg4(a.SL + (size_t)li[et] * 64 + f) + ldg4(a.SR + (size_t)ri[et] * 64 + f);
      v = v + ldg4(a.NT + (size_t)li[et] * MDX_NTW + MDX_NT_NFL + f);
      v = v + ldg4(a.NT + (size_t)ri[et] * MDX_NTW + MDX_NT_NFR + f);
      u[0][et] = v + bs;
    }
    gemm_tile<1, ET, 64>(u, a.w.Wself, 4, wave, Hep, LD64, lane);
    layernorm_relu<1, ET, 4>(u, a.w.lng, a.w.lnb, wave, red, red2, wave, lane, true);
    acc_to_lds<1, ET>(u, X, LD64, 0, wave, lane);
    __syncthreads();
    f32x4 d[1][ET];
    acc_bias<1, ET>(d, a.w.bout, wave, lane);
    gemm_tile<1, ET, 64>(d, a.w.Wout, 4, wave, X, LD64, lane);
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      if (!(a.flags & EB_DELTA)) d[0][et] = d[0][et] + lds4(Hep + (16 * et + c) * LD64 + f);
      if (valid[et]) stg4(a.He_out + (size_t)(e0 + 16 * et + c) * 64 + f, d[0][et]);
    }
    __syncthreads();  // every wave is done reading Hep (self_ffn) and X (U)
    acc_to_lds<1, ET>(d, Hep, LD64, 0, wave, lane);
    __syncthreads();
  }

  if (a.flags & EB_POS) {
    float* A = X;  // (TE x 64, ld 72): a = Lf[l] * Rf[r]
    for (int i = tid; i < TE * 16; i += MDX_WG) {
      const int row = i >> 4, c4 = i & 15;
      const int e = e0 + row;
      f32x4 v = splat4(0.f);
      if (e < E) v = ldg4(a.Lf + (size_t)a.l[e] * 64 + 4 * c4) * ldg4(a.Rf + (size_t)a.r[e] * 64 + 4 * c4);
      sts4(A + row * LD64 + 4 * c4, v);
    }
    __syncthreads();
    // gate: 129 -> 32 -> 1
    float gate[ET];
    {
      f32x4 g1[1][ET];
      const bool act = wave < 2;
      if (act) {
        const int f = 16 * wave + 4 * q;
        const f32x4 b = ldg4(a.w.bg1 + f), wt = ldg4(a.w.wtg1 + f);
#pragma unroll
        for (int et = 0; et < ET; ++et) g1[0][et] = b + splat4(tt[et]) * wt;
        gemm_tile<1, ET, 64>(g1, a.w.Wg1h, 2, wave, Hep, LD64, lane);
        gemm_tile<1, ET, 64>(g1, a.w.Wg1a, 2, wave, A, LD64, lane);
      } else {
        acc_zero<1, ET>(g1);
      }
      layernorm_relu<1, ET, 2>(g1, a.w.gg, a.w.gb, wave, red, red2, wave, lane, act);
      dot_rows<1, ET, 2>(g1, a.w.wg2, wave, smem + OFF_RED3, wave, lane, act, gate);
    }
    // inter: (W_bl He'') * (W_nl a) -> 256 -> LN/ReLU -> 1
    const int ft0 = 4 * wave;
    f32x4 acc[4][ET];
    {
      f32x4 acc2[4][ET];
      acc_zero<4, ET>(acc);
      acc_zero<4, ET>(acc2);
      gemm_tile<4, ET, 64>(acc, a.w.Wbl, 16, ft0, Hep, LD64, lane);
      gemm_tile<4, ET, 64>(acc2, a.w.Wnl, 16, ft0, A, LD64, lane);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et) acc[ft][et] = acc[ft][et] * acc2[ft][et];
    }
    __syncthreads();  // A (aliases X) fully consumed
    acc_to_lds<4, ET>(acc, X, LD256, 0, ft0, lane);
    __syncthreads();
    acc_bias<4, ET>(acc, a.w.bi1, ft0, lane);
    gemm_tile<4, ET, 256>(acc, a.w.Wi1, 16, ft0, X, LD256, lane);
    layernorm_relu<4, ET, 4>(acc, a.w.ig, a.w.ib, ft0, red, red2, wave, lane, true);
    float wd[ET];
    dot_rows<4, ET, 4>(acc, a.w.wi2, ft0, smem + OFF_RED3, wave, lane, true, wd);
    if (wave == 0 && q == 0) {
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        if (!valid[et]) continue;
        const int e = e0 + 16 * et + c;
        const float w = (wd[et] + a.w.bi2) * sigmoidf_(gate[et] + a.w.bg2);
        float rx, ry, rz, d;
        if (a.rel_in) {
          rx = a.rel_in[3 * (size_t)e + 0]; ry = a.rel_in[3 * (size_t)e + 1]; rz = a.rel_in[3 * (size_t)e + 2];
          d = a.dist_in[e];
        } else {
          rx = a.pos[3 * li[et] + 0] - a.pos[3 * ri[et] + 0];
          ry = a.pos[3 * li[et] + 1] - a.pos[3 * ri[et] + 1];
          rz = a.pos[3 * li[et] + 2] - a.pos[3 * ri[et] + 2];
          d = sqrtf(rx * rx + ry * ry + rz * rz);
        }
        const float dp = d + 1.0f;
        a.Fe[3 * (size_t)e + 0] = w * rx / d / dp;
        a.Fe[3 * (size_t)e + 1] = w * ry / d / dp;
        a.Fe[3 * (size_t)e + 2] = w * rz / d / dp;
      }
    }
  }
}

}  // namespace

static bool g_attr_set = false;
static void ensure_attr() {
  if (g_attr_set) return;
  hipFuncSetAttribute((const void*)edge_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
  hipFuncSetAttribute((const void*)edge_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
  g_attr_set = true;
}

void launch_edge_a(const EdgeAArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  ensure_attr();
  const int ntiles = (a.E + TE - 1) / TE;
  hipLaunchKernelGGL(edge_a_kernel, dim3(ntiles), dim3(MDX_WG), LDS_FLOATS * 4, s, a, ntiles);
}

void launch_edge_b(const EdgeBArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  ensure_attr();
  const int ntiles = (a.E + TE - 1) / TE;
  hipLaunchKernelGGL(edge_b_kernel, dim3(ntiles), dim3(MDX_WG), LDS_FLOATS * 4, s, a, ntiles);
}
