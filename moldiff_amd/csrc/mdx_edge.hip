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
#include "../../include/moldiff_hip.h"
int mdx_set_error(int code, const char* msg);

// Phase trace (tools/trace_edge_a.py; build with `make EXTRA=-DMDX_TRACE`): thread 0 of every workgroup stamps the
// shader clock at each phase boundary of edge_a into a 32-slot record.  Compiled out of the shipped library.
#ifdef MDX_TRACE
__device__ unsigned long long* mdx_trace_buf = nullptr;
__device__ int mdx_trace_sel = 0;  // 0: edge_a, 1: edge_b
extern "C" int mdx_debug_set_trace(void* p, int which) {
  hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace_sel), &which, sizeof(which));
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace_buf), &p, sizeof(p));
}
#define MDX_STAMP_K(k, i)                                                                                   \
  do {                                                                                                      \
    if (threadIdx.x == 0 && mdx_trace_buf && mdx_trace_sel == (k))                                          \
      mdx_trace_buf[(size_t)blockIdx.x * 32 + (i)] = ((i) >= 29) ? wall_clock64() : clock64();              \
  } while (0)
#else
#define MDX_STAMP_K(k, i) ((void)0)
#endif
#define MDX_STAMP(i) MDX_STAMP_K(0, i)
#define MDX_STAMPB(i) MDX_STAMP_K(1, i)

namespace {

constexpr int ET = MDX_ET;
constexpr int TE = 16 * ET;
constexpr int LD64 = mdx_ld(64);    // 72
constexpr int LD80 = mdx_ld(80);    // 88
constexpr int LD256 = mdx_ld(256);  // 264

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
  MDX_STAMP(30);
  tile_indices(a.l, a.r, a.te, e0, E, lane, li, ri, tt, valid);

  // Requested ahead of the first barrier so the latency overlaps the tile load: the edge_embs weight slice (register
  // resident, 5 fragments per wave) and the NodeBlock gate's per-node term b + gx[r] + t*wt (its accumulator's start).
  f32x4 wemb[5][1], ginit[4][ET];
  f32x4 wpre[4];  // first weight fragments of the next 256-wide layer (load_w0)
  if (a.flags & EA_EMB) load_wfrag<1, 80>(wemb, a.w.Wemb, 4, wave, lane);
  if (do_node) {
    load_w0<4>(wpre, a.w.Wg1e, 4 * wave, lane);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const int f = 16 * (4 * wave + ft) + 4 * q;
      const f32x4 b = ldg4(a.w.bg1 + f), wt = ldg4(a.w.wtg1 + f);
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        // the NodeBlock gate is fed node_time[col], the BondFFN gates edge_time
        const float tg = (a.tn_r && valid[et]) ? a.tn_r[e0 + 16 * et + c] : tt[et];
        ginit[ft][et] = b + ldg4(a.NT + (size_t)ri[et] * MDX_NTW + MDX_NT_GX + f) + splat4(tg) * wt;
      }
    }
  }

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
    gemm_tile_reg<1, ET, 80>(acc, wemb, X, LD80, lane);
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
    f32x4 sg[4][ET];
    {  // gate: sigmoid(W2 relu(LN(W1e He' + gx[r] + t*wt + b1)) + b2)
      f32x4 acc[4][ET];
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et) acc[ft][et] = ginit[ft][et];
      MDX_STAMP(3);
      gemm_tile_pre<4, ET, 64>(acc, wpre, a.w.Wg1e, 16, ft0, Hep, LD64, lane);
      MDX_STAMP(4);
      load_w0<4>(wpre, a.w.Wg2, ft0, lane);
      layernorm_relu<4, ET, 4>(acc, a.w.gg, a.w.gb, ft0, red, red2, wave, lane, true);
      acc_to_lds<4, ET>(acc, X, LD256, 0, ft0, lane);
      __syncthreads();
      MDX_STAMP(5);
      acc_bias<4, ET>(acc, a.w.bg2, ft0, lane);
      gemm_tile_pre<4, ET, 256>(acc, wpre, a.w.Wg2, 16, ft0, X, LD256, lane);
      MDX_STAMP(6);
      load_w0<4>(wpre, a.w.en.W1, ft0, lane);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et) {
          sg[ft][et] = sigmoid4(acc[ft][et]);
          if (a.tSG && valid[et]) stg4(a.tSG + (size_t)(e0 + 16 * et + c) * MDX_ND + 16 * (ft0 + ft) + 4 * q, sg[ft][et]);
        }
      // no barrier: X (read by the GEMM above) is next written after the barrier inside edge_net's LayerNorm,
      // which every wave reaches only after it has left this GEMM
      MDX_STAMP(7);
    }
    {  // edge_net, * h[r], msg_net
      f32x4 acc[4][ET];
      acc_bias<4, ET>(acc, a.w.en.b1, ft0, lane);
      gemm_tile_pre<4, ET, 64>(acc, wpre, a.w.en.W1, 16, ft0, Hep, LD64, lane);
      MDX_STAMP(8);
      load_w0<4>(wpre, a.w.en.W2, ft0, lane);
      layernorm_relu<4, ET, 4>(acc, a.w.en.g, a.w.en.be, ft0, red, red2, wave, lane, true);
      acc_to_lds<4, ET>(acc, X, LD256, 0, ft0, lane);
      __syncthreads();
      MDX_STAMP(9);
      acc_bias<4, ET>(acc, a.w.en.b2, ft0, lane);
      gemm_tile_pre<4, ET, 256>(acc, wpre, a.w.en.W2, 16, ft0, X, LD256, lane);
      MDX_STAMP(10);
      load_w0<4>(wpre, a.w.Wm, ft0, lane);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < ET; ++et) {
          if (a.tHE && valid[et]) stg4(a.tHE + (size_t)(e0 + 16 * et + c) * MDX_ND + 16 * (ft0 + ft) + 4 * q, acc[ft][et]);
          acc[ft][et] = acc[ft][et] * ldg4(a.H + (size_t)ri[et] * MDX_ND + 16 * (ft0 + ft) + 4 * q);
        }
      __syncthreads();
      acc_to_lds<4, ET>(acc, X, LD256, 0, ft0, lane);
      __syncthreads();
      MDX_STAMP(11);
      acc_bias<4, ET>(acc, a.w.bm, ft0, lane);
      gemm_tile_pre<4, ET, 256>(acc, wpre, a.w.Wm, 16, ft0, X, LD256, lane);
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
      f32x4 acc[5][ET], nlv[4][ET], gxv[ET], wa[5];
      load_w0<5>(wa, a.w.Wffa, 5 * wave, lane);
      const f32x4 bg = ldg4(w.bg1 + 16 * wh + 4 * q), wt = ldg4(w.wtg1 + 16 * wh + 4 * q);
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        gxv[et] = ldg4(a.NT + (size_t)idx[et] * MDX_NTW + gxcol);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) nlv[ft][et] = ldg4(a.NT + (size_t)idx[et] * MDX_NTW + nlcol + 16 * ft + 4 * q);
      }
      acc_zero<5, ET>(acc);
      gemm_tile_pre<5, ET, 64>(acc, wa, a.w.Wffa, 20, 5 * wave, Hep, LD64, lane);
      MDX_STAMP(14);
      load_w0<4>(wpre, w.inter.W1, 4 * wh, lane);
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
    f32x4 wc[2], wd[2];
    {
      f32x4 h[4][ET];
      acc_bias<4, ET>(h, w.inter.b1, 4 * wh, lane);
      gemm_tile_pre<4, ET, 128>(h, wpre, w.inter.W1, 8, 4 * wh, X + 128 * s, LD256, lane);
      MDX_STAMP(16);
      load_w0<2>(wc, w.inter.W2, 2 * wh, lane);
      load_w0<2>(wd, w.Wg2, 2 * wh, lane);
      layernorm_relu<4, ET, 2>(h, w.inter.g, w.inter.be, 4 * wh, red, red2, wave, lane, true, true, 2 * s);
      acc_to_lds<4, ET>(h, X, LD256, 128 * s, 4 * wh, lane);  // in place: every wave is past GEMM B (LN barrier)
    }
    __syncthreads();
    MDX_STAMP(17);
    {
      f32x4 o[2][ET], g2[2][ET];
      acc_bias<2, ET>(o, w.inter.b2, 2 * wh, lane);
      gemm_tile_pre<2, ET, 128>(o, wc, w.inter.W2, 4, 2 * wh, X + 128 * s, LD256, lane);
      MDX_STAMP(18);
      acc_bias<2, ET>(g2, w.bg2, 2 * wh, lane);
      gemm_tile_pre<2, ET, 32>(g2, wd, w.Wg2, 4, 2 * wh, GG + 32 * s, LD64, lane);
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
  MDX_STAMP(29);
}

__global__ __launch_bounds__(MDX_WG, MDX_EWPS) void edge_b_kernel(const EdgeBArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hep = smem + OFF_HEP;
  float* X = smem + OFF_X;  // U (ld 72) / A (ld 72) / inter (ld 264)
  float* red = smem + OFF_RED;
  float* red2 = smem + OFF_RED2;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int e0 = tile * TE;
  const int E = a.E;
  const bool do_edge = a.flags & EB_EDGE, do_pos = a.flags & EB_POS;
  const int f = 16 * wave + 4 * q;

  int li[ET], ri[ET];
  float tt[ET];
  bool valid[ET];
  MDX_STAMPB(0);
  MDX_STAMPB(30);
  tile_indices(a.l, a.r, a.te, e0, E, lane, li, ri, tt, valid);

  // Everything that depends only on the tile's indices is requested here, ahead of the first barrier, so that its
  // latency overlaps the He' tile load: the two 64x64 weight slices (register resident, 8 fragments per wave), the
  // per-node gathers of the EdgeBlock tail, the Lf[l]*Rf[r] rows and the edge geometry of PosUpdate.
  f32x4 wself[4][1], wout[4][1], u[1][ET];
  if (do_edge) {
    load_wfrag<1, 64>(wself, a.w.Wself, 4, wave, lane);
    load_wfrag<1, 64>(wout, a.w.Wout, 4, wave, lane);
    const f32x4 bs = ldg4(a.w.bself + f);
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      f32x4 v = ldg4(a.SL + (size_t)li[et] * 64 + f) + ldg4(a.SR + (size_t)ri[et] * 64 + f);
      v = v + ldg4(a.NT + (size_t)li[et] * MDX_NTW + MDX_NT_NFL + f);
      v = v + ldg4(a.NT + (size_t)ri[et] * MDX_NTW + MDX_NT_NFR + f);
      u[0][et] = v + bs;
    }
  }
  f32x4 lfv[ET], rfv[ET];  // A-tile element (row = tid/16 + 16 j, columns 4*(tid%16)..)
  float rx[ET], ry[ET], rz[ET], dd[ET];
  if (do_pos) {
#pragma unroll
    for (int j = 0; j < ET; ++j) {
      const int e = e0 + (tid >> 4) + 16 * j;
      lfv[j] = splat4(0.f);
      rfv[j] = splat4(0.f);
      if (e < E) {
        lfv[j] = ldg4(a.Lf + (size_t)a.l[e] * 64 + 4 * (tid & 15));
        rfv[j] = ldg4(a.Rf + (size_t)a.r[e] * 64 + 4 * (tid & 15));
      }
    }
    if (wave == 0 && q == 0) {
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        const int e = e0 + 16 * et + c;
        rx[et] = ry[et] = rz[et] = 0.f;
        dd[et] = 1.f;
        if (!valid[et]) continue;
        if (a.rel_in) {
          rx[et] = a.rel_in[3 * (size_t)e + 0]; ry[et] = a.rel_in[3 * (size_t)e + 1]; rz[et] = a.rel_in[3 * (size_t)e + 2];
          dd[et] = a.dist_in[e];
        } else {
          rx[et] = a.pos[3 * li[et] + 0] - a.pos[3 * ri[et] + 0];
          ry[et] = a.pos[3 * li[et] + 1] - a.pos[3 * ri[et] + 1];
          rz[et] = a.pos[3 * li[et] + 2] - a.pos[3 * ri[et] + 2];
          dd[et] = sqrtf(rx[et] * rx[et] + ry[et] * ry[et] + rz[et] * rz[et]);
        }
      }
    }
  }
  load_rows64(a.Hep, e0, E, Hep, LD64, tid);
  __syncthreads();
  MDX_STAMPB(1);

  if (do_edge) {
    MDX_STAMPB(2);
    gemm_tile_reg<1, ET, 64>(u, wself, Hep, LD64, lane);
    MDX_STAMPB(3);
    layernorm_relu<1, ET, 4>(u, a.w.lng, a.w.lnb, wave, red, red2, wave, lane, true);
    acc_to_lds<1, ET>(u, X, LD64, 0, wave, lane);
    __syncthreads();
    MDX_STAMPB(4);
    f32x4 d[1][ET];
    acc_bias<1, ET>(d, a.w.bout, wave, lane);
    gemm_tile_reg<1, ET, 64>(d, wout, X, LD64, lane);
    MDX_STAMPB(5);
#pragma unroll
    for (int et = 0; et < ET; ++et) {
      if (!(a.flags & EB_DELTA)) d[0][et] = d[0][et] + lds4(Hep + (16 * et + c) * LD64 + f);
      if (valid[et]) stg4(a.He_out + (size_t)(e0 + 16 * et + c) * 64 + f, d[0][et]);
    }
    if (do_pos) {
      __syncthreads();  // every wave is done reading Hep (self_ffn, residual) and X (U)
      acc_to_lds<1, ET>(d, Hep, LD64, 0, wave, lane);
    }
  }
  MDX_STAMPB(6);

  if (do_pos) {
    float* A = X;  // (TE x 64, ld 72): a = Lf[l] * Rf[r]
    f32x4 wa[5], wa2[5];
    load_w0<5>(wa, a.w.WblG, 5 * wave, lane);
    load_w0<5>(wa2, a.w.WnlG, 5 * wave, lane);
#pragma unroll
    for (int j = 0; j < ET; ++j) sts4(A + ((tid >> 4) + 16 * j) * LD64 + 4 * (tid & 15), lfv[j] * rfv[j]);
    __syncthreads();
    MDX_STAMPB(7);
    // (W_bl He'') and (W_nl a), 256 features each, with the 129 -> 32 gate layer riding along as a fifth feature tile
    // of waves 0,1 (its accumulator is chained through both GEMMs: ((b + t wt) + W_h He'') + W_a a, as in the reference)
    const int ft0 = 4 * wave;
    const bool act = wave < 2;
    f32x4 acc[5][ET], acc2[5][ET], wi[4];
    acc_zero<5, ET>(acc);
    acc_zero<5, ET>(acc2);
    if (act) {
      const f32x4 b = ldg4(a.w.bg1 + f), wt = ldg4(a.w.wtg1 + f);
#pragma unroll
      for (int et = 0; et < ET; ++et) acc[4][et] = b + splat4(tt[et]) * wt;
    }
    gemm_tile_pre<5, ET, 64>(acc, wa, a.w.WblG, 20, 5 * wave, Hep, LD64, lane);
#pragma unroll
    for (int et = 0; et < ET; ++et) acc2[4][et] = acc[4][et];
    gemm_tile_pre<5, ET, 64>(acc2, wa2, a.w.WnlG, 20, 5 * wave, A, LD64, lane);
    MDX_STAMPB(8);
    load_w0<4>(wi, a.w.Wi1, ft0, lane);
    float gate[ET];
    {
      f32x4 g1[1][ET];
#pragma unroll
      for (int et = 0; et < ET; ++et) g1[0][et] = acc2[4][et];
      // (the barrier inside also tells every wave that A, which aliases X, has been consumed)
      layernorm_relu<1, ET, 2>(g1, a.w.gg, a.w.gb, wave, red, red2, wave, lane, act);
      dot_rows<1, ET, 2>(g1, a.w.wg2, wave, smem + OFF_RED3, wave, lane, act, gate);
    }
    MDX_STAMPB(9);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int et = 0; et < ET; ++et)
        sts4(X + (16 * et + c) * LD256 + 16 * (ft0 + ft) + 4 * q, acc[ft][et] * acc2[ft][et]);
    MDX_STAMPB(10);
    __syncthreads();
    MDX_STAMPB(11);
    f32x4 h[4][ET];
    acc_bias<4, ET>(h, a.w.bi1, ft0, lane);
    gemm_tile_pre<4, ET, 256>(h, wi, a.w.Wi1, 16, ft0, X, LD256, lane);
    MDX_STAMPB(12);
    layernorm_relu<4, ET, 4>(h, a.w.ig, a.w.ib, ft0, red, red2, wave, lane, true);
    float wd[ET];
    dot_rows<4, ET, 4>(h, a.w.wi2, ft0, smem + OFF_RED3, wave, lane, true, wd);
    MDX_STAMPB(13);
    if (wave == 0 && q == 0) {
#pragma unroll
      for (int et = 0; et < ET; ++et) {
        if (!valid[et]) continue;
        const int e = e0 + 16 * et + c;
        const float w = (wd[et] + a.w.bi2) * sigmoidf_(gate[et] + a.w.bg2);
        const float d = dd[et], dp = d + 1.0f;
        a.Fe[3 * (size_t)e + 0] = w * rx[et] / d / dp;
        a.Fe[3 * (size_t)e + 1] = w * ry[et] / d / dp;
        a.Fe[3 * (size_t)e + 2] = w * rz[et] / d / dp;
      }
    }
  }
  MDX_STAMPB(14);
  MDX_STAMPB(29);
}

}  // namespace

static bool g_attr_set = false;
static void ensure_attr() {
  if (g_attr_set) return;
  hipFuncSetAttribute((const void*)edge_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
  hipFuncSetAttribute((const void*)edge_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
  g_attr_set = true;
}

int launch_edge_a(const EdgeAArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  if (a.flags & EA_SPLIT) return launch_edge_a2s(a, s);
  if (mdx_use_rowowner()) return launch_edge_a2(a, s);
  if (a.flags & EA_AGG) return mdx_set_error(MDX_ERR_UNSUPPORTED, "the tile kernels have no in-kernel aggregation (EA_AGG)");
  ensure_attr();
  const int ntiles = (a.E + TE - 1) / TE;
  hipLaunchKernelGGL(edge_a_kernel, dim3(ntiles), dim3(MDX_WG), LDS_FLOATS * 4, s, a, ntiles);
  return MDX_OK;
}

int launch_edge_b(const EdgeBArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  if (a.flags & EB_SPLIT) return launch_edge_b2s(a, s);
  if (mdx_use_rowowner()) return launch_edge_b2(a, s);
  ensure_attr();
  const int ntiles = (a.E + TE - 1) / TE;
  hipLaunchKernelGGL(edge_b_kernel, dim3(ntiles), dim3(MDX_WG), LDS_FLOATS * 4, s, a, ntiles);
  return MDX_OK;
}
