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

namespace {

constexpr int LD64 = mdx_ld(64);
constexpr int LD256 = mdx_ld(256);
constexpr int LD32 = mdx_ld(32);
constexpr int LD16 = mdx_ld(16);

// =================================================================================================
// B1: EdgeBlock tail backward.   He_{i+1} = He' + Wout relu(LN(u)) + b,
//      u = SL[l] + SR[r] + nfl[l] + nfr[r] + Wself He' + b
//   in : gHe = dL/dHe_{i+1};  out: GU = dL/du,  GHEP = gHe + Wself^T GU   (partial dL/dHe')
// =================================================================================================
constexpr int TET = MDX_ET;
constexpr int TTE = 16 * TET;
constexpr int T_HEP = 0;
constexpr int T_G = T_HEP + TTE * LD64;
constexpr int T_X = T_G + TTE * LD64;
constexpr int T_RED = T_X + TTE * LD64;
constexpr int T_TOTAL = T_RED + 16 * TTE;

__device__ __forceinline__ void load_rows64_t(const float* __restrict__ src, int e0, int E, float* dst, int ld, int tid,
                                              int TE) {
  for (int i = tid; i < TE * 16; i += MDX_WG) {
    const int row = i >> 4, c4 = i & 15;
    const int e = e0 + row;
    sts4(dst + row * ld + 4 * c4, (e < E) ? ldg4(src + (size_t)e * 64 + 4 * c4) : splat4(0.f));
  }
}

__global__ __launch_bounds__(MDX_WG, 2) void edge_tail_bwd_kernel(const EdgeTailBwdArgs a, const int ntiles) {
  __shared__ __attribute__((aligned(16))) float smem[T_TOTAL];
  float* Hep = smem + T_HEP;
  float* G = smem + T_G;
  float* X = smem + T_X;
  float* red = smem + T_RED;
  float *red2 = red + 4 * TTE, *red3 = red + 8 * TTE, *red4 = red + 12 * TTE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int e0 = tile * TTE, E = a.E;
  const int f = 16 * wave + 4 * q;
  int li[TET], ri[TET];
  bool valid[TET];
#pragma unroll
  for (int et = 0; et < TET; ++et) {
    const int e = e0 + 16 * et + c;
    valid[et] = e < E;
    li[et] = valid[et] ? a.l[e] : 0;
    ri[et] = valid[et] ? a.r[e] : 0;
  }
  load_rows64_t(a.Hep, e0, E, Hep, LD64, tid, TTE);
  load_rows64_t(a.gHe, e0, E, G, LD64, tid, TTE);
  __syncthreads();
  // recompute u -> x_hat
  f32x4 u[1][TET];
  const f32x4 bs = ldg4(a.w.bself + f);
#pragma unroll
  for (int et = 0; et < TET; ++et) {
    f32x4 v = ldg4(a.SL + (size_t)li[et] * 64 + f) + ldg4(a.SR + (size_t)ri[et] * 64 + f);
    v = v + ldg4(a.NT + (size_t)li[et] * MDX_NTW + MDX_NT_NFL + f);
    v = v + ldg4(a.NT + (size_t)ri[et] * MDX_NTW + MDX_NT_NFR + f);
    u[0][et] = v + bs;
  }
  gemm_tile<1, TET, 64>(u, a.w.Wself, 4, wave, Hep, LD64, lane);
  float rstd[TET];
  ln_xhat<1, TET, 4>(u, rstd, red, red2, wave, lane, true);
  // gy = Wout^T gHe
  f32x4 g[1][TET];
  acc_zero<1, TET>(g);
  gemm_tile<1, TET, 64>(g, a.WoutT, 4, wave, G, LD64, lane);
  ln_relu_bwd<1, TET, 4>(g, u, rstd, a.w.lng, a.w.lnb, wave, red3, red4, wave, lane, true);
#pragma unroll
  for (int et = 0; et < TET; ++et)
    if (valid[et]) stg4(a.GU + (size_t)(e0 + 16 * et + c) * 64 + f, g[0][et]);
  acc_to_lds<1, TET>(g, X, LD64, 0, wave, lane);
  __syncthreads();
  f32x4 o[1][TET];
#pragma unroll
  for (int et = 0; et < TET; ++et) o[0][et] = lds4(G + (16 * et + c) * LD64 + f);
  gemm_tile<1, TET, 64>(o, a.WselfT, 4, wave, X, LD64, lane);
#pragma unroll
  for (int et = 0; et < TET; ++et)
    if (valid[et]) stg4(a.GHEP + (size_t)(e0 + 16 * et + c) * 64 + f, o[0][et]);
}

// =================================================================================================
// B2: backward of edge kernel A for one tile of 32 edges (recompute + dgrad).
// =================================================================================================
constexpr int BET = 2;
constexpr int BTE = 16 * BET;
constexpr int B_HEP = 0;                      // [BTE][72]  He'
constexpr int B_X = B_HEP + BTE * LD64;       // [BTE][264]
constexpr int B_Y = B_X + BTE * LD256;        // [BTE][264]
// S, S2, GG are only live in the BondFFN / edge_embs sections, Y only in the message-path section: they share
// storage, which brings the workgroup to 78.8 KB of LDS = two workgroups per CU.
constexpr int B_S = B_Y;                      // [BTE][72]
constexpr int B_S2 = B_S + BTE * LD64;        // [BTE][72]
constexpr int B_GG = B_S2 + BTE * LD64;       // [BTE][40]
static_assert(B_GG + BTE * LD32 <= B_Y + BTE * LD256, "aliased buffers must fit inside Y");
constexpr int B_RED = B_Y + BTE * LD256;      // 4 x (4*BTE)
constexpr int B_TOTAL = B_RED + 16 * BTE;

__global__ __launch_bounds__(MDX_WG, 2) void edge_bwd_kernel(const EdgeBwdArgs a, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Hep = smem + B_HEP;
  float* X = smem + B_X;
  float* Y = smem + B_Y;
  float* S = smem + B_S;
  float* S2 = smem + B_S2;
  float* GG = smem + B_GG;
  float* red = smem + B_RED;
  float *red2 = red + 4 * BTE, *red3 = red + 8 * BTE, *red4 = red + 12 * BTE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, q = lane >> 4;
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int e0 = tile * BTE, E = a.E;
  int li[BET], ri[BET];
  float tt[BET];
  bool valid[BET];
#pragma unroll
  for (int et = 0; et < BET; ++et) {
    const int e = e0 + 16 * et + c;
    valid[et] = e < E;
    li[et] = valid[et] ? a.l[e] : 0;
    ri[et] = valid[et] ? a.r[e] : 0;
    tt[et] = valid[et] ? a.te[e] : 0.f;
  }
  load_rows64_t(a.Hep, e0, E, Hep, LD64, tid, BTE);
  __syncthreads();

  // running dL/dHe' for this wave's 16-feature slice (starts from the tail's partial gradient)
  const int f1 = 16 * wave + 4 * q;
  f32x4 ghe[1][BET];
#pragma unroll
  for (int et = 0; et < BET; ++et)
    ghe[0][et] = valid[et] ? ldg4(a.GHEP + (size_t)(e0 + 16 * et + c) * 64 + f1) : splat4(0.f);

  const int ft0 = 4 * wave;
  // ---------------- NodeBlock message path ------------------------------------------------------
  {
    // forward values of this section come from the tape the forward kernel wrote (sigmoid(gate), edge_net output, gated
    // message): three (E,256) reads replace ~390 kFLOP/edge of recomputation (gate and edge_net second layers, msg_net)
    // backward: gm = dL/d(aggr)[l];  M = m0 * sg
    f32x4 gg2[4][BET];
    {
      f32x4 gm0[4][BET];
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < BET; ++et) {
          const size_t o = (size_t)(e0 + 16 * et + c) * MDX_ND + 16 * (ft0 + ft) + 4 * q;
          const f32x4 sg = valid[et] ? ldg4(a.SG + o) : splat4(0.f);
          const f32x4 mg = valid[et] ? ldg4(a.M + o) : splat4(0.f);
          const f32x4 gm = ldg4(a.GNT + (size_t)li[et] * MDX_NTW + MDX_NT_C + 16 * (ft0 + ft) + 4 * q);
          gm0[ft][et] = gm * sg;
          gg2[ft][et] = gm * mg * (splat4(1.f) - sg);  // m0 * sg = M
        }
      acc_to_lds<4, BET>(gm0, Y, LD256, 0, ft0, lane);
    }
    __syncthreads();
    {
      f32x4 gp[4][BET];
      acc_zero<4, BET>(gp);
      gemm_tile<4, BET, 256>(gp, a.wt.WmT, 16, ft0, Y, LD256, lane);
      // p = he * h[r]:  d he = gp * h[r] ; d h[r] = gp * he
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < BET; ++et) {
          const int f = 16 * (ft0 + ft) + 4 * q;
          if (valid[et]) {
            const size_t o = (size_t)(e0 + 16 * et + c) * MDX_ND + f;
            stg4(a.GH + o, gp[ft][et] * ldg4(a.HE + o));
          }
          gp[ft][et] = gp[ft][et] * ldg4(a.H + (size_t)ri[et] * MDX_ND + f);
        }
      __syncthreads();  // all waves done reading Y (gm0)
      acc_to_lds<4, BET>(gp, Y, LD256, 0, ft0, lane);
    }
    __syncthreads();
    {  // through edge_net: he = W2 relu(LN(q)) + b2, q = W1 He' + b1
      f32x4 gt[4][BET], xh[4][BET];
      acc_zero<4, BET>(gt);
      gemm_tile<4, BET, 256>(gt, a.wt.W2T, 16, ft0, Y, LD256, lane);
      acc_bias<4, BET>(xh, a.w.en.b1, ft0, lane);
      gemm_tile<4, BET, 64>(xh, a.w.en.W1, 16, ft0, Hep, LD64, lane);
      float rstd[BET];
      ln_xhat<4, BET, 4>(xh, rstd, red, red2, wave, lane, true);
      ln_relu_bwd<4, BET, 4>(gt, xh, rstd, a.w.en.g, a.w.en.be, ft0, red3, red4, wave, lane, true);
      acc_to_lds<4, BET>(gt, X, LD256, 0, ft0, lane);
    }
    __syncthreads();
    gemm_tile<1, BET, 256>(ghe, a.wt.W1T, 4, wave, X, LD256, lane);
    __syncthreads();  // X, Y free
    {  // gate backward: g = Wg2 relu(LN(qg)) + b, qg = Wg1e He' + gx[r] + t wt + b
      acc_to_lds<4, BET>(gg2, Y, LD256, 0, ft0, lane);
      __syncthreads();
      f32x4 gt[4][BET], xh[4][BET];
      acc_zero<4, BET>(gt);
      gemm_tile<4, BET, 256>(gt, a.wt.Wg2T, 16, ft0, Y, LD256, lane);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const int f = 16 * (ft0 + ft) + 4 * q;
        const f32x4 b = ldg4(a.w.bg1 + f), wt = ldg4(a.w.wtg1 + f);
#pragma unroll
        for (int et = 0; et < BET; ++et)
          xh[ft][et] = b + ldg4(a.NT + (size_t)ri[et] * MDX_NTW + MDX_NT_GX + f) + splat4(tt[et]) * wt;
      }
      gemm_tile<4, BET, 64>(xh, a.w.Wg1e, 16, ft0, Hep, LD64, lane);
      float rstd[BET];
      ln_xhat<4, BET, 4>(xh, rstd, red, red2, wave, lane, true);
      ln_relu_bwd<4, BET, 4>(gt, xh, rstd, a.w.gg, a.w.gb, ft0, red3, red4, wave, lane, true);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int et = 0; et < BET; ++et)
          if (valid[et]) stg4(a.GGX + (size_t)(e0 + 16 * et + c) * MDX_ND + 16 * (ft0 + ft) + 4 * q, gt[ft][et]);
      acc_to_lds<4, BET>(gt, X, LD256, 0, ft0, lane);
    }
    __syncthreads();
    gemm_tile<1, BET, 256>(ghe, a.wt.Wg1eT, 4, wave, X, LD256, lane);
    __syncthreads();
  }

  // ---------------- the two BondFFNs -----------------------------------------------------------------
#pragma unroll 1
  for (int s = 0; s < 2; ++s) {
    const FfnW& w = a.w.ffn[s];
    const FfnWT& wt = a.wt.ffn[s];
    const int nlcol = s ? MDX_NT_NLR : MDX_NT_NLL;
    const int gxcol = s ? MDX_NT_GXR : MDX_NT_GXL;
    const int gfcol = s ? MDX_NT_NFR : MDX_NT_NFL;  // A_r (for right) / A_l (for left) live in these columns of GNT
    int idx[BET], oidx[BET];
#pragma unroll
    for (int et = 0; et < BET; ++et) {
      idx[et] = s ? ri[et] : li[et];    // node whose features enter the FFN
      oidx[et] = s ? li[et] : ri[et];   // node the FFN output is summed into
    }
    const int fa = 2 * wave;
    f32x4 bl[2][BET], nlv[2][BET], xh1[2][BET], o[1][BET], sgt[1][BET], xhg[1][BET];
    float rstd1[BET], rstdg[BET];
    const bool act = wave < 2;
    {  // forward recompute
      acc_zero<2, BET>(bl);
      gemm_tile<2, BET, 64>(bl, w.Wbl, 8, fa, Hep, LD64, lane);
      f32x4 inter[2][BET];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int et = 0; et < BET; ++et) {
          nlv[ft][et] = ldg4(a.NT + (size_t)idx[et] * MDX_NTW + nlcol + 16 * (fa + ft) + 4 * q);
          inter[ft][et] = bl[ft][et] * nlv[ft][et];
        }
      acc_to_lds<2, BET>(inter, X, LD256, 0, fa, lane);
      __syncthreads();
      acc_bias<2, BET>(xh1, w.inter.b1, fa, lane);
      gemm_tile<2, BET, 128>(xh1, w.inter.W1, 8, fa, X, LD256, lane);
      ln_xhat<2, BET, 4>(xh1, rstd1, red, red2, wave, lane, true);
      f32x4 i1[2][BET];
      ln_apply_relu<2, BET>(i1, xh1, w.inter.g, w.inter.be, fa, lane);
      acc_to_lds<2, BET>(i1, X, LD256, 128, fa, lane);
      __syncthreads();
      acc_bias<1, BET>(o, w.inter.b2, wave, lane);
      gemm_tile<1, BET, 128>(o, w.inter.W2, 4, wave, X + 128, LD256, lane);
      // gate
      if (act) {
        const int f = 16 * wave + 4 * q;
        const f32x4 b = ldg4(w.bg1 + f), wtv = ldg4(w.wtg1 + f);
#pragma unroll
        for (int et = 0; et < BET; ++et)
          xhg[0][et] = b + ldg4(a.NT + (size_t)idx[et] * MDX_NTW + gxcol + f) + splat4(tt[et]) * wtv;
        gemm_tile<1, BET, 64>(xhg, w.Wg1e, 2, wave, Hep, LD64, lane);
      } else {
        acc_zero<1, BET>(xhg);
      }
      ln_xhat<1, BET, 2>(xhg, rstdg, red, red2, wave, lane, act);
      if (act) {
        f32x4 g1[1][BET];
        ln_apply_relu<1, BET>(g1, xhg, w.gg, w.gb, wave, lane);
        acc_to_lds<1, BET>(g1, GG, LD32, 0, wave, lane);
      }
      __syncthreads();
      acc_bias<1, BET>(sgt, w.bg2, wave, lane);
      gemm_tile<1, BET, 32>(sgt, w.Wg2, 4, wave, GG, LD32, lane);
#pragma unroll
      for (int et = 0; et < BET; ++et) sgt[0][et] = sigmoid4(sgt[0][et]);
    }
    // backward: f = o * sigmoid(gate);  gf = A[oidx]
    {
      f32x4 go[1][BET], ggt[1][BET];
#pragma unroll
      for (int et = 0; et < BET; ++et) {
        const f32x4 gf = ldg4(a.GNT + (size_t)oidx[et] * MDX_NTW + gfcol + f1);
        go[0][et] = gf * sgt[0][et];
        ggt[0][et] = gf * o[0][et] * sgt[0][et] * (splat4(1.f) - sgt[0][et]);
      }
      acc_to_lds<1, BET>(go, S, LD64, 0, wave, lane);
      acc_to_lds<1, BET>(ggt, S2, LD64, 0, wave, lane);
    }
    __syncthreads();
    {
      f32x4 gi1[2][BET];
      acc_zero<2, BET>(gi1);
      gemm_tile<2, BET, 64>(gi1, wt.Wi2T, 8, fa, S, LD64, lane);
      ln_relu_bwd<2, BET, 4>(gi1, xh1, rstd1, w.inter.g, w.inter.be, fa, red3, red4, wave, lane, true);
      acc_to_lds<2, BET>(gi1, X, LD256, 0, fa, lane);  // X[:, :128] (inter) no longer needed
      __syncthreads();
      f32x4 gin[2][BET];
      acc_zero<2, BET>(gin);
      gemm_tile<2, BET, 128>(gin, wt.Wi1T, 8, fa, X, LD256, lane);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int et = 0; et < BET; ++et) {
          if (valid[et]) stg4(a.GNL[s] + (size_t)(e0 + 16 * et + c) * 128 + 16 * (fa + ft) + 4 * q, gin[ft][et] * bl[ft][et]);
          gin[ft][et] = gin[ft][et] * nlv[ft][et];
        }
      acc_to_lds<2, BET>(gin, X, LD256, 128, fa, lane);  // X[:, 128:] (i1) no longer needed
      __syncthreads();
      gemm_tile<1, BET, 128>(ghe, wt.WblT, 4, wave, X + 128, LD256, lane);
    }
    {  // gate backward
      f32x4 ggg[1][BET];
      acc_zero<1, BET>(ggg);
      if (act) gemm_tile<1, BET, 64>(ggg, wt.Wg2T, 2, wave, S2, LD64, lane);
      ln_relu_bwd<1, BET, 2>(ggg, xhg, rstdg, w.gg, w.gb, wave, red3, red4, wave, lane, act);
      if (act) {
#pragma unroll
        for (int et = 0; et < BET; ++et)
          if (valid[et]) stg4(a.GGXS[s] + (size_t)(e0 + 16 * et + c) * 32 + 16 * wave + 4 * q, ggg[0][et]);
        acc_to_lds<1, BET>(ggg, GG, LD32, 0, wave, lane);
      }
      __syncthreads();
      gemm_tile<1, BET, 32>(ghe, wt.Wg1eT, 4, wave, GG, LD32, lane);
    }
    __syncthreads();
  }

  // ---------------- edge_embs backward: He' = Wemb [He_i | D] + b -----------------------------------------
  acc_to_lds<1, BET>(ghe, S, LD64, 0, wave, lane);
  __syncthreads();
  {
    f32x4 gi[1][BET];
    acc_zero<1, BET>(gi);
    gemm_tile<1, BET, 64>(gi, a.wt.WembHT, 4, wave, S, LD64, lane);
#pragma unroll
    for (int et = 0; et < BET; ++et)
      if (valid[et]) stg4(a.gHe_out + (size_t)(e0 + 16 * et + c) * 64 + f1, gi[0][et]);
  }
  if (wave == 0) {
    f32x4 gd[1][BET];
    acc_zero<1, BET>(gd);
    gemm_tile<1, BET, 64>(gd, a.wt.WembDT, 1, 0, S, LD64, lane);
#pragma unroll
    for (int et = 0; et < BET; ++et) {
      // dD_k/dd = D_k * 2 c_k (dc - o_k) for 0 <= d <= cutoff (clamp passes the gradient inclusively)
      const float dx = a.pos[3 * li[et] + 0] - a.pos[3 * ri[et] + 0];
      const float dy = a.pos[3 * li[et] + 1] - a.pos[3 * ri[et] + 1];
      const float dz = a.pos[3 * li[et] + 2] - a.pos[3 * ri[et] + 2];
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      const float dc = fminf(fmaxf(d, 0.f), a.cutoff);
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 4 * q + r;
        const float uu = dc - a.soff[k];
        const float Dk = expf(a.scoef[k] * (uu * uu));
        sacc += gd[0][et][r] * Dk * 2.0f * a.scoef[k] * uu;
      }
      sacc = red_q(sacc);
      if (q == 0 && valid[et]) {
        const int e = e0 + 16 * et + c;
        a.gdist[e] += (d <= a.cutoff) ? sacc : 0.f;
      }
    }
  }
}

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
// four waves each, the two (E,128) BondFFN payloads two each, the two (E,32) gate payloads half a wave each); and the two
// (E,64) reductions that follow the EdgeBlock-tail backward.  Same per-lane loop as seg_reduce_ld_kernel: same bits.
__global__ __launch_bounds__(832) void seg_reduce_bwd_block_kernel(const SegBwdArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int v0 = blockIdx.x * 4, N = a.N;
  if (wave < 8) {  // (E,256) by right endpoint: dL/dH rows and the gate's node part
    const int v = v0 + (wave & 3);
    if (v >= N) return;
    if (wave < 4)
      stg4(a.gH + (size_t)v * MDX_ND + 4 * lane, seg_sum_ld<256>(a.GH, a.col_ptr, a.col_eids, v, lane));
    else
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_GX + 4 * lane, seg_sum_ld<256>(a.GGX, a.col_ptr, a.col_eids, v, lane));
  } else if (wave < 12) {  // (E,128): left FFN by left endpoint, right FFN by right endpoint
    const int v = v0 + 2 * (wave & 1) + (lane >> 5), c4 = lane & 31;
    if (v >= N) return;
    if (wave < 10)
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_NLL + 4 * c4, seg_sum_ld<128>(a.GNL0, a.row_ptr, nullptr, v, c4));
    else
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_NLR + 4 * c4, seg_sum_ld<128>(a.GNL1, a.col_ptr, a.col_eids, v, c4));
  } else {  // (E,32)
    const int v = v0 + ((lane & 31) >> 3), c4 = lane & 7;
    if (v >= N) return;
    if (lane < 32)
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_GXL + 4 * c4, seg_sum_ld<32>(a.GGXS0, a.row_ptr, nullptr, v, c4));
    else
      stg4(a.GNT + (size_t)v * MDX_NTW + MDX_NT_GXR + 4 * c4, seg_sum_ld<32>(a.GGXS1, a.col_ptr, a.col_eids, v, c4));
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

void launch_edge_tail_bwd(const EdgeTailBwdArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const int ntiles = (a.E + TTE - 1) / TTE;
  hipLaunchKernelGGL(edge_tail_bwd_kernel, dim3(ntiles), dim3(MDX_WG), 0, s, a, ntiles);
}

void launch_edge_bwd(const EdgeBwdArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)edge_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, B_TOTAL * 4);
    attr = true;
  }
  const int ntiles = (a.E + BTE - 1) / BTE;
  hipLaunchKernelGGL(edge_bwd_kernel, dim3(ntiles), dim3(MDX_WG), B_TOTAL * 4, s, a, ntiles);
}

void launch_node_bwd(const NodeBwdArgs& a, hipStream_t s) {
  if (a.N <= 0) return;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)node_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, N_TOTAL * 4);
    attr = true;
  }
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
