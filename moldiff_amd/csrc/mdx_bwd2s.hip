// Split-precision build of the row-owner guidance backward (gfx950) -- round 4, opt-in; the exact kernels are mdx_bwd2.hip.
//
// Same argument blocks, work decomposition, tape reads and outputs as edge_bwd2s_kernel<true> / edge_tail_bwd2s_kernel (reference:
// torch.autograd through models/graph.py:352-357, :42-47, :133-141, :286-294 as driven by models/model.py:312-325); only the matrix
// products differ: every GEMM operand (gradients, recomputed activations, He') is split into float16 hi / lo halves on the way
// into v_mfma_f32_16x16x32_f16 (mdx_split.h: rgemm_x = to_xs + rgemm_s), against split packs of the same (transposed) weights.
// Only the tape form exists (the forward of this path always writes the BondFFN tape).
#include "mdx_kernels.h"
#ifndef MDX_RING
#define MDX_RING 3  // weight-ring depth in steps (9.41 / 9.25 / 9.31 ms per guided step at depth 2 / 3 / 4 (4 spills))
#endif
#include "mdx_row.h"
#include "mdx_split.h"
#include "../../include/moldiff_hip.h"
#include <algorithm>

#define STAMPW(i) ((void)0)

namespace {

constexpr int BW_FO = 32 + 7 * 256, BW_FS = 640;
constexpr int BW_CONST_FLOATS = BW_FO + 2 * BW_FS;

// TAPE: the BondFFN intermediates (W_bl He', the inter MLP's pre-LayerNorm activation and its output) come from the forward's tape
// instead of three recomputed GEMMs per side -- the kernel is bound by the matrix pipe, the 2.5 KB per edge of reads are not.
template <bool TAPE>
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void edge_bwd2s_kernel(const EdgeBwdArgs a, const int nunits, const WorkQ wq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  const int E = a.E;
  f32x4* park = reinterpret_cast<f32x4*>(smem + (size_t)wave * PARK_FLOATS) + lane;
  const unsigned lane_off = 16u * lane;
  auto W = [&](const float* p) { return make_ws(p, lane_off); };

  // constants of the layers that are recomputed (fixed LDS layout, offsets in floats)
  float* cb = smem + 4 * PARK_FLOATS;
  const float* c_soff = lds_put<0, 16>(cb, a.soff, tid);
  const float* c_scoef = lds_put<16, 16>(cb, a.scoef, tid);
  const float* c_eb1 = lds_put<32, 256>(cb, a.w.en.b1, tid);
  const float* c_eg = lds_put<32 + 256, 256>(cb, a.w.en.g, tid);
  const float* c_ebe = lds_put<32 + 512, 256>(cb, a.w.en.be, tid);
  const float* c_bg1 = lds_put<32 + 768, 256>(cb, a.w.bg1, tid);
  const float* c_wtg1 = lds_put<32 + 1024, 256>(cb, a.w.wtg1, tid);
  const float* c_gg = lds_put<32 + 1280, 256>(cb, a.w.gg, tid);
  const float* c_gb = lds_put<32 + 1536, 256>(cb, a.w.gb, tid);
  const float *f_bg1[2], *f_wtg1[2], *f_gg[2], *f_gb[2], *f_ib1[2], *f_ig[2], *f_ibe[2], *f_ib2[2], *f_bg2[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* fb = cb + BW_FO + BW_FS * s;
    f_bg1[s] = fb; f_wtg1[s] = fb + 32; f_gg[s] = fb + 64; f_gb[s] = fb + 96; f_ib1[s] = fb + 128; f_ig[s] = fb + 256;
    f_ibe[s] = fb + 384; f_ib2[s] = fb + 512; f_bg2[s] = fb + 576;
  }
  {
    constexpr int FO = BW_FO, FS = BW_FS;
    const FfnW& w0 = a.w.ffn[0];
    const FfnW& w1 = a.w.ffn[1];
    lds_put<FO, 32>(cb, w0.bg1, tid); lds_put<FO + 32, 32>(cb, w0.wtg1, tid); lds_put<FO + 64, 32>(cb, w0.gg, tid);
    lds_put<FO + 96, 32>(cb, w0.gb, tid); lds_put<FO + 128, 128>(cb, w0.inter.b1, tid); lds_put<FO + 256, 128>(cb, w0.inter.g, tid);
    lds_put<FO + 384, 128>(cb, w0.inter.be, tid); lds_put<FO + 512, 64>(cb, w0.inter.b2, tid); lds_put<FO + 576, 64>(cb, w0.bg2, tid);
    lds_put<FO + FS, 32>(cb, w1.bg1, tid); lds_put<FO + FS + 32, 32>(cb, w1.wtg1, tid); lds_put<FO + FS + 64, 32>(cb, w1.gg, tid);
    lds_put<FO + FS + 96, 32>(cb, w1.gb, tid); lds_put<FO + FS + 128, 128>(cb, w1.inter.b1, tid);
    lds_put<FO + FS + 256, 128>(cb, w1.inter.g, tid); lds_put<FO + FS + 384, 128>(cb, w1.inter.be, tid);
    lds_put<FO + FS + 512, 64>(cb, w1.inter.b2, tid); lds_put<FO + FS + 576, 64>(cb, w1.bg2, tid);
  }
  __syncthreads();

  // units of this wave: drawn from its pair's counter (mdx_row.h, WorkQ), or a contiguous range of the static split
  const bool dyn = wq.ctr != nullptr;
  WorkPair wp{};
  int ubeg, uend;
  if (dyn) {
    wp = wq_pair(wq);
    uend = wp.end;
    ubeg = wp.beg + wq_take(wq_request(wp.line, lane));
  } else {
    const int nslots = gridDim.x * 4;
    const int slot0 = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    const int per = (nunits + nslots - 1) / nslots;
    ubeg = slot0 * per;
    uend = min(nunits, ubeg + per);
  }
  if (ubeg >= uend) {
    if (dyn) wq_leave(wp, lane);
    return;
  }

  const float* wfirst = a.wt.ss.WmT;
  WRing ring;
  ring_prime(ring, W(wfirst));

#pragma unroll 1
  for (int unit = ubeg;;) {
    int q = q0;
    asm volatile("" : "+v"(q));  // opaque per iteration (no address hoisting out of the persistent loop)
    const int ureq = dyn ? wq_request(wp.line, lane) : 0;  // the next unit, consumed at the end of this one
    STAMPW(46);
    STAMPW(0);
    const RowTile t = load_tile(a.l, a.r, a.te, unit * ROWS, E, c);
    // He' (tape) and the running dL/dHe'.  Both enter late on purpose: the first two GEMMs need every register, so He' is
    // requested under the second one and the EdgeBlock tail's part of dL/dHe' (GHEP) is added at the very end.
    f32x4 hep[4][RR], ghe[4][RR];

    // ---------------- NodeBlock message path: M = msg_net(he * h[r]) * sg, aggr[l] += M ----------------
    {
      f32x4 u[16][RR], v[16][RR];
      // gm = dL/d aggr [l];  d m0 = gm * sg (-> u);  d gate_pre = gm * M * (1 - sg) (parked until the gate section)
      row_gather<16, RR>(u, a.SG, t.row, MDX_ND, q);
      row_gather<16, RR>(v, a.M, t.row, MDX_ND, q);
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // the gradient-table rows in two bursts of 8 feature tiles
        f32x4 gm[8][RR];
        row_gather<8, RR>(gm, a.GNT + MDX_NT_C + 128 * h, t.li, MDX_NTW, q);
#pragma unroll
        for (int f8 = 0; f8 < 8; ++f8)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) {
            const int ft = 8 * h + f8;
            const f32x4 sg = u[ft][rt];
            u[ft][rt] = gm[f8][rt] * sg;
            park[(ft * RR + rt) * 64] = gm[f8][rt] * v[ft][rt] * (splat4(1.f) - sg);
          }
      }
      row_zero<16, RR>(v);
      STAMPW(1);
      rgemm_x<16, 16>(v, u, W(a.wt.ss.WmT), ring, W(a.wt.ss.W2T));
      STAMPW(2);
      // p = he * h[r]:  d h[r] = gp * he (-> GH, reduced by right endpoint afterwards);  d he = gp * h[r]
      {
        f32x4 hr[16][RR];
        row_gather<16, RR>(u, a.HE, t.row, MDX_ND, q);
        row_gather<16, RR>(hr, a.H, t.ri, MDX_ND, q);
        row_gather<4, RR>(hep, a.Hep, t.row, 64, q);
        mul_inplace<16>(u, v);
        row_store<16, RR>(u, a.GH, t.row, t.valid, MDX_ND, q);
        mul_inplace<16>(v, hr);
      }
      // through edge_net: he = W2 relu(LN(x)) + b2, x = W1 He' + b1
      row_zero<16, RR>(u);
      STAMPW(3);
      rgemm_x<16, 16>(u, v, W(a.wt.ss.W2T), ring, W(a.w.ss.W1));
      STAMPW(4);
      row_bias<16, RR>(v, c_eb1, q);
      rgemm_x<4, 16>(v, hep, W(a.w.ss.W1), ring, W(a.wt.ss.W1T));
      STAMPW(5);
      {
        float rstd[RR];
        row_ln_xhat<16, RR>(v, rstd);
        row_ln_relu_bwd<16, RR>(u, v, rstd, c_eg, c_ebe, q);
      }
      row_zero<4, RR>(ghe);
      STAMPW(6);
      rgemm_x<16, 4>(ghe, u, W(a.wt.ss.W1T), ring, W(a.wt.ss.Wg2T));
      STAMPW(7);
      // gate: g = Wg2 relu(LN(xg)) + b, xg = Wg1e He' + gx[r] + t wt + b
#pragma unroll
      for (int ft = 0; ft < 16; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) v[ft][rt] = park[(ft * RR + rt) * 64];
      row_zero<16, RR>(u);
      STAMPW(8);
      rgemm_x<16, 16>(u, v, W(a.wt.ss.Wg2T), ring, W(a.w.ss.Wg1e));
      STAMPW(9);
      row_gather<16, RR>(v, a.NT + MDX_NT_GX, t.ri, MDX_NTW, q);
#pragma unroll
      for (int ft = 0; ft < 16; ++ft) {
        const f32x4 b = lds4(c_bg1 + 16 * ft + 4 * q), wt = lds4(c_wtg1 + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) v[ft][rt] = (b + v[ft][rt]) + splat4(t.tt[rt]) * wt;
      }
      STAMPW(10);
      rgemm_x<4, 16>(v, hep, W(a.w.ss.Wg1e), ring, W(a.wt.ss.Wg1eT));
      STAMPW(11);
      {
        float rstd[RR];
        row_ln_xhat<16, RR>(v, rstd);
        row_ln_relu_bwd<16, RR>(u, v, rstd, c_gg, c_gb, q);
      }
      row_store<16, RR>(u, a.GGX, t.row, t.valid, MDX_ND, q);
      STAMPW(12);
      rgemm_x<16, 4>(ghe, u, W(a.wt.ss.Wg1eT), ring, W(TAPE ? a.w.ss.ffn[0].Wg1e : a.w.ss.ffn[0].Wbl));
      STAMPW(13);
    }

    // ---------------- the two BondFFNs: f = inter((Wbl He') * nl[idx]) * sigmoid(gate([He' | x[idx] | t])) ----------------
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s == 1) STAMPW(21);
      const FfnS& ws = a.w.ss.ffn[s];
      const FfnTS& wts = a.wt.ss.ffn[s];
      constexpr int nlcol = s ? MDX_NT_NLR : MDX_NT_NLL, gxcol = s ? MDX_NT_GXR : MDX_NT_GXL;
      constexpr int gfcol = s ? MDX_NT_NFR : MDX_NT_NFL;  // A_r (for right) / A_l (for left) live in these columns of GNT
      int idx[RR], oidx[RR];
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        idx[rt] = s ? t.ri[rt] : t.li[rt];   // node whose features enter the FFN
        oidx[rt] = s ? t.li[rt] : t.ri[rt];  // node the FFN output is summed into
      }
      // bl = Wbl He' and nl[idx] are needed again at the end of the backward: they wait in the wave's LDS area (free in this
      // section) instead of 128 registers
      f32x4 xh1[8][RR], o[4][RR], sgt[4][RR], xhg[2][RR];
      float rstd1[RR], rstdg[RR];
      if constexpr (TAPE) {  // forward values from the tape; only the gate (two small GEMMs) is recomputed
        {
          f32x4 blv[8][RR], nlv[8][RR];
          row_gather<8, RR>(blv, a.BL[s], t.row, 128, q);
          row_gather<8, RR>(nlv, a.NT + nlcol, idx, MDX_NTW, q);
#pragma unroll
          for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int rt = 0; rt < RR; ++rt) {
              park[(ft * RR + rt) * 64] = blv[ft][rt];
              park[((8 + ft) * RR + rt) * 64] = nlv[ft][rt];
            }
        }
        row_gather<8, RR>(xh1, a.H1[s], t.row, 128, q);
        row_gather<4, RR>(o, a.O[s], t.row, 64, q);
        row_gather<2, RR>(xhg, a.NT + gxcol, idx, MDX_NTW, q);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
          const f32x4 b = lds4(f_bg1[s] + 16 * ft + 4 * q), wt = lds4(f_wtg1[s] + 16 * ft + 4 * q);
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) xhg[ft][rt] = (b + xhg[ft][rt]) + splat4(t.tt[rt]) * wt;
        }
        rgemm_x<4, 2>(xhg, hep, W(ws.Wg1e), ring, W(ws.Wg2));
        row_ln_xhat<8, RR>(xh1, rstd1);
        row_ln_xhat<2, RR>(xhg, rstdg);
        f32x4 g1[2][RR];
        row_ln_apply_relu<2, RR>(g1, xhg, f_gg[s], f_gb[s], q);
        row_bias<4, RR>(sgt, f_bg2[s], q);
        rgemm_x<2, 4>(sgt, g1, W(ws.Wg2), ring, W(wts.Wi2T));
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) sgt[ft][rt] = row_sigmoid4(sgt[ft][rt]);
      } else
      {  // forward recompute
        f32x4 tmp[8][RR];
        {
          f32x4 nlv[8][RR];
          row_zero<8, RR>(tmp);
          rgemm_x<4, 8>(tmp, hep, W(ws.Wbl), ring, W(ws.W1));
          row_gather<8, RR>(nlv, a.NT + nlcol, idx, MDX_NTW, q);
#pragma unroll
          for (int ft = 0; ft < 8; ++ft)
#pragma unroll
            for (int rt = 0; rt < RR; ++rt) {
              park[(ft * RR + rt) * 64] = tmp[ft][rt];
              park[((8 + ft) * RR + rt) * 64] = nlv[ft][rt];
              tmp[ft][rt] = tmp[ft][rt] * nlv[ft][rt];
            }
        }
        row_bias<8, RR>(xh1, f_ib1[s], q);
        rgemm_x<8, 8>(xh1, tmp, W(ws.W1), ring, W(ws.W2));
        row_ln_xhat<8, RR>(xh1, rstd1);
        row_ln_apply_relu<8, RR>(tmp, xh1, f_ig[s], f_ibe[s], q);
        row_bias<4, RR>(o, f_ib2[s], q);
        rgemm_x<8, 4>(o, tmp, W(ws.W2), ring, W(ws.Wg1e));
        row_gather<2, RR>(xhg, a.NT + gxcol, idx, MDX_NTW, q);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
          const f32x4 b = lds4(f_bg1[s] + 16 * ft + 4 * q), wt = lds4(f_wtg1[s] + 16 * ft + 4 * q);
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) xhg[ft][rt] = (b + xhg[ft][rt]) + splat4(t.tt[rt]) * wt;
        }
        rgemm_x<4, 2>(xhg, hep, W(ws.Wg1e), ring, W(ws.Wg2));
        row_ln_xhat<2, RR>(xhg, rstdg);
        f32x4 g1[2][RR];
        row_ln_apply_relu<2, RR>(g1, xhg, f_gg[s], f_gb[s], q);
        row_bias<4, RR>(sgt, f_bg2[s], q);
        rgemm_x<2, 4>(sgt, g1, W(ws.Wg2), ring, W(wts.Wi2T));
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) sgt[ft][rt] = row_sigmoid4(sgt[ft][rt]);
      }
      STAMPW(14 + 3 * s);
      // backward: f = o * sigmoid(gate);  gf = A[oidx]
      f32x4 go[4][RR];
      row_gather<4, RR>(go, a.GNT + gfcol, oidx, MDX_NTW, q);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) {
          const f32x4 gf = go[ft][rt];
          go[ft][rt] = gf * sgt[ft][rt];
          sgt[ft][rt] = gf * o[ft][rt] * sgt[ft][rt] * (splat4(1.f) - sgt[ft][rt]);  // d gate_pre
        }
      {
        f32x4 gi1[8][RR], gin[8][RR];
        row_zero<8, RR>(gi1);
        rgemm_x<4, 8>(gi1, go, W(wts.Wi2T), ring, W(wts.Wi1T));
        row_ln_relu_bwd<8, RR>(gi1, xh1, rstd1, f_ig[s], f_ibe[s], q);
        row_zero<8, RR>(gin);
        rgemm_x<8, 8>(gin, gi1, W(wts.Wi1T), ring, W(wts.WblT));
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) {
            gi1[ft][rt] = gin[ft][rt] * park[(ft * RR + rt) * 64];  // d nl[idx] = gin * bl
            gin[ft][rt] = gin[ft][rt] * park[((8 + ft) * RR + rt) * 64];
          }
        row_store<8, RR>(gi1, a.GNL[s], t.row, t.valid, 128, q);
        rgemm_x<8, 4>(ghe, gin, W(wts.WblT), ring, W(wts.Wg2T));
      }
      STAMPW(15 + 3 * s);
      {  // gate backward
        f32x4 ggg[2][RR];
        row_zero<2, RR>(ggg);
        rgemm_x<4, 2>(ggg, sgt, W(wts.Wg2T), ring, W(wts.Wg1eT));
        row_ln_relu_bwd<2, RR>(ggg, xhg, rstdg, f_gg[s], f_gb[s], q);
        row_store<2, RR>(ggg, a.GGXS[s], t.row, t.valid, 32, q);
        rgemm_x<2, 4>(ghe, ggg, W(wts.Wg1eT), ring, W(s == 0 ? (TAPE ? a.w.ss.ffn[1].Wg1e : a.w.ss.ffn[1].Wbl) : a.wt.ss.WembHT));
      }
    });

    STAMPW(20);
    // ---------------- edge_embs backward: He' = Wemb [He_i | D(d)] + b ----------------
    {
      f32x4 gi[4][RR];
      row_gather<4, RR>(gi, a.GHEP, t.row, 64, q);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) ghe[ft][rt] = ghe[ft][rt] + gi[ft][rt];
      row_zero<4, RR>(gi);
      rgemm_x<4, 4>(gi, ghe, W(a.wt.ss.WembHT), ring, W(a.wt.ss.WembDT));
      row_store<4, RR>(gi, a.gHe_out, t.row, t.valid, 64, q);
      f32x4 gd[2][RR];  // 16 distance features, padded to 32 by the pack
      row_zero<2, RR>(gd);
      rgemm_x<4, 2>(gd, ghe, W(a.wt.ss.WembDT), ring, W(wfirst));
      const f32x4 off = lds4(c_soff + 4 * q), coef = lds4(c_scoef + 4 * q);
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        // dD_k/dd = D_k * 2 c_k (dc - o_k) for 0 <= d <= cutoff (clamp passes the gradient inclusively)
        const float dx = a.pos[3 * t.li[rt] + 0] - a.pos[3 * t.ri[rt] + 0];
        const float dy = a.pos[3 * t.li[rt] + 1] - a.pos[3 * t.ri[rt] + 1];
        const float dz = a.pos[3 * t.li[rt] + 2] - a.pos[3 * t.ri[rt] + 2];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float dc = fminf(fmaxf(d, a.smear_start), a.cutoff);
        float sacc = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float uu = dc - off[r];
          const float Dk = expf(coef[r] * (uu * uu));
          sacc += gd[0][rt][r] * Dk * 2.0f * coef[r] * uu;
        }
        sacc = red_q(sacc);
        if (q == 0 && t.valid[rt]) a.gdist[t.row[rt]] += (d >= a.smear_start && d <= a.cutoff) ? sacc : 0.f;  // clamp passes the gradient inside [start, stop]
      }
    }
    STAMPW(40);
    STAMPW(47);
    unit = dyn ? wp.beg + wq_take(ureq) : unit + 1;
    if (unit >= uend) break;
  }
  if (dyn) wq_leave(wp, lane);
}

// EdgeBlock tail backward (reference models/graph.py:286-294 through autograd), row-owner: He'' = He' + out(relu(LN(u))),
// u = self_ffn(He') + SL[l] + SR[r] + nfl[l] + nfr[r].  In: dL/dHe''.  Out: GU = dL/du (reduced per node by the caller) and
// GHEP = dL/dHe'' + self_ffn^T dL/du (the part of dL/dHe' that does not go through the BondFFNs).  Same math as
// mdx_bondpred.hip's edge_tail_bwd_kernel (tile design, MDX_TILE_KERNELS=1).
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void edge_tail_bwd2s_kernel(const EdgeTailBwdArgs a, const int nunits, const WorkQ wq) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  const unsigned lane_off = 16u * lane;
  auto W = [&](const float* p) { return make_ws(p, lane_off); };
  // units of this wave: drawn from its pair's counter (mdx_row.h, WorkQ), or a contiguous range of the static split
  const bool dyn = wq.ctr != nullptr;
  WorkPair wp{};
  int ubeg, uend;
  if (dyn) {
    wp = wq_pair(wq);
    uend = wp.end;
    ubeg = wp.beg + wq_take(wq_request(wp.line, lane));
  } else {
    const int nslots = gridDim.x * 4;
    const int slot0 = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    const int per = (nunits + nslots - 1) / nslots;
    ubeg = slot0 * per;
    uend = min(nunits, ubeg + per);
  }
  if (ubeg >= uend) {
    if (dyn) wq_leave(wp, lane);
    return;
  }
  WRing ring;
  ring_prime(ring, W(a.w.ss.Wself));
#pragma unroll 1
  for (int unit = ubeg;;) {
    int q = q0;
    asm volatile("" : "+v"(q));
    const int ureq = dyn ? wq_request(wp.line, lane) : 0;
    const RowTile t = load_tile(a.l, a.r, a.te, unit * ROWS, a.E, c);
    f32x4 hep[4][RR], g[4][RR], u[4][RR];
    row_gather<4, RR>(hep, a.Hep, t.row, 64, q);
    row_gather<4, RR>(g, a.gHe, t.row, 64, q);
    {  // u's per-node part, in the forward's order of additions
      f32x4 v1[4][RR], v2[4][RR], v3[4][RR];
      row_gather<4, RR>(u, a.SL, t.li, 64, q);
      row_gather<4, RR>(v1, a.SR, t.ri, 64, q);
      row_gather<4, RR>(v2, a.NT + MDX_NT_NFL, t.li, MDX_NTW, q);
      row_gather<4, RR>(v3, a.NT + MDX_NT_NFR, t.ri, MDX_NTW, q);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const f32x4 bs = ldg4(a.w.bself + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) u[ft][rt] = (((u[ft][rt] + v1[ft][rt]) + v2[ft][rt]) + v3[ft][rt]) + bs;
      }
    }
    rgemm_x<4, 4>(u, hep, W(a.w.ss.Wself), ring, W(a.ssWoutT));
    float rstd[RR];
    row_ln_xhat<4, RR>(u, rstd);
    f32x4 gy[4][RR];
    row_zero<4, RR>(gy);
    rgemm_x<4, 4>(gy, g, W(a.ssWoutT), ring, W(a.ssWselfT));
    row_ln_relu_bwd<4, RR>(gy, u, rstd, a.w.lng, a.w.lnb, q);
    row_store<4, RR>(gy, a.GU, t.row, t.valid, 64, q);
    rgemm_x<4, 4>(g, gy, W(a.ssWselfT), ring, W(a.w.ss.Wself));
    row_store<4, RR>(g, a.GHEP, t.row, t.valid, 64, q);
    unit = dyn ? wp.beg + wq_take(ureq) : unit + 1;
    if (unit >= uend) break;
  }
  if (dyn) wq_leave(wp, lane);
}

}  // namespace

void launch_edge_tail_bwd2s(const EdgeTailBwdArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const int nunits = (a.E + ROWS - 1) / ROWS;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  hipLaunchKernelGGL(edge_tail_bwd2s_kernel, dim3(grid), dim3(MDX_WG), 0, s, a, nunits, make_workq(a.wq, nunits, grid, mdx_num_cus()));
}

int launch_edge_bwd2s(const EdgeBwdArgs& a, hipStream_t s) {
  if (a.E <= 0) return 0;
  if (!(a.BL[0] && a.BL[1] && a.H1[0] && a.H1[1] && a.O[0] && a.O[1])) return 1;  // the split backward reads the BondFFN tape
  static bool attr = false;
  constexpr int lds = (4 * PARK_FLOATS + BW_CONST_FLOATS) * 4;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)edge_bwd2s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int nunits = (a.E + ROWS - 1) / ROWS;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  const WorkQ wq = make_workq(a.wq, nunits, grid, mdx_num_cus());
  hipLaunchKernelGGL(edge_bwd2s_kernel<true>, dim3(grid), dim3(MDX_WG), lds, s, a, nunits, wq);
  return 0;
}
