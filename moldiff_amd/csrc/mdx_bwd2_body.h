// Body of the row-owner guidance-backward kernels, shared by the exact fp32 build (mdx_bwd2.hip) and the split float16 build
// (mdx_bwd2s.hip): the two differ only in the matrix products, so the including file defines
//     BW_GEMM(KG, FT)     the GEMM primitive: rgemm<KG, FT, RR> (mdx_row.h) or rgemm_x<KG, FT> (mdx_split.h)
//     BW_S                the stream-pack member of the weight blocks: s or ss
//     BW_KERNEL, BW_TAIL_KERNEL, BW_TAIL_WOUTT, BW_TAIL_WSELFT   kernel names / the tail kernel's transposed packs
// and includes this file inside its anonymous namespace.
//
// edge kernel: the per-edge part of d(bond-predictor logits)/d(pos) for one NodeEdgeNet block of the guidance chain (reference
// models/model.py:312-325 runs torch.autograd through models/graph.py:352-357 edge_embs, :42-47 the NodeBlock message path and
// :133-141 the two BondFFNs; this is that backward, hand-derived).  One wave owns 16 edges and every feature of every layer for
// them; activations and gradients stay in registers in the MFMA accumulator layout (= the B-operand layout of the next GEMM),
// LayerNorm forward/backward reductions are wave-local, transposed weights stream L2 -> registers through the ring of mdx_row.h.
// No barrier after the constant prologue.
//
// Round 5: the wave's 16 edges are 16 consecutive positions of the BY-RIGHT edge order (col_eids), not of the by-left order the
// forward kernels walk.  672 of the 832 gradient floats an edge contributes to the node tables are reduced by its RIGHT end
// point (dL/dh[r], the gate's node part, the right BondFFN's node_linear and gate parts): in by-right order a right node's run is
// contiguous, so those payloads are summed over the run inside the kernel (seg_sum_store, the forward's in-kernel aggregation)
// and leave as one partial row per (node, unit) -- 2.7 KB per edge of stores and the 515 MB read-back of the reduction pass are
// gone.  Tape rows are gathered by edge id either way (a row is >= 256 contiguous bytes).  The wave's LDS area has to be free
// when a payload is ready, so the message section no longer parks dL/d gate_pre across the message chain: the gate chain runs
// first and the message chain re-reads sigmoid(gate) (1 KB per edge) instead.
constexpr int BW_FO = 32 + 7 * 256, BW_FS = 640;
constexpr int BW_CONST_FLOATS = BW_FO + 2 * BW_FS;

template <bool FUSE_TAIL>
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void BW_KERNEL(const EdgeBwdArgs a, const int nunits, const WorkQ wq) {
  static_assert(RR == 1, "the in-kernel segment sums are written for one 16-row tile per wave");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  float* wbuf = smem + (size_t)wave * PARK_FLOATS;
  f32x4* park = reinterpret_cast<f32x4*>(wbuf) + lane;
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
  const float *f_bg1[2], *f_wtg1[2], *f_gg[2], *f_gb[2], *f_ig[2], *f_ibe[2], *f_bg2[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* fb = cb + BW_FO + BW_FS * s;
    f_bg1[s] = fb; f_wtg1[s] = fb + 32; f_gg[s] = fb + 64; f_gb[s] = fb + 96; f_ig[s] = fb + 256;
    f_ibe[s] = fb + 384; f_bg2[s] = fb + 576;
  }
  {
    constexpr int FO = BW_FO, FS = BW_FS;
    const FfnW& w0 = a.w.ffn[0];
    const FfnW& w1 = a.w.ffn[1];
    lds_put<FO, 32>(cb, w0.bg1, tid); lds_put<FO + 32, 32>(cb, w0.wtg1, tid); lds_put<FO + 64, 32>(cb, w0.gg, tid);
    lds_put<FO + 96, 32>(cb, w0.gb, tid); lds_put<FO + 256, 128>(cb, w0.inter.g, tid);
    lds_put<FO + 384, 128>(cb, w0.inter.be, tid); lds_put<FO + 576, 64>(cb, w0.bg2, tid);
    lds_put<FO + FS, 32>(cb, w1.bg1, tid); lds_put<FO + FS + 32, 32>(cb, w1.wtg1, tid); lds_put<FO + FS + 64, 32>(cb, w1.gg, tid);
    lds_put<FO + FS + 96, 32>(cb, w1.gb, tid);
    lds_put<FO + FS + 256, 128>(cb, w1.inter.g, tid); lds_put<FO + FS + 384, 128>(cb, w1.inter.be, tid);
    lds_put<FO + FS + 576, 64>(cb, w1.bg2, tid);
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

  const float* wfirst = a.wt.BW_S.Wg2T;
  WRing ring;
  ring_prime(ring, W(wfirst));

#pragma unroll 1
  for (int unit = ubeg;;) {
    int q = q0;
    asm volatile("" : "+v"(q));  // opaque per iteration (no address hoisting out of the persistent loop)
    const int ureq = dyn ? wq_request(wp.line, lane) : 0;  // the next unit, consumed at the end of this one
    STAMPW(46);
    STAMPW(0);
    const int2 ue = reinterpret_cast<const int2*>(a.units_r)[unit];
    const RowTile t = load_tile_r(a.col_eids, a.col_left, a.col_right, a.te, a.epo_r, ue.x, ue.y, c);
    const int ucnt = __builtin_amdgcn_readfirstlane(t.cnt);
    const int prow = t.pf[0] + unit;  // the partial row of this lane's right node in this unit
    f32x4 hep[4][RR], ghe[4][RR];     // He' (tape) and the running dL/dHe'; the EdgeBlock tail's part (GHEP) is added at the very end

    // ---------------- NodeBlock message path: M = msg_net(he * h[r]) * sg, aggr[l] += M ----------------
    {
      f32x4 u[16][RR], v[16][RR];
      // ---- gate first: d gate_pre = gm * M * (1 - sg), gm = dL/d aggr [l];  g = Wg2 relu(LN(xg)) + b, xg = Wg1e He' + gx[r] + t wt + b
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
            v[ft][rt] = gm[f8][rt] * v[ft][rt] * (splat4(1.f) - u[ft][rt]);
          }
      }
      row_zero<16, RR>(u);
      STAMPW(1);
      BW_GEMM(16, 16)(u, v, W(a.wt.BW_S.Wg2T), ring, W(a.w.BW_S.Wg1e));
      STAMPW(2);
      row_gather<16, RR>(v, a.NT + MDX_NT_GX, t.ri, MDX_NTW, q);
      f32x4 hg[4][RR];  // He' for the gate's recompute; requested again under the message chain's second GEMM (not held across it)
      row_gather<4, RR>(hg, a.Hep, t.row, 64, q);
#pragma unroll
      for (int ft = 0; ft < 16; ++ft) {
        const f32x4 b = lds4(c_bg1 + 16 * ft + 4 * q), wt = lds4(c_wtg1 + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) v[ft][rt] = (b + v[ft][rt]) + splat4(t.tt[rt]) * wt;
      }
      STAMPW(3);
      BW_GEMM(4, 16)(v, hg, W(a.w.BW_S.Wg1e), ring, W(a.wt.BW_S.Wg1eT));
      STAMPW(4);
      {
        float rstd[RR];
        row_ln_xhat<16, RR>(v, rstd);
        row_ln_relu_bwd<16, RR>(u, v, rstd, c_gg, c_gb, q);
      }
      // dL/d gx[r]: summed over the unit's runs of equal right node (partial rows of GGX)
      seg_sum_put<16>(u, wbuf, lane);  // to the LDS area now, summed after the next GEMM (the writes retire under its MFMAs)
      row_zero<4, RR>(ghe);
      STAMPW(5);
      BW_GEMM(16, 4)(ghe, u, W(a.wt.BW_S.Wg1eT), ring, W(a.wt.BW_S.WmT));
      seg_sum_flush<16, 4>(wbuf, lane, ucnt, t.ri[0], prow, a.GGX);
      STAMPW(6);
      // ---- message chain: d m0 = gm * sg (sigmoid(gate) is read again: the wave's LDS area stays free for the segment sums)
      row_gather<16, RR>(u, a.SG, t.row, MDX_ND, q);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 gm[8][RR];
        row_gather<8, RR>(gm, a.GNT + MDX_NT_C + 128 * h, t.li, MDX_NTW, q);
#pragma unroll
        for (int f8 = 0; f8 < 8; ++f8)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) u[8 * h + f8][rt] = gm[f8][rt] * u[8 * h + f8][rt];
      }
      row_zero<16, RR>(v);
      STAMPW(7);
      BW_GEMM(16, 16)(v, u, W(a.wt.BW_S.WmT), ring, W(a.wt.BW_S.W2T));
      STAMPW(8);
      // p = he * h[r]:  d h[r] = gp * he (-> GH, summed over the runs of equal right node);  d he = gp * h[r]
      {
        f32x4 hr[16][RR];
        row_gather<16, RR>(u, a.HE, t.row, MDX_ND, q);
        row_gather<16, RR>(hr, a.H, t.ri, MDX_ND, q);
        mul_inplace<16>(u, v);
        seg_sum_put<16>(u, wbuf, lane);  // GH goes to the LDS area now (u is free for the next GEMM) and is summed after it
        mul_inplace<16>(v, hr);
      }
      row_gather<4, RR>(hep, a.Hep, t.row, 64, q);
      // through edge_net: he = W2 relu(LN(x)) + b2, x = W1 He' + b1
      row_zero<16, RR>(u);
      STAMPW(9);
      BW_GEMM(16, 16)(u, v, W(a.wt.BW_S.W2T), ring, W(a.w.BW_S.W1));
      STAMPW(10);
      seg_sum_flush<16, 4>(wbuf, lane, ucnt, t.ri[0], prow, a.GH);
      row_bias<16, RR>(v, c_eb1, q);
      BW_GEMM(4, 16)(v, hep, W(a.w.BW_S.W1), ring, W(a.wt.BW_S.W1T));
      STAMPW(11);
      {
        float rstd[RR];
        row_ln_xhat<16, RR>(v, rstd);
        row_ln_relu_bwd<16, RR>(u, v, rstd, c_eg, c_ebe, q);
      }
      STAMPW(12);
      BW_GEMM(16, 4)(ghe, u, W(a.wt.BW_S.W1T), ring, W(a.w.BW_S.ffn[0].Wg1e));
      STAMPW(13);
    }

    // ---------------- the two BondFFNs: f = inter((Wbl He') * nl[idx]) * sigmoid(gate([He' | x[idx] | t])) ----------------
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if (s == 1) STAMPW(21);
      const FfnS& ws = a.w.BW_S.ffn[s];
      const FfnTS& wts = a.wt.BW_S.ffn[s];
      constexpr int nlcol = s ? MDX_NT_NLR : MDX_NT_NLL, gxcol = s ? MDX_NT_GXR : MDX_NT_GXL;
      constexpr int gfcol = s ? MDX_NT_NFR : MDX_NT_NFL;  // A_r (for right) / A_l (for left) live in these columns of GNT
      int idx[RR], oidx[RR];
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        idx[rt] = s ? t.ri[rt] : t.li[rt];   // node whose features enter the FFN
        oidx[rt] = s ? t.li[rt] : t.ri[rt];  // node the FFN output is summed into
      }
      // forward values from the tape (W_bl He', the inter MLP's pre-LayerNorm activation and its output); only the gate (two small
      // GEMMs) is recomputed.  bl = Wbl He' and nl[idx] are needed again at the end of the backward: they wait in the wave's LDS
      // area (free in this section) instead of 128 registers
      f32x4 xh1[8][RR], o[4][RR], sgt[4][RR], xhg[2][RR];
      float rstd1[RR], rstdg[RR];
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
      BW_GEMM(4, 2)(xhg, hep, W(ws.Wg1e), ring, W(ws.Wg2));
      row_ln_xhat<8, RR>(xh1, rstd1);
      row_ln_xhat<2, RR>(xhg, rstdg);
      {
        f32x4 g1[2][RR];
        row_ln_apply_relu<2, RR>(g1, xhg, f_gg[s], f_gb[s], q);
        row_bias<4, RR>(sgt, f_bg2[s], q);
        BW_GEMM(2, 4)(sgt, g1, W(ws.Wg2), ring, W(wts.Wi2T));
      }
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) sgt[ft][rt] = row_sigmoid4(sgt[ft][rt]);
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
        BW_GEMM(4, 8)(gi1, go, W(wts.Wi2T), ring, W(wts.Wi1T));
        row_ln_relu_bwd<8, RR>(gi1, xh1, rstd1, f_ig[s], f_ibe[s], q);
        row_zero<8, RR>(gin);
        BW_GEMM(8, 8)(gin, gi1, W(wts.Wi1T), ring, W(wts.WblT));
#pragma unroll
        for (int ft = 0; ft < 8; ++ft)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) {
            gi1[ft][rt] = gin[ft][rt] * park[(ft * RR + rt) * 64];  // d nl[idx] = gin * bl
            gin[ft][rt] = gin[ft][rt] * park[((8 + ft) * RR + rt) * 64];
          }
        if constexpr (s == 1) {  // idx = right node: summed in the kernel (the LDS area is free again: bl / nl have been consumed)
          __builtin_amdgcn_wave_barrier();
          seg_sum_store<8>(gi1, wbuf, lane, ucnt, t.ri[0], prow, a.GNL[1]);
        } else {
          row_store<8, RR>(gi1, a.GNL[0], t.row, t.valid, 128, q);
        }
        BW_GEMM(8, 4)(ghe, gin, W(wts.WblT), ring, W(wts.Wg2T));
      }
      STAMPW(15 + 3 * s);
      {  // gate backward
        f32x4 ggg[2][RR];
        row_zero<2, RR>(ggg);
        BW_GEMM(4, 2)(ggg, sgt, W(wts.Wg2T), ring, W(wts.Wg1eT));
        row_ln_relu_bwd<2, RR>(ggg, xhg, rstdg, f_gg[s], f_gb[s], q);
        if constexpr (s == 1) seg_sum_store<2>(ggg, wbuf, lane, ucnt, t.ri[0], prow, a.GGXS[1]);
        else row_store<2, RR>(ggg, a.GGXS[0], t.row, t.valid, 32, q);
        BW_GEMM(2, 4)(ghe, ggg, W(wts.Wg1eT), ring, W(s == 0 ? a.w.BW_S.ffn[1].Wg1e : (FUSE_TAIL ? a.wt.BW_S.WembHT : a.wt.BW_S.WembDT)));
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
      if constexpr (FUSE_TAIL) {
        row_zero<4, RR>(gi);
        BW_GEMM(4, 4)(gi, ghe, W(a.wt.BW_S.WembHT), ring, W(a.wt.BW_S.WembDT));  // gi = dL/dHe''_{i-1}, stays in registers for the tail below
      }
      f32x4 gd[2][RR];  // 16 distance features, padded to 32 by the pack
      row_zero<2, RR>(gd);
      BW_GEMM(4, 2)(gd, ghe, W(a.wt.BW_S.WembDT), ring, W(FUSE_TAIL ? a.tWself : wfirst));
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
      // ---------------- EdgeBlock tail of block i - 1, backward (models/graph.py:286-294 through autograd) ----------------
      // He''_{i-1} = He'_{i-1} + out(relu(LN(u))),  u = self_ffn(He') + SL[l] + SR[r] + nfl[l] + nfr[r];  in: gi = dL/dHe''_{i-1}.
      // Out: GU = dL/du (reduced per node by the caller) and GHEP = dL/dHe'' + self_ffn^T dL/du for block i - 1's own launch.
      if constexpr (FUSE_TAIL) {
        f32x4 hp[4][RR], u[4][RR];
        row_gather<4, RR>(hp, a.tHep, t.row, 64, q);
        {  // u's per-node part, in the forward's order of additions
          f32x4 v1[4][RR], v2[4][RR], v3[4][RR];
          row_gather<4, RR>(u, a.tSL, t.li, 64, q);
          row_gather<4, RR>(v1, a.tSR, t.ri, 64, q);
          row_gather<4, RR>(v2, a.tNT + MDX_NT_NFL, t.li, MDX_NTW, q);
          row_gather<4, RR>(v3, a.tNT + MDX_NT_NFR, t.ri, MDX_NTW, q);
#pragma unroll
          for (int ft = 0; ft < 4; ++ft) {
            const f32x4 bs = ldg4(a.tbself + 16 * ft + 4 * q);
#pragma unroll
            for (int rt = 0; rt < RR; ++rt) u[ft][rt] = (((u[ft][rt] + v1[ft][rt]) + v2[ft][rt]) + v3[ft][rt]) + bs;
          }
        }
        BW_GEMM(4, 4)(u, hp, W(a.tWself), ring, W(a.tWoutT));
        float rstd[RR];
        row_ln_xhat<4, RR>(u, rstd);
        f32x4 gy[4][RR];
        row_zero<4, RR>(gy);
        BW_GEMM(4, 4)(gy, gi, W(a.tWoutT), ring, W(a.tWselfT));
        row_ln_relu_bwd<4, RR>(gy, u, rstd, a.tlng, a.tlnb, q);
        row_store<4, RR>(gy, a.tGU, t.row, t.valid, 64, q);
        BW_GEMM(4, 4)(gi, gy, W(a.tWselfT), ring, W(wfirst));
        row_store<4, RR>(gi, a.tGHEP, t.row, t.valid, 64, q);
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
// GHEP = dL/dHe'' + self_ffn^T dL/du (the part of dL/dHe' that does not go through the BondFFNs).
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void BW_TAIL_KERNEL(const EdgeTailBwdArgs a, const int nunits, const WorkQ wq) {
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
  ring_prime(ring, W(a.w.BW_S.Wself));
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
    BW_GEMM(4, 4)(u, hep, W(a.w.BW_S.Wself), ring, W(a.BW_TAIL_WOUTT));
    float rstd[RR];
    row_ln_xhat<4, RR>(u, rstd);
    f32x4 gy[4][RR];
    row_zero<4, RR>(gy);
    BW_GEMM(4, 4)(gy, g, W(a.BW_TAIL_WOUTT), ring, W(a.BW_TAIL_WSELFT));
    row_ln_relu_bwd<4, RR>(gy, u, rstd, a.w.lng, a.w.lnb, q);
    row_store<4, RR>(gy, a.GU, t.row, t.valid, 64, q);
    BW_GEMM(4, 4)(g, gy, W(a.BW_TAIL_WSELFT), ring, W(a.w.BW_S.Wself));
    row_store<4, RR>(g, a.GHEP, t.row, t.valid, 64, q);
    unit = dyn ? wp.beg + wq_take(ureq) : unit + 1;
    if (unit >= uend) break;
  }
  if (dyn) wq_leave(wp, lane);
}
