// Split-precision build of the row-owner edge kernel A (gfx950) -- round 4, opt-in (mdx_model_set_matrix_path(m, 1)).
//
// Same sections, same work decomposition, same argument block and the same outputs as edge_a2_kernel (mdx_edge2.hip):
//   reference models/graph.py:352-357 edge_embs, :42-47 NodeBlock message path, :133-141/:278,:282 the two EdgeBlock BondFFNs,
//   :50 / :283 the by-left segment sums (EA_AGG).
// Only the matrix products differ: every GEMM runs on v_mfma_f32_16x16x32_f16 with operands split into float16 hi / lo halves
// (mdx_split.h: three products per k-group, fp32 accumulation, ~22 significand bits per operand).  Everything that is not a
// matrix product (smearing, biases, LayerNorm, gates, segment sums, stores) is the fp32 code of the exact kernel.
#include "mdx_kernels.h"
// Decomposition of the split build (measured on the bench workload, ms per sampling step, kernel A / kernel B): 16 rows x 2 waves per
// SIMD like the exact kernels 2.80 / 0.96 -- there the per-wave weight stream (the same bytes as fp32, 11.9 GB per kernel-A launch)
// runs at the L1/L2 limit (~25 TB/s) and IS the kernel time; 32 rows x 1 wave (every weight fragment feeds two row tiles: half the
// stream) 2.46 / 0.83 with a ring of 4 half-steps, 2.30 / 0.82 with 8 (a lone wave per SIMD has only its own prefetch depth to
// cover the L2 latency).  The exact fp32 kernels measured the other way round (4.66 vs 4.97 ms): they are bound by the matrix pipe.
#ifndef MDX_RR
#define MDX_RR 2
#endif
#ifndef MDX_WPS
#define MDX_WPS 1
#endif
#ifndef MDX_RING
#define MDX_RING 8  // half-steps of 2 KiB in flight per wave
#endif
#include "mdx_row.h"
#ifndef MDX_SPLIT_SEAMLESS
#define MDX_SPLIT_SEAMLESS 1  // the weight ring runs through GEMM boundaries (mdx_split.h)
#endif
#include "mdx_split.h"
#include "mdx_edge2_plan.h"
#include "../../include/moldiff_hip.h"
#include <algorithm>
int mdx_set_error(int code, const char* msg);

// phase trace of the split kernel (tools/trace_edge2s.py); compiled out of the library
#ifdef MDX_TRACE2S
__device__ unsigned long long* mdx_trace2s_buf = nullptr;
extern "C" int mdx_debug_set_trace2s(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace2s_buf), &p, sizeof(p)); }
#define STAMPS(i)                                                                                  \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (lane == 0 && mdx_trace2s_buf) mdx_trace2s_buf[(size_t)unit * 32 + (i)] = clock64();        \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define STAMPS(i) ((void)0)
#endif

namespace {

template <int FLAGS>
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void edge_a2s_kernel(const EdgeAArgs a, const EdgePlan plan, const WorkQA wq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  const int E = a.E;
  constexpr bool do_emb = FLAGS & EA_EMB, do_node = FLAGS & EA_NODE, do_ffn = FLAGS & EA_FFN, do_agg = FLAGS & EA_AGG;
  constexpr bool do_tape = FLAGS & EA_TAPE;
  constexpr bool do_tape_ffn = FLAGS & EA_TAPE_FFN;
  constexpr bool TAPE_NT = do_tape;
#define TAPE_ST(p, v) do { if (TAPE_NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else stg4(p, v); } while (0)
  // Work item u of an EA_AGG launch = the RR consecutive graph-aligned 16-row units RR u .. RR u + RR - 1 of the plan's table, one
  // per row tile (a wave with two row tiles shares every weight fragment between two units: half the weight stream per edge).
  // Each row tile keeps its own unit's rows, row count and partial-row index, so the partial rows -- and every result -- are the
  // same whatever RR is.
  int ucnt_next[RR];
  auto tile_of = [&](int u) {
    if constexpr (do_agg) {
      RowTile t;
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        const int uu = min(RR * u + rt, a.nunits - 1);
        const int2 ue = reinterpret_cast<const int2*>(a.units)[uu];
        const int cnt = (RR * u + rt < a.nunits) ? ue.y : 0;
        ucnt_next[rt] = cnt;
        t.valid[rt] = c < cnt;
        t.row[rt] = t.valid[rt] ? ue.x + c : ue.x;  // clamped to the unit's first row (always a row of the same graph)
        t.li[rt] = a.l[t.row[rt]];
        t.ri[rt] = a.r[t.row[rt]];
        t.tt[rt] = a.te[t.row[rt]];
        t.pf[rt] = a.epo[t.row[rt]];
      }
      t.cnt = ucnt_next[0];
      return t;
    } else {
      return load_tile(a.l, a.r, a.te, u * ROWS, E, c);
    }
  };
  f32x4* park = reinterpret_cast<f32x4*>(smem + (size_t)wave * PARK_FLOATS) + lane;
  const unsigned lane_off = 16u * lane;
  auto W = [&](const float* p) { return make_ws(p, lane_off); };
  const EdgeAS& S = a.w.ss;  // split stream packs

  // LDS constants as in edge_a2_kernel
  float* cb = smem + 4 * PARK_FLOATS;
  const float* c_soff = lds_put<0, 16>(cb, a.soff, tid);
  const float* c_scoef = lds_put<16, 16>(cb, a.scoef, tid);
  const float* c_bemb = cb + 32;
  if (do_emb) lds_put<32, 64>(cb, a.w.bemb, tid);
  const float *c_bg1 = cb + 96, *c_wtg1 = c_bg1 + 256, *c_gg = c_bg1 + 512, *c_gb = c_bg1 + 768, *c_bg2 = c_bg1 + 1024,
              *c_eb1 = c_bg1 + 1280, *c_eg = c_bg1 + 1536, *c_ebe = c_bg1 + 1792, *c_eb2 = c_bg1 + 2048, *c_bm = c_bg1 + 2304;
  if (do_node) {
    lds_put<96, 256>(cb, a.w.bg1, tid); lds_put<96 + 256, 256>(cb, a.w.wtg1, tid); lds_put<96 + 512, 256>(cb, a.w.gg, tid);
    lds_put<96 + 768, 256>(cb, a.w.gb, tid); lds_put<96 + 1024, 256>(cb, a.w.bg2, tid); lds_put<96 + 1280, 256>(cb, a.w.en.b1, tid);
    lds_put<96 + 1536, 256>(cb, a.w.en.g, tid); lds_put<96 + 1792, 256>(cb, a.w.en.be, tid);
    lds_put<96 + 2048, 256>(cb, a.w.en.b2, tid); lds_put<96 + 2304, 256>(cb, a.w.bm, tid);
  }
  constexpr int FO = 96 + 2560, FS = 640;  // per BondFFN: bg1 32 | wtg1 32 | gg 32 | gb 32 | ib1 128 | ig 128 | ibe 128 | ib2 64 | bg2 64
  const float *f_bg1[2], *f_wtg1[2], *f_gg[2], *f_gb[2], *f_ib1[2], *f_ig[2], *f_ibe[2], *f_ib2[2], *f_bg2[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float* fb = cb + FO + FS * s;
    f_bg1[s] = fb; f_wtg1[s] = fb + 32; f_gg[s] = fb + 64; f_gb[s] = fb + 96; f_ib1[s] = fb + 128; f_ig[s] = fb + 256;
    f_ibe[s] = fb + 384; f_ib2[s] = fb + 512; f_bg2[s] = fb + 576;
  }
  if (do_ffn) {
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

  const bool dyn = wq.q.ctr != nullptr;
  const int slot = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  WorkPair wp{};
  int nitems, xcnt = 0, xnt = 0, mode, mode_next, unit;
  if (dyn) {
    wp = wq_pair(wq.q);
    xcnt = wp.end - wp.beg;
    xnt = min(xcnt, wp.waves * wq.tail8 >> 3);
    const int i0 = wq_take(wq_request(wp.line, lane));
    nitems = i0 < xcnt + xnt ? 1 : 0;
    unit = wp.beg + wq_item(i0, xcnt, xnt, mode);
  } else {
    nitems = plan_items(plan, slot);
    unit = plan_item(plan, slot, 0, mode);
  }
  if (nitems <= 0) {
    if (dyn) wq_leave(wp, lane);
    return;
  }

  const float* wfirst = do_emb ? S.Wemb : do_node ? S.Wg1e : S.ffn[0].Wbl;
  WRing ring;
  ring_prime(ring, W(wfirst));
  Prolog pr;
  pr.t = tile_of(unit);
  prolog_rows(pr, a, q0);

#pragma unroll 1
  for (int it = 0;; ++it) {
    int q = q0;
    asm volatile("" : "+v"(q));
    const RowTile t = pr.t;
    int ucnt[RR];
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) ucnt[rt] = do_agg ? __builtin_amdgcn_readfirstlane(ucnt_next[rt]) : 0;
    const int ureq = dyn ? wq_request(wp.line, lane) : 0;
    const bool inode = do_node && (mode & 1), iffn = do_ffn && (mode & 10);
    const int sfirst = (mode & 2) ? 0 : 1, slast = (mode & 8) ? 1 : 0;
    // ---- He' = edge_embs([He | smear(d)]) ; hx = its split operand (the fp32 rows are only needed for the store) ------------
    STAMPS(0);
    XS<2> hx;
    {
      f32x4 hep[4][RR];
      if (do_emb) {
        f32x4 x[5][RR];
        const f32x4 off = lds4(c_soff + 4 * q), coef = lds4(c_scoef + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) {
#pragma unroll
          for (int g = 0; g < 4; ++g) x[g][rt] = pr.x[g][rt];
          const float u0 = fminf(fmaxf(pr.d[rt], a.smear_start), a.cutoff);
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float u = u0 - off[s];
            x[4][rt][s] = expf(coef[s] * (u * u));
          }
        }
        XS<3> xs;  // K = 80, zero-padded to 96
        to_xs<5>(xs, x);
        row_bias<4, RR>(hep, c_bemb, q);
        rgemm_s<3, 4>(hep, xs, W(S.Wemb), ring, W(inode ? S.Wg1e : iffn ? S.ffn[sfirst].Wbl : wfirst));
        STAMPS(1);
        if (mode & 4) row_store<4, RR>(hep, a.He_out, t.row, t.valid, 64, q);
      } else {
#pragma unroll
        for (int rt = 0; rt < RR; ++rt)
#pragma unroll
          for (int g = 0; g < 4; ++g) hep[g][rt] = pr.x[g][rt];
      }
      to_xs<4>(hx, hep);
    }

    // ---- NodeBlock message path: M = msg_net(edge_net(He') * h[r]) * sigmoid(gate([He' | x[r] | t])) --------
    if (inode) {
      f32x4 y[16][RR], z[16][RR];
      {  // gate layer 1: accumulator starts at b + gx[r] + t*wt (the hoisted node part and the time column)
        row_gather<16, RR>(y, a.NT + MDX_NT_GX, t.ri, MDX_NTW, q);
        float tg[RR];
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) tg[rt] = a.tn_r ? a.tn_r[t.row[rt]] : t.tt[rt];
#pragma unroll
        for (int ft = 0; ft < 16; ++ft) {
          const f32x4 b = lds4(c_bg1 + 16 * ft + 4 * q), wt = lds4(c_wtg1 + 16 * ft + 4 * q);
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) y[ft][rt] = (b + y[ft][rt]) + splat4(tg[rt]) * wt;
        }
      }
      STAMPS(2);
      rgemm_s<2, 16>(y, hx, W(S.Wg1e), ring, W(S.Wg2));
      STAMPS(3);
      row_layernorm<16, RR>(y, c_gg, c_gb, q);
      {
        XS<8> ys;
        to_xs<16>(ys, y);
        row_bias<16, RR>(z, c_bg2, q);
        STAMPS(4);
        rgemm_s<8, 16>(z, ys, W(S.Wg2), ring, W(S.W1));
        STAMPS(5);
      }
#pragma unroll
      for (int ft = 0; ft < 16; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) {
          const f32x4 sg = row_sigmoid4(z[ft][rt]);
          if (a.tSG && t.valid[rt]) TAPE_ST(a.tSG + (size_t)t.row[rt] * MDX_ND + 16 * ft + 4 * q, sg);
          park[(ft * RR + rt) * 64] = sg;
        }
      // edge_net
      row_bias<16, RR>(y, c_eb1, q);
      STAMPS(6);
      rgemm_s<2, 16>(y, hx, W(S.W1), ring, W(S.W2));
      STAMPS(7);
      row_layernorm<16, RR>(y, c_eg, c_ebe, q);
      {
        XS<8> ys;
        to_xs<16>(ys, y);
        row_bias<16, RR>(z, c_eb2, q);
        STAMPS(8);
        rgemm_s<8, 16>(z, ys, W(S.W2), ring, W(S.Wm));
        STAMPS(9);
      }
      if (a.tHE) row_store<16, RR, TAPE_NT>(z, a.tHE, t.row, t.valid, MDX_ND, q);
      row_gather<16, RR>(y, a.H, t.ri, MDX_ND, q);
      mul_inplace<16>(z, y);
      // msg_net, gated
      {
        XS<8> zs;
        to_xs<16>(zs, z);
        row_bias<16, RR>(y, c_bm, q);
        STAMPS(10);
        rgemm_s<8, 16>(y, zs, W(S.Wm), ring, W(iffn ? S.ffn[sfirst].Wbl : wfirst));
        STAMPS(11);
      }
#pragma unroll
      for (int ft = 0; ft < 16; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) y[ft][rt] = y[ft][rt] * park[(ft * RR + rt) * 64];
      if constexpr (do_agg) {
        if (a.M) row_store<16, RR, TAPE_NT>(y, a.M, t.row, t.valid, MDX_ND, q);
        static_for<0, RR>([&](auto rc) {
          constexpr int rt = decltype(rc)::value;
          seg_sum_store<16, RR, rt>(y, smem + (size_t)wave * PARK_FLOATS, lane, ucnt[rt], t.li[rt], t.pf[rt] + RR * unit + rt, a.P);
        });
      } else {
        row_store<16, RR>(y, a.M, t.row, t.valid, MDX_ND, q);
      }
    }
    int unext;
    bool more;
    if (dyn) {
      const int i = wq_take(ureq);
      more = i < xcnt + xnt;
      mode_next = mode;
      unext = more ? wp.beg + wq_item(i, xcnt, xnt, mode_next) : unit;
    } else {
      more = it + 1 < nitems;
      unext = plan_item(plan, slot, min(it + 1, nitems - 1), mode_next);
    }
    pr.t = tile_of(unext);
    prolog_rows(pr, a, q);

    // ---- EdgeBlock BondFFNs: F_s = inter_s((W_bl He') * nl_s[idx_s]) * sigmoid(gate_s([He' | x[idx_s] | t])) ----
    if (iffn) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (!(mode & (s ? 8 : 2))) continue;
        const FfnS& ws = S.ffn[s];
        int idx[RR];
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) idx[rt] = s ? t.ri[rt] : t.li[rt];
        f32x4 bl[8][RR], nl[8][RR], g1[2][RR];
        row_gather<8, RR>(nl, a.NT + (s ? MDX_NT_NLR : MDX_NT_NLL), idx, MDX_NTW, q);
        row_gather<2, RR>(g1, a.NT + (s ? MDX_NT_GXR : MDX_NT_GXL), idx, MDX_NTW, q);
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
          const f32x4 b = lds4(f_bg1[s] + 16 * ft + 4 * q), wt = lds4(f_wtg1[s] + 16 * ft + 4 * q);
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) g1[ft][rt] = (b + g1[ft][rt]) + splat4(t.tt[rt]) * wt;
        }
        row_zero<8, RR>(bl);
        rgemm_s<2, 8>(bl, hx, W(ws.Wbl), ring, W(ws.Wg1e));
        if constexpr (do_tape_ffn) row_store<8, RR, TAPE_NT>(bl, a.tBL[s], t.row, t.valid, 128, q);
        mul_inplace<8>(bl, nl);
        rgemm_s<2, 2>(g1, hx, W(ws.Wg1e), ring, W(ws.W1));
        row_layernorm<2, RR>(g1, f_gg[s], f_gb[s], q);
        f32x4 h[8][RR];
        {
          XS<4> bs;
          to_xs<8>(bs, bl);
          row_bias<8, RR>(h, f_ib1[s], q);
          rgemm_s<4, 8>(h, bs, W(ws.W1), ring, W(ws.W2));
        }
        if constexpr (do_tape_ffn) row_store<8, RR, TAPE_NT>(h, a.tH1[s], t.row, t.valid, 128, q);
        row_layernorm<8, RR>(h, f_ig[s], f_ibe[s], q);
        f32x4 o[4][RR], g2[4][RR];
        {
          XS<4> hs;
          to_xs<8>(hs, h);
          row_bias<4, RR>(o, f_ib2[s], q);
          rgemm_s<4, 4>(o, hs, W(ws.W2), ring, W(ws.Wg2));
        }
        if constexpr (do_tape_ffn) row_store<4, RR, TAPE_NT>(o, a.tO[s], t.row, t.valid, 64, q);
        {
          XS<1> gs;
          to_xs<2>(gs, g1);
          row_bias<4, RR>(g2, f_bg2[s], q);
          rgemm_s<1, 4>(g2, gs, W(ws.Wg2), ring, W(s < slast ? S.ffn[1].Wbl : wfirst));
        }
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) o[ft][rt] = o[ft][rt] * row_sigmoid4(g2[ft][rt]);
        if constexpr (do_agg) {
          if (s == 1) {
            if (a.F[1]) row_store<4, RR>(o, a.F[1], t.row, t.valid, 64, q);
            static_for<0, RR>([&](auto rc) {
              constexpr int rt = decltype(rc)::value;
              seg_sum_store<4, RR, rt>(o, smem + (size_t)wave * PARK_FLOATS, lane, ucnt[rt], t.li[rt], t.pf[rt] + RR * unit + rt, a.PR);
            });
          } else {
            row_store<4, RR>(o, a.F[s], t.row, t.valid, 64, q);
          }
        } else {
          row_store<4, RR>(o, a.F[s], t.row, t.valid, 64, q);
        }
      }
    }
    if (!more) break;
    unit = unext;
    mode = mode_next;
  }
  if (dyn) wq_leave(wp, lane);
#undef TAPE_ST
}

}  // namespace

template <int FLAGS>
static void launch_a2s(const EdgeAArgs& a, hipStream_t s) {
  static bool attr = false;
  constexpr int lds = (4 * PARK_FLOATS + EA_CONST_FLOATS) * 4;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)edge_a2s_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int nunits = (FLAGS & EA_AGG) ? (a.nunits + RR - 1) / RR : (a.E + ROWS - 1) / ROWS;   // work items (RR units each)
  if (nunits <= 0) return;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  constexpr bool all = (FLAGS & ~(EA_AGG | EA_TAPE | EA_TAPE_FFN)) == (EA_EMB | EA_NODE | EA_FFN);
  const EdgePlan plan = make_plan(nunits, grid * 4, all);
  WorkQA wq{};
  wq.q = make_workq(a.wq, nunits, grid, mdx_num_cus());
  if (a.wq && all) wq.tail8 = 10;  // section-cut tail of each pair's list, as in the exact kernel
  hipLaunchKernelGGL(edge_a2s_kernel<FLAGS>, dim3(grid), dim3(MDX_WG), lds, s, a, plan, wq);
}

// same dispatch contract as launch_edge_a2 (flags without EA_SPLIT)
int launch_edge_a2s(const EdgeAArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  switch (a.flags & ~EA_SPLIT) {
    case EA_EMB | EA_NODE | EA_FFN | EA_AGG: launch_a2s<EA_EMB | EA_NODE | EA_FFN | EA_AGG>(a, s); return MDX_OK;
    case EA_EMB | EA_NODE | EA_FFN | EA_AGG | EA_TAPE | EA_TAPE_FFN:
      launch_a2s<EA_EMB | EA_NODE | EA_FFN | EA_AGG | EA_TAPE | EA_TAPE_FFN>(a, s); return MDX_OK;
    default: return mdx_set_error(MDX_ERR_UNSUPPORTED, "split-precision edge kernel A: unsupported section flags");
  }
}
