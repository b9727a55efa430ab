// Row-owner fused per-edge kernels of one NodeEdgeNet block (gfx950) -- round 2 replacement for mdx_edge.hip's tile kernels.
//
// Same math, same argument blocks (EdgeAArgs / EdgeBArgs), different work decomposition (see mdx_row.h):
//   edge kernel A  (reference models/graph.py:352-357 edge_embs, :42-47 NodeBlock message path,
//                   :133-141/:278,:282 the two EdgeBlock BondFFNs)
//   edge kernel B  (models/graph.py:286-294 EdgeBlock tail, :384-393 PosUpdate)
// One WAVE owns 16*R consecutive edges of the (left,right)-sorted edge list and computes every layer for them; its
// activations stay in registers from the He tile load to the M / F / He'' / Fe stores.  No LDS tile, no barrier.
// LDS is only a wave-private parking area for sigmoid(gate) while the message chain runs (16 KiB per wave at 16 rows).
#include "mdx_kernels.h"
#ifndef MDX_RING
#define MDX_RING 4  // weight-ring depth in steps (kernel A: 4.65 / 4.58 / 4.56 ms per step at depth 2 / 3 / 4)
#endif
#include "mdx_row.h"
#include "mdx_edge2_plan.h"
#include "../../include/moldiff_hip.h"
#include <algorithm>
int mdx_set_error(int code, const char* msg);

// Phase trace (tools/trace_edge2.py; build with EXTRA=-DMDX_TRACE2): lane 0 of every wave stamps the shader clock at each
// phase boundary into a 48-slot record per unit (slot 46/47: 100 MHz wall clock at entry/exit).  Compiled out of the library.
#ifdef MDX_TRACE2
__device__ unsigned long long* mdx_trace2_buf = nullptr;
__device__ int mdx_trace2_sel = 0;  // 0: edge_a2, 1: edge_b2
extern "C" int mdx_debug_set_trace2(void* p, int which) {
  hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace2_sel), &which, sizeof(which));
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace2_buf), &p, sizeof(p));
}
#define STAMP_K(k, i)                                                                                       \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (lane == 0 && mdx_trace2_buf && mdx_trace2_sel == (k))                                               \
      mdx_trace2_buf[(size_t)unit * 48 + (i)] = ((i) >= 46) ? wall_clock64() : clock64();                   \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
#else
#define STAMP_K(k, i) ((void)0)
#endif
#define STAMP(i) STAMP_K(0, i)
#define STAMPB(i) STAMP_K(1, i)

namespace {

// FLAGS (EA_*) is a template parameter: with run-time section flags every section sits behind a branch and the values that
// cross it (He', the tile, the weight ring) get spilled around the control flow.
template <int FLAGS>
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void edge_a2_kernel(const EdgeAArgs a, const EdgePlan plan, const WorkQA wq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  const int E = a.E;
  constexpr bool do_emb = FLAGS & EA_EMB, do_node = FLAGS & EA_NODE, do_ffn = FLAGS & EA_FFN, do_agg = FLAGS & EA_AGG;
  // The BondFFN tape stores (round 3) are compiled into the tape instantiation only.  Round 2's three stores (tSG, tHE, M / F[1]
  // with EA_AGG) stay behind their run-time pointer tests in every instantiation: compiling them out of the denoiser's kernel as
  // well measured 0.8 % SLOWER on the same box (6.94 vs 6.89 ms per step; the register allocation of a 256-VGPR kernel is that
  // sensitive -- 5 spilled registers instead of 4), so the form that measured best ships.
  constexpr bool do_tape = FLAGS & EA_TAPE;
  constexpr bool do_tape_ffn = FLAGS & EA_TAPE_FFN;  // unconditional stores: no pointer tests inside the BondFFN sections
  // the tape is read back by the backward a whole forward later: streaming (non-temporal) stores keep its 870 MB per launch from
  // displacing weights and node rows in L2 (guided step 26.00 -> 25.90 ms)
  constexpr bool TAPE_NT = do_tape;
#define TAPE_ST(p, v) do { if (TAPE_NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); else stg4(p, v); } while (0)
  static_assert(!do_agg || RR == 1, "the in-kernel segment sums are written for one 16-row tile per wave");
  // rows of unit u: graph-aligned units from the plan's table (EA_AGG) or 16 consecutive rows of the batch
  auto tile_of = [&](int u) {
    if constexpr (do_agg) {
      const int2 ue = reinterpret_cast<const int2*>(a.units)[u];
      return load_tile_u(a.l, a.r, a.te, a.epo, ue.x, ue.y, E, c);
    } else {
      return load_tile(a.l, a.r, a.te, u * ROWS, E, c);
    }
  };
  f32x4* park = reinterpret_cast<f32x4*>(smem + (size_t)wave * PARK_FLOATS) + lane;
  const unsigned lane_off = 16u * lane;
  auto W = [&](const float* p) { return make_ws(p, lane_off); };

  // fixed LDS layout (offsets in floats) so every constant address is cbase + immediate
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
  __syncthreads();  // the only barrier of the kernel: constants visible to every wave

  // persistent wave; slots are XCD-contiguous so that neighbouring units (same molecule -> same node rows) share an L2
  // (static split; with a work queue the wave draws its items from its pair's counter instead)
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

  // first stream of a unit (the tail of every unit primes it again for the next one)
  const float* wfirst = do_emb ? a.w.s.Wemb : do_node ? a.w.s.Wg1e : a.w.s.ffn[0].Wbl;
  WRing ring;
  ring_prime(ring, W(wfirst));
  Prolog pr;
  pr.t = tile_of(unit);
  prolog_rows(pr, a, q0);

#pragma unroll 1
  for (int it = 0;; ++it) {
    int q = q0;
    asm volatile("" : "+v"(q));  // opaque per iteration: lane-dependent address parts stay next to their loads instead of
                                 // being hoisted out of the persistent loop into (spilled) registers
    const RowTile t = pr.t;
    const int ucnt = do_agg ? __builtin_amdgcn_readfirstlane(t.cnt) : 0;
    const int ureq = dyn ? wq_request(wp.line, lane) : 0;  // the next item, consumed where its tile is requested
    const bool inode = do_node && (mode & 1), iffn = do_ffn && (mode & 10);
    const int sfirst = (mode & 2) ? 0 : 1, slast = (mode & 8) ? 1 : 0;  // BondFFN sections of this item (wave-uniform)
    STAMP(46);
    STAMP(0);
    // ---- He' = edge_embs([He | smear(d)]) -------------------------------------------------------
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
      row_bias<4, RR>(hep, c_bemb, q);
      STAMP(1);
      rgemm<5, 4, RR>(hep, x, W(a.w.s.Wemb), ring, W(inode ? a.w.s.Wg1e : iffn ? a.w.s.ffn[sfirst].Wbl : wfirst));
      STAMP(2);
      if (mode & 4) row_store<4, RR>(hep, a.He_out, t.row, t.valid, 64, q);
    } else {
#pragma unroll
      for (int rt = 0; rt < RR; ++rt)
#pragma unroll
        for (int g = 0; g < 4; ++g) hep[g][rt] = pr.x[g][rt];
    }

    // ---- NodeBlock message path: M = msg_net(edge_net(He') * h[r]) * sigmoid(gate([He' | x[r] | t])) --------
    if (inode) {
      f32x4 y[16][RR], z[16][RR];
      {  // gate layer 1: accumulator starts at b + gx[r] + t*wt (the hoisted node part and the time column)
        row_gather<16, RR>(y, a.NT + MDX_NT_GX, t.ri, MDX_NTW, q);
        float tg[RR];
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) tg[rt] = a.tn_r ? a.tn_r[t.row[rt]] : t.tt[rt];  // the NodeBlock gate sees node_time[col]
#pragma unroll
        for (int ft = 0; ft < 16; ++ft) {
          const f32x4 b = lds4(c_bg1 + 16 * ft + 4 * q), wt = lds4(c_wtg1 + 16 * ft + 4 * q);
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) y[ft][rt] = (b + y[ft][rt]) + splat4(tg[rt]) * wt;
        }
      }
      STAMP(3);
      rgemm<4, 16, RR>(y, hep, W(a.w.s.Wg1e), ring, W(a.w.s.Wg2));
      STAMP(4);
      row_layernorm<16, RR>(y, c_gg, c_gb, q);
      row_bias<16, RR>(z, c_bg2, q);
      STAMP(5);
      rgemm<16, 16, RR>(z, y, W(a.w.s.Wg2), ring, W(a.w.s.W1));
      STAMP(6);
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
      STAMP(7);
      rgemm<4, 16, RR>(y, hep, W(a.w.s.W1), ring, W(a.w.s.W2));
      STAMP(8);
      row_layernorm<16, RR>(y, c_eg, c_ebe, q);
      row_bias<16, RR>(z, c_eb2, q);
      STAMP(9);
      rgemm<16, 16, RR>(z, y, W(a.w.s.W2), ring, W(a.w.s.Wm));
      STAMP(10);
      if (a.tHE) row_store<16, RR, TAPE_NT>(z, a.tHE, t.row, t.valid, MDX_ND, q);
      row_gather<16, RR>(y, a.H, t.ri, MDX_ND, q);
      mul_inplace<16>(z, y);
      // msg_net, gated
      row_bias<16, RR>(y, c_bm, q);
      STAMP(11);
      rgemm<16, 16, RR>(y, z, W(a.w.s.Wm), ring, W(iffn ? a.w.s.ffn[sfirst].Wbl : wfirst));
      STAMP(12);
#pragma unroll
      for (int ft = 0; ft < 16; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) y[ft][rt] = y[ft][rt] * park[(ft * RR + rt) * 64];
      if constexpr (do_agg) {
        // aggr[v] = sum over v's edge run of M (models/graph.py:50), the part of it that lies in this unit: segmented sum over
        // the tile's rows, one partial row per left node stored by the last row of its segment
        if (a.M) row_store<16, RR, TAPE_NT>(y, a.M, t.row, t.valid, MDX_ND, q);  // the guidance tape keeps M itself
        seg_sum_store<16>(y, smem + (size_t)wave * PARK_FLOATS, lane, ucnt, t.li[0], t.pf[0] + unit, a.P);
      } else {
        row_store<16, RR>(y, a.M, t.row, t.valid, MDX_ND, q);
      }
      STAMP(13);
    }
    // next unit's tile: indices, He rows and edge lengths are requested here, ahead of the BondFFN sections, and arrive under
    // their GEMMs (requested inside a section they would sit behind that section's run-time condition: spills)
    int unext;
    bool more;
    if (dyn) {
      const int i = wq_take(ureq);
      more = i < xcnt + xnt;
      mode_next = mode;
      unext = more ? wp.beg + wq_item(i, xcnt, xnt, mode_next) : unit;  // last item: the look-ahead repeats this unit's rows
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
        const FfnS& ws = a.w.s.ffn[s];
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
        STAMP(14 + 10 * s);
        rgemm<4, 8, RR>(bl, hep, W(ws.Wbl), ring, W(ws.Wg1e));
        STAMP(15 + 10 * s);
        if constexpr (do_tape_ffn) row_store<8, RR, TAPE_NT>(bl, a.tBL[s], t.row, t.valid, 128, q);
        mul_inplace<8>(bl, nl);
        rgemm<4, 2, RR>(g1, hep, W(ws.Wg1e), ring, W(ws.W1));
        STAMP(16 + 10 * s);
        row_layernorm<2, RR>(g1, f_gg[s], f_gb[s], q);
        f32x4 h[8][RR];
        row_bias<8, RR>(h, f_ib1[s], q);
        STAMP(17 + 10 * s);
        rgemm<8, 8, RR>(h, bl, W(ws.W1), ring, W(ws.W2));
        STAMP(18 + 10 * s);
        if constexpr (do_tape_ffn) row_store<8, RR, TAPE_NT>(h, a.tH1[s], t.row, t.valid, 128, q);
        row_layernorm<8, RR>(h, f_ig[s], f_ibe[s], q);
        f32x4 o[4][RR], g2[4][RR];
        row_bias<4, RR>(o, f_ib2[s], q);
        STAMP(19 + 10 * s);
        rgemm<8, 4, RR>(o, h, W(ws.W2), ring, W(ws.Wg2));
        STAMP(20 + 10 * s);
        if constexpr (do_tape_ffn) row_store<4, RR, TAPE_NT>(o, a.tO[s], t.row, t.valid, 64, q);
        row_bias<4, RR>(g2, f_bg2[s], q);
        rgemm<2, 4, RR>(g2, g1, W(ws.Wg2), ring, W(s < slast ? a.w.s.ffn[1].Wbl : wfirst));
        STAMP(21 + 10 * s);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) o[ft][rt] = o[ft][rt] * row_sigmoid4(g2[ft][rt]);
        if constexpr (do_agg) {
          if (s == 1) {  // SR[v] = sum over v's edge run of bond_ffn_right (graph.py:283): same segments as M
            if (a.F[1]) row_store<4, RR>(o, a.F[1], t.row, t.valid, 64, q);
            seg_sum_store<4>(o, smem + (size_t)wave * PARK_FLOATS, lane, ucnt, t.li[0], t.pf[0] + unit, a.PR);
          } else {
            row_store<4, RR>(o, a.F[s], t.row, t.valid, 64, q);
          }
        } else {
          row_store<4, RR>(o, a.F[s], t.row, t.valid, 64, q);
        }
        STAMP(22 + 10 * s);
      }
    }
    STAMP(40);
    STAMP(47);
    if (!more) break;
    unit = unext;
    mode = mode_next;
  }
  if (dyn) wq_leave(wp, lane);
}

}  // namespace

int mdx_num_cus() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
  }();
  return n;
}

template <int FLAGS>
static void launch_a2(const EdgeAArgs& a, hipStream_t s) {
  static bool attr = false;
  constexpr int lds = (4 * PARK_FLOATS + EA_CONST_FLOATS) * 4;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)edge_a2_kernel<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int nunits = (FLAGS & EA_AGG) ? a.nunits : (a.E + ROWS - 1) / ROWS;
  if (nunits <= 0) return;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  constexpr bool all = (FLAGS & ~(EA_AGG | EA_TAPE | EA_TAPE_FFN)) == (EA_EMB | EA_NODE | EA_FFN);
  const EdgePlan plan = make_plan(nunits, grid * 4, all);
  WorkQA wq{};
  wq.q = make_workq(a.wq, nunits, grid, mdx_num_cus());
  if (a.wq && all) wq.tail8 = 10;  // units cut by section at the end of each pair's list: 10 eighths of a unit per wave (measured:
                                   // none 6.99 ms per step, 8 eighths 6.92, 10 6.91, 16 7.01)
  hipLaunchKernelGGL(edge_a2_kernel<FLAGS>, dim3(grid), dim3(MDX_WG), lds, s, a, plan, wq);
}

int launch_edge_a2(const EdgeAArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  switch (a.flags) {
    case EA_EMB | EA_NODE | EA_FFN | EA_AGG: launch_a2<EA_EMB | EA_NODE | EA_FFN | EA_AGG>(a, s); return MDX_OK;  // product path
    case EA_EMB | EA_NODE | EA_FFN | EA_AGG | EA_TAPE | EA_TAPE_FFN:
      launch_a2<EA_EMB | EA_NODE | EA_FFN | EA_AGG | EA_TAPE | EA_TAPE_FFN>(a, s); return MDX_OK;                     // + BondFFN tape
    case EA_NODE: launch_a2<EA_NODE>(a, s); return MDX_OK;                                    // NodeBlock.forward
    case EA_FFN: launch_a2<EA_FFN>(a, s); return MDX_OK;                                      // EdgeBlock.forward
    default: return mdx_set_error(MDX_ERR_UNSUPPORTED, "edge kernel A: unsupported section flags");
  }
}

// the entry point the host API uses: exact fp32 build or, with EA_SPLIT in the flags, the split float16 build (mdx_edge2s.hip)
int launch_edge_a(const EdgeAArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  return (a.flags & EA_SPLIT) ? launch_edge_a2s(a, s) : launch_edge_a2(a, s);
}
