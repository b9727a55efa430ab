// Row-owner edge kernel B of one NodeEdgeNet block (gfx950): EdgeBlock tail + PosUpdate
// (reference models/graph.py:286-294 and :384-393).  Design notes: mdx_edge2.hip / mdx_row.h.
#include "mdx_kernels.h"
#ifndef MDX_RING
#define MDX_RING 4  // weight-ring depth in steps (kernel B: 1.743 / 1.727 / 1.723 ms per step at depth 2 / 3 / 4)
#endif
#include "mdx_row.h"
#include "../../include/moldiff_hip.h"
#include <algorithm>
int mdx_set_error(int code, const char* msg);

// phase trace, see mdx_edge2.hip (tools/trace_edge2.py b); compiled out of the library
#ifdef MDX_TRACE2
__device__ unsigned long long* mdx_trace2b_buf = nullptr;
extern "C" int mdx_debug_set_trace2b(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace2b_buf), &p, sizeof(p)); }
#define STAMPB(i)                                                                                           \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (lane == 0 && mdx_trace2b_buf) mdx_trace2b_buf[(size_t)unit * 48 + (i)] = ((i) >= 46) ? wall_clock64() : clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
  } while (0)
#else
#define STAMPB(i) ((void)0)
#endif

namespace {

constexpr int EB_CONST_FLOATS = 4 * 64 + 5 * 32 + 4 * 256;

struct PrologB {
  RowTile t;
  f32x4 he[4][RR];
};

template <int FLAGS>
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void edge_b2_kernel(const EdgeBArgs a, const int nunits, const WorkQ wq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  const int E = a.E;
  constexpr bool do_edge = FLAGS & EB_EDGE, do_pos = FLAGS & EB_POS;
  const unsigned lane_off = 16u * lane;
  auto W = [&](const float* p) { return make_ws(p, lane_off); };

  float* cb = smem;
  const float *c_bself = cb, *c_lng = cb + 64, *c_lnb = cb + 128, *c_bout = cb + 192;
  if (do_edge) {
    lds_put<0, 64>(cb, a.w.bself, tid); lds_put<64, 64>(cb, a.w.lng, tid); lds_put<128, 64>(cb, a.w.lnb, tid);
    lds_put<192, 64>(cb, a.w.bout, tid);
  }
  const float *c_bg1 = cb + 256, *c_wtg1 = cb + 288, *c_gg = cb + 320, *c_gb = cb + 352, *c_wg2 = cb + 384, *c_bi1 = cb + 416,
              *c_ig = cb + 672, *c_ib = cb + 928, *c_wi2 = cb + 1184;
  if (do_pos) {
    lds_put<256, 32>(cb, a.w.bg1, tid); lds_put<288, 32>(cb, a.w.wtg1, tid); lds_put<320, 32>(cb, a.w.gg, tid);
    lds_put<352, 32>(cb, a.w.gb, tid); lds_put<384, 32>(cb, a.w.wg2, tid); lds_put<416, 256>(cb, a.w.bi1, tid);
    lds_put<672, 256>(cb, a.w.ig, tid); lds_put<928, 256>(cb, a.w.ib, tid); lds_put<1184, 256>(cb, a.w.wi2, tid);
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

  const float* wfirst = do_edge ? a.w.s.Wself : a.w.s.Wbl;
  WRing ring;
  ring_prime(ring, W(wfirst));
  PrologB pr;
  pr.t = load_tile(a.l, a.r, a.te, ubeg * ROWS, E, c);
  row_gather<4, RR>(pr.he, a.Hep, pr.t.row, 64, q0);

#pragma unroll 1
  for (int unit = ubeg;;) {
    int q = q0;
    asm volatile("" : "+v"(q));  // see edge_a2_kernel
    const RowTile t = pr.t;
    const int ureq = dyn ? wq_request(wp.line, lane) : 0;  // consumed where the next tile is loaded
    STAMPB(46);
    STAMPB(0);
    f32x4 he[4][RR];  // He' on entry, He'' after the EdgeBlock tail
#pragma unroll
    for (int rt = 0; rt < RR; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) he[g][rt] = pr.he[g][rt];

    // PosUpdate inputs that only depend on the tile's indices: requested first, consumed after the EdgeBlock tail
    f32x4 aa[4][RR], bb[4][RR];
    float rx[RR], ry[RR], rz[RR], dd[RR];
    if (do_pos) {
      row_gather<4, RR>(aa, a.Lf, t.li, 64, q);
      row_gather<4, RR>(bb, a.Rf, t.ri, 64, q);
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        if (a.rel_in) {
          rx[rt] = a.rel_in[3 * (size_t)t.row[rt] + 0]; ry[rt] = a.rel_in[3 * (size_t)t.row[rt] + 1]; rz[rt] = a.rel_in[3 * (size_t)t.row[rt] + 2];
          dd[rt] = a.dist_in[t.row[rt]];
        } else {
          rx[rt] = a.pos[3 * t.li[rt] + 0] - a.pos[3 * t.ri[rt] + 0];
          ry[rt] = a.pos[3 * t.li[rt] + 1] - a.pos[3 * t.ri[rt] + 1];
          rz[rt] = a.pos[3 * t.li[rt] + 2] - a.pos[3 * t.ri[rt] + 2];
          dd[rt] = sqrtf(rx[rt] * rx[rt] + ry[rt] * ry[rt] + rz[rt] * rz[rt]);
        }
      }
    }

    // ---- EdgeBlock tail: He'' = He' + out_transform(relu(LN(SL[l] + SR[r] + nfl[l] + nfr[r] + self_ffn(He')))) ----
    if (do_edge) {
      f32x4 u[4][RR], v[4][RR], v2[4][RR], v3[4][RR];
      row_gather<4, RR>(u, a.SL, t.li, 64, q);
      row_gather<4, RR>(v, a.SR, t.ri, 64, q);
      row_gather<4, RR>(v2, a.NT + MDX_NT_NFL, t.li, MDX_NTW, q);
      row_gather<4, RR>(v3, a.NT + MDX_NT_NFR, t.ri, MDX_NTW, q);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft) {
        const f32x4 bs = lds4(c_bself + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) u[ft][rt] = (((u[ft][rt] + v[ft][rt]) + v2[ft][rt]) + v3[ft][rt]) + bs;
      }
      STAMPB(1);
      rgemm<4, 4, RR>(u, he, W(a.w.s.Wself), ring, W(a.w.s.Wout));
      STAMPB(2);
      row_layernorm<4, RR>(u, c_lng, c_lnb, q);
      row_bias<4, RR>(v, c_bout, q);
      STAMPB(3);
      rgemm<4, 4, RR>(v, u, W(a.w.s.Wout), ring, W(do_pos ? a.w.s.Wbl : wfirst));
      STAMPB(4);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) {
          if (!(FLAGS & EB_DELTA)) v[ft][rt] = v[ft][rt] + he[ft][rt];
          he[ft][rt] = v[ft][rt];
        }
      row_store<4, RR>(he, a.He_out, t.row, t.valid, 64, q);
    }
    int unext = dyn ? wp.beg + wq_take(ureq) : unit + 1;
    const bool more = unext < uend;
    if (!more) unext = unit;  // last unit of the wave: the look-ahead loads repeat this unit's rows
    pr.t = load_tile(a.l, a.r, a.te, unext * ROWS, E, c);  // next unit's indices travel under the PosUpdate GEMMs

    // ---- PosUpdate: w = inter((W_bl He'') * (W_nl a)) * sigmoid(gate([He'' | a | t])), a = Lf[l] * Rf[r]; Fe = w rel / d / (d+1) ----
    if (do_pos) {
      mul_inplace<4>(aa, bb);
      f32x4 x[16][RR], h[16][RR], g1[2][RR];
      // x = (W_bl He'') * (W_nl a), formed one pair of feature tiles at a time from the two streams alternately, so the two
      // (rows x 256) products are never both live.  (Written as one 64->256 GEMM followed by pairwise 64->32 GEMMs the compiler
      // sinks each pair's W_bl MFMAs down to their use -- same schedule, but with the weight ring loaded in the old order and
      // spilled; stated explicitly, the ring order is the execution order.)
      STAMPB(5);
      static_for<0, 8>([&](auto fc) {
        constexpr int ftp = decltype(fc)::value;
        f32x4 xb[2][RR], xn[2][RR];
        row_zero<2, RR>(xb);
        rgemm<4, 2, RR>(xb, he, W(a.w.s.Wbl + ftp * 2048), ring, W(a.w.s.Wnl + ftp * 2048));
        row_zero<2, RR>(xn);
        rgemm<4, 2, RR>(xn, aa, W(a.w.s.Wnl + ftp * 2048), ring, W(ftp < 7 ? a.w.s.Wbl + (ftp + 1) * 2048 : a.w.s.Wg1h));
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rt = 0; rt < RR; ++rt) x[2 * ftp + j][rt] = xb[j][rt] * xn[j][rt];
      });
      STAMPB(6);
      STAMPB(7);
      // gate: ((b + t wt) + W_h He'') + W_a a, LN(32), ReLU, 32 -> 1
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const f32x4 b = lds4(c_bg1 + 16 * ft + 4 * q), wt = lds4(c_wtg1 + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) g1[ft][rt] = b + splat4(t.tt[rt]) * wt;
      }
      rgemm<4, 2, RR>(g1, he, W(a.w.s.Wg1h), ring, W(a.w.s.Wg1a));
      rgemm<4, 2, RR>(g1, aa, W(a.w.s.Wg1a), ring, W(a.w.s.Wi1));
      STAMPB(8);
      row_layernorm<2, RR>(g1, c_gg, c_gb, q);
      float gate[RR], wd[RR];
      row_dot<2, RR>(g1, c_wg2, q, gate);
      row_bias<16, RR>(h, c_bi1, q);
      STAMPB(9);
      rgemm<16, 16, RR>(h, x, W(a.w.s.Wi1), ring, W(wfirst));
      STAMPB(10);
      row_gather<4, RR>(pr.he, a.Hep, pr.t.row, 64, q);  // next unit's He' rows: their latency hides under the LayerNorm below
      row_layernorm<16, RR>(h, c_ig, c_ib, q);
      row_dot<16, RR>(h, c_wi2, q, wd);
      if (q == 0) {
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) {
          if (!t.valid[rt]) continue;
          const float w = (wd[rt] + a.w.bi2) * sigmoidf_(gate[rt] + a.w.bg2);
          const float d = dd[rt], dp = d + 1.0f;
          float* fe = a.Fe + 3 * (size_t)t.row[rt];
          fe[0] = w * rx[rt] / d / dp;
          fe[1] = w * ry[rt] / d / dp;
          fe[2] = w * rz[rt] / d / dp;
        }
      }
    } else {
      row_gather<4, RR>(pr.he, a.Hep, pr.t.row, 64, q);
    }
    STAMPB(40);
    STAMPB(47);
    if (!more) break;
    unit = unext;
  }
  if (dyn) wq_leave(wp, lane);
}

}  // namespace

template <int FLAGS>
static void launch_b2(const EdgeBArgs& a, hipStream_t s) {
  const int nunits = (a.E + ROWS - 1) / ROWS;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  hipLaunchKernelGGL(edge_b2_kernel<FLAGS>, dim3(grid), dim3(MDX_WG), EB_CONST_FLOATS * 4, s, a, nunits,
                     make_workq(a.wq, nunits, grid, mdx_num_cus()));
}

int launch_edge_b2(const EdgeBArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  switch (a.flags) {
    case EB_EDGE | EB_POS: launch_b2<EB_EDGE | EB_POS>(a, s); return MDX_OK;      // MolDiff denoiser
    case EB_EDGE: launch_b2<EB_EDGE>(a, s); return MDX_OK;                        // bond predictor (update_pos = False)
    case EB_EDGE | EB_DELTA: launch_b2<EB_EDGE | EB_DELTA>(a, s); return MDX_OK;  // EdgeBlock.forward
    case EB_POS: launch_b2<EB_POS>(a, s); return MDX_OK;                          // PosUpdate.forward
    default: return mdx_set_error(MDX_ERR_UNSUPPORTED, "edge kernel B: unsupported section flags");
  }
}

int launch_edge_b(const EdgeBArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  return (a.flags & EB_SPLIT) ? launch_edge_b2s(a, s) : launch_edge_b2(a, s);
}
