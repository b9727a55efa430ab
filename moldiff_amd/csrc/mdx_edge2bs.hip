// Split-precision build of the row-owner edge kernel B (gfx950): EdgeBlock tail + PosUpdate
// (reference models/graph.py:286-294 and :384-393) -- round 4, opt-in; see mdx_edge2s.hip / mdx_split.h.
// Same argument block, work decomposition and outputs as edge_b2_kernel (mdx_edge2b.hip); only the matrix products differ.
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
#define MDX_RING 4  // half-steps of 2 KiB in flight per wave (seamless: 133.0 us per launch; 8: 135.5; the primed ring of 8: 135.6)
#endif
#ifndef MDX_SPLIT_SEAMLESS
#define MDX_SPLIT_SEAMLESS 1  // the weight ring runs through GEMM boundaries (mdx_split.h); at 4 half-steps no GEMM of this kernel idles
#endif
#include "mdx_row.h"
#include "mdx_split.h"
#include "../../include/moldiff_hip.h"
#include <algorithm>
int mdx_set_error(int code, const char* msg);

namespace {

constexpr int EB_CONST_FLOATS = 4 * 64 + 5 * 32 + 4 * 256;

struct PrologB {
  RowTile t;
  f32x4 he[4][RR];
};

template <int FLAGS>
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void edge_b2s_kernel(const EdgeBArgs a, const int nunits, const WorkQ wq) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  const int E = a.E;
  constexpr bool do_edge = FLAGS & EB_EDGE, do_pos = FLAGS & EB_POS;
  const unsigned lane_off = 16u * lane;
  auto W = [&](const float* p) { return make_ws(p, lane_off); };
  const EdgeBS& S = a.w.ss;

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

  const float* wfirst = do_edge ? S.Wself : S.Wbl;
  WRing ring;
  ring_prime(ring, W(wfirst));
  PrologB pr;
  pr.t = load_tile(a.l, a.r, a.te, ubeg * ROWS, E, c);
  row_gather<4, RR>(pr.he, a.Hep, pr.t.row, 64, q0);

#pragma unroll 1
  for (int unit = ubeg;;) {
    int q = q0;
    asm volatile("" : "+v"(q));
    const RowTile t = pr.t;
    const int ureq = dyn ? wq_request(wp.line, lane) : 0;
    f32x4 he[4][RR];  // He' on entry, He'' after the EdgeBlock tail
#pragma unroll
    for (int rt = 0; rt < RR; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) he[g][rt] = pr.he[g][rt];

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
      {
        XS<2> hx;
        to_xs<4>(hx, he);
        rgemm_s<2, 4>(u, hx, W(S.Wself), ring, W(S.Wout));
      }
      row_layernorm<4, RR>(u, c_lng, c_lnb, q);
      {
        XS<2> us;
        to_xs<4>(us, u);
        row_bias<4, RR>(v, c_bout, q);
        rgemm_s<2, 4>(v, us, W(S.Wout), ring, W(do_pos ? S.Wbl : wfirst));
      }
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
    if (!more) unext = unit;
    pr.t = load_tile(a.l, a.r, a.te, unext * ROWS, E, c);

    // ---- PosUpdate: w = inter((W_bl He'') * (W_nl a)) * sigmoid(gate([He'' | a | t])), a = Lf[l] * Rf[r]; Fe = w rel / d / (d+1) ----
    if (do_pos) {
      mul_inplace<4>(aa, bb);
      XS<2> hx, ax;
      to_xs<4>(hx, he);
      to_xs<4>(ax, aa);
      XS<8> xs;  // (W_bl He'') * (W_nl a) as the split operand of the inter layer, formed one pair of feature tiles at a time
      static_for<0, 8>([&](auto fc) {
        constexpr int ftp = decltype(fc)::value;
        constexpr int PF = split_stream_pair_floats(64);
        f32x4 xb[2][RR], xn[2][RR];
        row_zero<2, RR>(xb);
        rgemm_s<2, 2>(xb, hx, W(S.Wbl + ftp * PF), ring, W(S.Wnl + ftp * PF));
        row_zero<2, RR>(xn);
        rgemm_s<2, 2>(xn, ax, W(S.Wnl + ftp * PF), ring, W(ftp < 7 ? S.Wbl + (ftp + 1) * PF : S.Wg1h));
#pragma unroll
        for (int rt = 0; rt < RR; ++rt)
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) {
            const float v = xb[tt / 4][rt][tt % 4] * xn[tt / 4][rt][tt % 4];
            const _Float16 hh = (_Float16)v;
            xs.hi[ftp][rt][tt] = hh;
            xs.lo[ftp][rt][tt] = (_Float16)((v - (float)hh) * MDX_LO_UP);
          }
      });
      // gate: ((b + t wt) + W_h He'') + W_a a, LN(32), ReLU, 32 -> 1
      f32x4 h[16][RR], g1[2][RR];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const f32x4 b = lds4(c_bg1 + 16 * ft + 4 * q), wt = lds4(c_wtg1 + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) g1[ft][rt] = b + splat4(t.tt[rt]) * wt;
      }
      rgemm_s<2, 2>(g1, hx, W(S.Wg1h), ring, W(S.Wg1a));
      rgemm_s<2, 2>(g1, ax, W(S.Wg1a), ring, W(S.Wi1));
      row_layernorm<2, RR>(g1, c_gg, c_gb, q);
      float gate[RR], wd[RR];
      row_dot<2, RR>(g1, c_wg2, q, gate);
      row_bias<16, RR>(h, c_bi1, q);
      rgemm_s<8, 16>(h, xs, W(S.Wi1), ring, W(wfirst));
      row_gather<4, RR>(pr.he, a.Hep, pr.t.row, 64, q);
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
    if (!more) break;
    unit = unext;
  }
  if (dyn) wq_leave(wp, lane);
}

}  // namespace

template <int FLAGS>
static void launch_b2s(const EdgeBArgs& a, hipStream_t s) {
  const int nunits = (a.E + ROWS - 1) / ROWS;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  hipLaunchKernelGGL(edge_b2s_kernel<FLAGS>, dim3(grid), dim3(MDX_WG), EB_CONST_FLOATS * 4, s, a, nunits,
                     make_workq(a.wq, nunits, grid, mdx_num_cus()));
}

int launch_edge_b2s(const EdgeBArgs& a, hipStream_t s) {
  if (a.E <= 0) return MDX_OK;
  switch (a.flags & ~EB_SPLIT) {
    case EB_EDGE | EB_POS: launch_b2s<EB_EDGE | EB_POS>(a, s); return MDX_OK;
    case EB_EDGE: launch_b2s<EB_EDGE>(a, s); return MDX_OK;
    default: return mdx_set_error(MDX_ERR_UNSUPPORTED, "split-precision edge kernel B: unsupported section flags");
  }
}
