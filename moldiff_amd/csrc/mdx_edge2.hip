// Row-owner fused per-edge kernels of one NodeEdgeNet block (gfx950) -- round 2 replacement for mdx_edge.hip's tile kernels.
//
// Same math, same argument blocks (EdgeAArgs / EdgeBArgs), different work decomposition (see mdx_row.h):
//   edge kernel A  (reference models/graph.py:352-357 edge_embs, :42-47 NodeBlock message path,
//                   :133-141/:278,:282 the two EdgeBlock BondFFNs)
//   edge kernel B  (models/graph.py:286-294 EdgeBlock tail, :384-393 PosUpdate)
// One WAVE owns 16*R consecutive edges of the (left,right)-sorted edge list and computes every layer for them; its
// activations stay in registers from the He tile load to the M / F / He'' / Fe stores.  No LDS tile, no barrier.
// LDS is only a wave-private parking area for sigmoid(gate) while the message chain runs (32 KiB per wave).
#include "mdx_kernels.h"
#include "mdx_row.h"

namespace {

constexpr int RR = 2;                 // row tiles per wave: 32 edges
constexpr int ROWS = 16 * RR;
constexpr int PARK_FLOATS = ROWS * MDX_ND;  // per wave

struct RowTile {
  int row[RR], li[RR], ri[RR];
  float tt[RR];
  bool valid[RR];
};

__device__ __forceinline__ RowTile load_tile(const int* __restrict__ l, const int* __restrict__ r, const float* __restrict__ te,
                                             int e0, int E, int c) {
  RowTile t;
#pragma unroll
  for (int rt = 0; rt < RR; ++rt) {
    const int e = e0 + 16 * rt + c;
    t.valid[rt] = e < E;
    t.row[rt] = t.valid[rt] ? e : E - 1;  // clamped: loads stay in bounds, stores are predicated on valid
    t.li[rt] = l[t.row[rt]];
    t.ri[rt] = r[t.row[rt]];
    t.tt[rt] = te[t.row[rt]];
  }
  return t;
}

template <int FT>
__device__ __forceinline__ void mul_inplace(f32x4 (&y)[FT][RR], const f32x4 (&v)[FT][RR]) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) y[ft][rt] = y[ft][rt] * v[ft][rt];
}

__global__ __launch_bounds__(MDX_WG, 1) void edge_a2_kernel(const EdgeAArgs a, const int nunits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int unit = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (unit >= nunits) return;  // no barrier anywhere below: a wave may leave on its own
  const int E = a.E;
  const RowTile t = load_tile(a.l, a.r, a.te, unit * ROWS, E, c);
  const bool do_node = a.flags & EA_NODE, do_ffn = a.flags & EA_FFN;
  f32x4* park = reinterpret_cast<f32x4*>(smem + (size_t)wave * PARK_FLOATS) + lane;
  auto W = [&](const float* p) { return reinterpret_cast<const f32x4*>(p) + lane; };
  WRing ring;

  // ---- He' = edge_embs([He | smear(d)]) -------------------------------------------------------
  f32x4 hep[4][RR];
  if (a.flags & EA_EMB) {
    ring_prime(ring, W(a.w.s.Wemb));
    f32x4 x[5][RR];
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) {
      const float* p = a.He_in + (size_t)t.row[rt] * 64 + 4 * q;
#pragma unroll
      for (int g = 0; g < 4; ++g) x[g][rt] = ldg4(p + 16 * g);
    }
    const f32x4 off = ldg4(a.soff + 4 * q), coef = ldg4(a.scoef + 4 * q);
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) {
      float d;
      if (a.dist_in) {
        d = a.dist_in[t.row[rt]];
      } else {
        const float dx = a.pos[3 * t.li[rt] + 0] - a.pos[3 * t.ri[rt] + 0];
        const float dy = a.pos[3 * t.li[rt] + 1] - a.pos[3 * t.ri[rt] + 1];
        const float dz = a.pos[3 * t.li[rt] + 2] - a.pos[3 * t.ri[rt] + 2];
        d = sqrtf(dx * dx + dy * dy + dz * dz);
      }
      const float u0 = fminf(fmaxf(d, 0.f), a.cutoff);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float u = u0 - off[s];
        x[4][rt][s] = expf(coef[s] * (u * u));
      }
    }
    row_bias<4, RR>(hep, a.w.bemb, q);
    rgemm<5, 4, RR>(hep, x, W(a.w.s.Wemb), ring);
    row_store<4, RR>(hep, a.He_out, t.row, t.valid, 64, q);
  } else {
    row_gather<4, RR>(hep, a.He_in, t.row, 64, q);
  }

  // ---- NodeBlock message path: M = msg_net(edge_net(He') * h[r]) * sigmoid(gate([He' | x[r] | t])) --------
  if (do_node) {
    ring_prime(ring, W(a.w.s.Wg1e));
    f32x4 y[16][RR], z[16][RR];
    {  // gate layer 1: accumulator starts at b + gx[r] + t*wt (the hoisted node part and the time column)
      row_gather<16, RR>(y, a.NT + MDX_NT_GX, t.ri, MDX_NTW, q);
      float tg[RR];
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) tg[rt] = a.tn_r ? a.tn_r[t.row[rt]] : t.tt[rt];  // the NodeBlock gate sees node_time[col]
#pragma unroll
      for (int ft = 0; ft < 16; ++ft) {
        const f32x4 b = ldg4(a.w.bg1 + 16 * ft + 4 * q), wt = ldg4(a.w.wtg1 + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) y[ft][rt] = (b + y[ft][rt]) + splat4(tg[rt]) * wt;
      }
    }
    rgemm<4, 16, RR>(y, hep, W(a.w.s.Wg1e), ring);
    ring_prime(ring, W(a.w.s.Wg2));
    row_layernorm<16, RR>(y, a.w.gg, a.w.gb, q);
    row_bias<16, RR>(z, a.w.bg2, q);
    rgemm<16, 16, RR>(z, y, W(a.w.s.Wg2), ring);
    ring_prime(ring, W(a.w.s.W1));
#pragma unroll
    for (int ft = 0; ft < 16; ++ft)
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        const f32x4 sg = sigmoid4(z[ft][rt]);
        if (a.tSG && t.valid[rt]) stg4(a.tSG + (size_t)t.row[rt] * MDX_ND + 16 * ft + 4 * q, sg);
        park[(ft * RR + rt) * 64] = sg;
      }
    // edge_net
    row_bias<16, RR>(y, a.w.en.b1, q);
    rgemm<4, 16, RR>(y, hep, W(a.w.s.W1), ring);
    ring_prime(ring, W(a.w.s.W2));
    row_layernorm<16, RR>(y, a.w.en.g, a.w.en.be, q);
    row_bias<16, RR>(z, a.w.en.b2, q);
    rgemm<16, 16, RR>(z, y, W(a.w.s.W2), ring);
    ring_prime(ring, W(a.w.s.Wm));
    if (a.tHE) row_store<16, RR>(z, a.tHE, t.row, t.valid, MDX_ND, q);
    row_gather<16, RR>(y, a.H, t.ri, MDX_ND, q);
    mul_inplace<16>(z, y);
    // msg_net, gated
    row_bias<16, RR>(y, a.w.bm, q);
    rgemm<16, 16, RR>(y, z, W(a.w.s.Wm), ring);
#pragma unroll
    for (int ft = 0; ft < 16; ++ft)
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) y[ft][rt] = y[ft][rt] * park[(ft * RR + rt) * 64];
    row_store<16, RR>(y, a.M, t.row, t.valid, MDX_ND, q);
  }

  // ---- EdgeBlock BondFFNs: F_s = inter_s((W_bl He') * nl_s[idx_s]) * sigmoid(gate_s([He' | x[idx_s] | t])) ----
  if (do_ffn) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const FfnW& w = a.w.ffn[s];
      const FfnS& ws = a.w.s.ffn[s];
      int idx[RR];
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) idx[rt] = s ? t.ri[rt] : t.li[rt];
      ring_prime(ring, W(ws.Wbl));
      f32x4 bl[8][RR], nl[8][RR], g1[2][RR];
      row_gather<8, RR>(nl, a.NT + (s ? MDX_NT_NLR : MDX_NT_NLL), idx, MDX_NTW, q);
      row_gather<2, RR>(g1, a.NT + (s ? MDX_NT_GXR : MDX_NT_GXL), idx, MDX_NTW, q);
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        const f32x4 b = ldg4(w.bg1 + 16 * ft + 4 * q), wt = ldg4(w.wtg1 + 16 * ft + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) g1[ft][rt] = (b + g1[ft][rt]) + splat4(t.tt[rt]) * wt;
      }
      row_bias<8, RR>(bl, nullptr, q);
      rgemm<4, 8, RR>(bl, hep, W(ws.Wbl), ring);
      ring_prime(ring, W(ws.Wg1e));
      mul_inplace<8>(bl, nl);
      rgemm<4, 2, RR>(g1, hep, W(ws.Wg1e), ring);
      ring_prime(ring, W(ws.W1));
      row_layernorm<2, RR>(g1, w.gg, w.gb, q);
      f32x4 h[8][RR];
      row_bias<8, RR>(h, w.inter.b1, q);
      rgemm<8, 8, RR>(h, bl, W(ws.W1), ring);
      ring_prime(ring, W(ws.W2));
      row_layernorm<8, RR>(h, w.inter.g, w.inter.be, q);
      f32x4 o[4][RR], g2[4][RR];
      row_bias<4, RR>(o, w.inter.b2, q);
      rgemm<8, 4, RR>(o, h, W(ws.W2), ring);
      ring_prime(ring, W(ws.Wg2));
      row_bias<4, RR>(g2, w.bg2, q);
      rgemm<2, 4, RR>(g2, g1, W(ws.Wg2), ring);
#pragma unroll
      for (int ft = 0; ft < 4; ++ft)
#pragma unroll
        for (int rt = 0; rt < RR; ++rt) o[ft][rt] = o[ft][rt] * sigmoid4(g2[ft][rt]);
      row_store<4, RR>(o, a.F[s], t.row, t.valid, 64, q);
    }
  }
}

__global__ __launch_bounds__(MDX_WG, 1) void edge_b2_kernel(const EdgeBArgs a, const int nunits) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const int unit = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
  if (unit >= nunits) return;
  const int E = a.E;
  const RowTile t = load_tile(a.l, a.r, a.te, unit * ROWS, E, c);
  const bool do_edge = a.flags & EB_EDGE, do_pos = a.flags & EB_POS;
  auto W = [&](const float* p) { return reinterpret_cast<const f32x4*>(p) + lane; };
  WRing ring;

  f32x4 he[4][RR];  // He' on entry, He'' after the EdgeBlock tail
  row_gather<4, RR>(he, a.Hep, t.row, 64, q);

  // ---- EdgeBlock tail: He'' = He' + out_transform(relu(LN(SL[l] + SR[r] + nfl[l] + nfr[r] + self_ffn(He')))) ----
  if (do_edge) {
    ring_prime(ring, W(a.w.s.Wself));
    f32x4 u[4][RR], v[4][RR];
    row_gather<4, RR>(u, a.SL, t.li, 64, q);
    row_gather<4, RR>(v, a.SR, t.ri, 64, q);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) u[ft][rt] = u[ft][rt] + v[ft][rt];
    row_gather<4, RR>(v, a.NT + MDX_NT_NFL, t.li, MDX_NTW, q);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) u[ft][rt] = u[ft][rt] + v[ft][rt];
    row_gather<4, RR>(v, a.NT + MDX_NT_NFR, t.ri, MDX_NTW, q);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
      const f32x4 bs = ldg4(a.w.bself + 16 * ft + 4 * q);
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) u[ft][rt] = (u[ft][rt] + v[ft][rt]) + bs;
    }
    rgemm<4, 4, RR>(u, he, W(a.w.s.Wself), ring);
    ring_prime(ring, W(a.w.s.Wout));
    row_layernorm<4, RR>(u, a.w.lng, a.w.lnb, q);
    row_bias<4, RR>(v, a.w.bout, q);
    rgemm<4, 4, RR>(v, u, W(a.w.s.Wout), ring);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft)
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) {
        if (!(a.flags & EB_DELTA)) v[ft][rt] = v[ft][rt] + he[ft][rt];
        he[ft][rt] = v[ft][rt];
      }
    row_store<4, RR>(he, a.He_out, t.row, t.valid, 64, q);
  }

  // ---- PosUpdate: w = inter((W_bl He'') * (W_nl a)) * sigmoid(gate([He'' | a | t])), a = Lf[l] * Rf[r]; Fe = w rel / d / (d+1) ----
  if (do_pos) {
    ring_prime(ring, W(a.w.s.Wbl));
    f32x4 aa[4][RR], bb[4][RR];
    row_gather<4, RR>(aa, a.Lf, t.li, 64, q);
    row_gather<4, RR>(bb, a.Rf, t.ri, 64, q);
    mul_inplace<4>(aa, bb);
    float rx[RR], ry[RR], rz[RR], dd[RR];
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
    f32x4 x[16][RR], h[16][RR], g1[2][RR];
    row_bias<16, RR>(x, nullptr, q);
    rgemm<4, 16, RR>(x, he, W(a.w.s.Wbl), ring);
    ring_prime(ring, W(a.w.s.Wnl));
    row_bias<16, RR>(h, nullptr, q);
    rgemm<4, 16, RR>(h, aa, W(a.w.s.Wnl), ring);
    ring_prime(ring, W(a.w.s.Wg1h));
    mul_inplace<16>(x, h);
    // gate: ((b + t wt) + W_h He'') + W_a a, LN(32), ReLU, 32 -> 1
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      const f32x4 b = ldg4(a.w.bg1 + 16 * ft + 4 * q), wt = ldg4(a.w.wtg1 + 16 * ft + 4 * q);
#pragma unroll
      for (int rt = 0; rt < RR; ++rt) g1[ft][rt] = b + splat4(t.tt[rt]) * wt;
    }
    rgemm<4, 2, RR>(g1, he, W(a.w.s.Wg1h), ring);
    ring_prime(ring, W(a.w.s.Wg1a));
    rgemm<4, 2, RR>(g1, aa, W(a.w.s.Wg1a), ring);
    ring_prime(ring, W(a.w.s.Wi1));
    row_layernorm<2, RR>(g1, a.w.gg, a.w.gb, q);
    float gate[RR], wd[RR];
    row_dot<2, RR>(g1, a.w.wg2, q, gate);
    row_bias<16, RR>(h, a.w.bi1, q);
    rgemm<16, 16, RR>(h, x, W(a.w.s.Wi1), ring);
    row_layernorm<16, RR>(h, a.w.ig, a.w.ib, q);
    row_dot<16, RR>(h, a.w.wi2, q, wd);
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
  }
}

}  // namespace

bool mdx_use_rowowner() {
  static const bool v = [] {
    const char* e = getenv("MDX_TILE_KERNELS");
    return !(e && e[0] == '1');
  }();
  return v;
}

void launch_edge_a2(const EdgeAArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  static bool attr = false;
  constexpr int lds = 4 * PARK_FLOATS * 4;
  if (!attr) {
    hipFuncSetAttribute((const void*)edge_a2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int nunits = (a.E + ROWS - 1) / ROWS;
  hipLaunchKernelGGL(edge_a2_kernel, dim3((nunits + 3) / 4), dim3(MDX_WG), lds, s, a, nunits);
}

void launch_edge_b2(const EdgeBArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const int nunits = (a.E + ROWS - 1) / ROWS;
  hipLaunchKernelGGL(edge_b2_kernel, dim3((nunits + 3) / 4), dim3(MDX_WG), 0, s, a, nunits);
}
