// Shared by the exact (mdx_edge2.hip) and the split-precision (mdx_edge2s.hip) row-owner edge kernel A: the LDS constant budget,
// the static work plan with its section-cut tail, the work-queue item map and the one-unit-ahead prologue loads.
#pragma once
#include "mdx_kernels.h"
#include "mdx_row.h"

namespace {

constexpr int EA_CONST_FLOATS = 96 + 2560 + 2 * 640;

// Work list of a persistent wave ("slot"): nf full units (a contiguous range), then its share of the last, partial round.
// When the last round is short and the kernel runs both sections, that round is cut by SECTION instead of by rows (a
// 16-row MFMA tile cannot be split further).  Unit costs measured with tools/trace_edge2.py (fractions of a unit): message path
// incl. edge_embs 0.72, left BondFFN 0.13, right BondFFN 0.15 (with its in-kernel segment sum), edge_embs recomputed by a slot
// that only runs BondFFNs 0.03.  Two cuts:
//   split 1: the first `rem` slots run the message path of one unit each (0.72), the other slots both BondFFNs of `m` <= 2 units
//            each (0.31 m);
//   split 2: the first `rem` slots run the message path AND the right BondFFN of one unit each (0.87), the other slots the left
//            BondFFN of `m` <= 5 units each (0.16 m)   -- covers rem up to 5/6 of the slots (split 1 with m = 3 would take 0.93).
struct EdgePlan {
  int nslots, nf, rem, split, m;
};
__host__ __device__ inline int plan_items(const EdgePlan& p, int slot) {
  if (!p.split) return p.nf + (slot < p.rem ? 1 : 0);
  if (slot < p.rem) return p.nf + 1;
  const int k = slot - p.rem, left = p.rem - k * p.m;
  return p.nf + (left < 0 ? 0 : left < p.m ? left : p.m);
}
// item it of a slot -> unit index; mode bits: 1 message path, 2 left BondFFN, 8 right BondFFN, 4 this item owns the He' store
__device__ __forceinline__ int plan_item(const EdgePlan& p, int slot, int it, int& mode) {
  if (it < p.nf) {
    mode = 15;
    return slot * p.nf + it;
  }
  const int base = p.nf * p.nslots;
  if (!p.split || slot < p.rem) {
    mode = p.split == 0 ? 15 : p.split == 1 ? 5 : 13;
    return base + slot;
  }
  mode = p.split == 1 ? 10 : 2;
  return base + (slot - p.rem) * p.m + (it - p.nf);
}
inline EdgePlan make_plan(int nunits, int nslots, bool can_split) {
  EdgePlan p{nslots, nunits / nslots, nunits % nslots, 0, 0};
  if (can_split && p.rem > 0 && p.rem < nslots) {
    const int m = (p.rem + (nslots - p.rem) - 1) / (nslots - p.rem);
    if (m <= 2) {
      p.split = 1;
      p.m = m;
    } else if (m <= 5) {
      p.split = 2;
      p.m = m;
    }
  }
  return p;
}

// Work queue of the full kernel (mdx_row.h, WorkQ): a pair of workgroups hands out its units in order; the last few of them
// (tail8 eighths of a unit per wave) are cut by section so that the waves, which arrive at the end of the list up to one unit
// apart, finish close together: first the message-path halves of those units (0.72 of a unit), then their BondFFN halves
// (0.31).  Largest pieces first: the spread at the end is bounded by the smallest piece.
struct WorkQA {
  WorkQ q;
  int tail8;
};
// item i of a pair's list -> unit (relative to the XCD's first) and mode
__device__ __forceinline__ int wq_item(int i, int cnt, int nt, int& mode) {
  if (i < cnt) {
    mode = i < cnt - nt ? 15 : 5;
    return i;
  }
  mode = 10;
  return i - nt;
}

// rows of the He tile + edge length of one unit: loaded one unit ahead by the persistent loop
struct Prolog {
  RowTile t;
  f32x4 x[4][RR];
  float d[RR];
};

__device__ __forceinline__ void prolog_rows(Prolog& p, const EdgeAArgs& a, int q) {
  row_gather<4, RR>(p.x, a.He_in, p.t.row, 64, q);
  if (a.flags & EA_EMB) {
#pragma unroll
    for (int rt = 0; rt < RR; ++rt) {
      if (a.dist_in) {
        p.d[rt] = a.dist_in[p.t.row[rt]];
      } else {
        const float dx = a.pos[3 * p.t.li[rt] + 0] - a.pos[3 * p.t.ri[rt] + 0];
        const float dy = a.pos[3 * p.t.li[rt] + 1] - a.pos[3 * p.t.ri[rt] + 1];
        const float dz = a.pos[3 * p.t.li[rt] + 2] - a.pos[3 * p.t.ri[rt] + 2];
        p.d[rt] = sqrtf(dx * dx + dy * dy + dz * dz);
      }
    }
  }
}

}  // namespace
