// Split-precision build of the row-owner guidance backward (gfx950) -- round 4, opt-in; the exact build is mdx_bwd2.hip and the
// kernels' text mdx_bwd2_body.h.  Same argument blocks, work decomposition, tape reads, in-kernel sums and outputs; only the matrix
// products differ: every GEMM operand (gradients, recomputed activations, He') is split into float16 hi / lo halves on the way
// into v_mfma_f32_16x16x32_f16 (mdx_split.h: rgemm_x = to_xs + rgemm_s), against split packs of the same (transposed) weights.
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
#define BW_GEMM(KG, FT) rgemm_x<KG, FT>
#define BW_S ss
#define BW_KERNEL edge_bwd2s_kernel
#define BW_TAIL_KERNEL edge_tail_bwd2s_kernel
#define BW_TAIL_WOUTT ssWoutT
#define BW_TAIL_WSELFT ssWselfT
#include "mdx_bwd2_body.h"
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
    (void)hipFuncSetAttribute((const void*)edge_bwd2s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int nunits = a.nunits_r;  // graph-aligned units of the by-right order (mdx_graph_s units_r)
  if (nunits <= 0) return 0;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  const WorkQ wq = make_workq(a.wq, nunits, grid, mdx_num_cus());
  if (a.fuse_tail) hipLaunchKernelGGL(edge_bwd2s_kernel<true>, dim3(grid), dim3(MDX_WG), lds, s, a, nunits, wq);
  else hipLaunchKernelGGL(edge_bwd2s_kernel<false>, dim3(grid), dim3(MDX_WG), lds, s, a, nunits, wq);
  return 0;
}
