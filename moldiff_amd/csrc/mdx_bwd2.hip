// Row-owner guidance backward (gfx950), exact fp32 build: edge_bwd2_kernel / edge_tail_bwd2_kernel.  The kernels' text is
// mdx_bwd2_body.h (shared with the split float16 build, mdx_bwd2s.hip); this file binds it to v_mfma_f32_16x16x4_f32 (rgemm,
// mdx_row.h) and the fp32 stream packs, and holds the launchers.
#include "mdx_kernels.h"
#ifndef MDX_RING
#define MDX_RING 3  // weight-ring depth in steps (9.41 / 9.25 / 9.31 ms per guided step at depth 2 / 3 / 4 (4 spills))
#endif
#include "mdx_row.h"
#include "../../include/moldiff_hip.h"
#include <algorithm>

// phase trace, see mdx_edge2.hip (tools/trace_edge2.py w); compiled out of the library
#ifdef MDX_TRACE2
__device__ unsigned long long* mdx_trace_bwd_buf = nullptr;
extern "C" int mdx_debug_set_trace_bwd(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(mdx_trace_bwd_buf), &p, sizeof(p)); }
#define STAMPW(i)                                                                                             \
  do {                                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    if (lane == 0 && mdx_trace_bwd_buf)                                                                       \
      mdx_trace_bwd_buf[(size_t)unit * 48 + (i)] = ((i) >= 46) ? wall_clock64() : clock64();                  \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  } while (0)
#else
#define STAMPW(i) ((void)0)
#endif

namespace {
#define BW_GEMM(KG, FT) rgemm<KG, FT, RR>
#define BW_S s
#define BW_KERNEL edge_bwd2_kernel
#define BW_TAIL_KERNEL edge_tail_bwd2_kernel
#define BW_TAIL_WOUTT sWoutT
#define BW_TAIL_WSELFT sWselfT
#include "mdx_bwd2_body.h"
}  // namespace

void launch_edge_tail_bwd2(const EdgeTailBwdArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  const int nunits = (a.E + ROWS - 1) / ROWS;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  hipLaunchKernelGGL(edge_tail_bwd2_kernel, dim3(grid), dim3(MDX_WG), 0, s, a, nunits, make_workq(a.wq, nunits, grid, mdx_num_cus()));
}

void launch_edge_bwd2(const EdgeBwdArgs& a, hipStream_t s) {
  if (a.E <= 0) return;
  static bool attr = false;
  constexpr int lds = (4 * PARK_FLOATS + BW_CONST_FLOATS) * 4;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)edge_bwd2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)edge_bwd2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr = true;
  }
  const int nunits = a.nunits_r;  // graph-aligned units of the by-right order (mdx_graph_s units_r)
  if (nunits <= 0) return;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  const WorkQ wq = make_workq(a.wq, nunits, grid, mdx_num_cus());
  if (a.fuse_tail) hipLaunchKernelGGL(edge_bwd2_kernel<true>, dim3(grid), dim3(MDX_WG), lds, s, a, nunits, wq);
  else hipLaunchKernelGGL(edge_bwd2_kernel<false>, dim3(grid), dim3(MDX_WG), lds, s, a, nunits, wq);
}
