// Row-owner Linear for the training path (gfx950): Y = X W^T (+ bias) (+ addend) on MANY rows (the E ~ 155 k edge rows of a
// training batch), replacing torch.nn.functional.linear / its grad_input as autograd runs them for every nn.Linear of
// models/graph.py and models/common.py:181-201.  Same decomposition as the fused sampling kernels (mdx_row.h): one wave owns 16
// rows and ALL output features, the input rows sit in registers as the MFMA B operand for the whole contraction, the weight is
// streamed L2 -> registers in consumption order through the buffer-load ring, no LDS tile and no barrier; two waves per SIMD cover
// each other's row loads and stores.  The LDS-staged tile kernel (sgemm_nt_kernel, mdx_train.hip) reaches 0.53 of the fp32 MFMA
// peak on the 256 x 256 layers; this one is built for those and the other wide layers and leaves odd shapes to it.
//
// The weight changes every optimizer step, so it is brought into stream-pack order by a small kernel in front of each GEMM
// (<= 256 KiB); `transW` reads it transposed, which is what the grad_input GEMM needs (no separate transpose pass).
#include "mdx_kernels.h"
#ifndef MDX_RING
#define MDX_RING 4
#endif
#include "mdx_row.h"
#include "../../include/moldiff_hip.h"
#include <algorithm>
int mdx_set_error(int code, const char* msg);

namespace {

// pack[((((ftp*KG + g)*2 + j)*64 + lane)*4 + s)] = Wsel(f = 16 (2 ftp + j) + (lane & 15), k = 16 g + 4 (lane >> 4) + s), zero outside
// (N,K); Wsel(f,k) = W[f][k] (transW = 0: W is (N,K)) or W[k][f] (transW = 1: W is (K,N)); MDX_RING zero steps at the end.
__global__ void pack_stream_kernel(const float* __restrict__ W, int ldw, int transW, int N, int K, int KG, int FTP,
                                   float* __restrict__ out, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int s = i & 3, lane = (i >> 2) & 63, j = (i >> 8) & 1, step = i >> 9;
  float v = 0.f;
  if (step < FTP * KG) {
    const int ftp = step / KG, g = step - ftp * KG;
    const int f = 16 * (2 * ftp + j) + (lane & 15), k = 16 * g + 4 * (lane >> 4) + s;
    if (f < N && k < K) v = transW ? W[(size_t)k * ldw + f] : W[(size_t)f * ldw + k];
  }
  out[i] = v;
}

struct LinArgs {
  const float *X, *Wp, *bias, *addend;
  float* Y;
  int ldx, ldd, ldy, M, N;
};

template <int KG, int FT>
__global__ __launch_bounds__(MDX_WG, MDX_WPS) void linear_rows_kernel(const LinArgs a, const int nunits) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q0 = lane >> 4;
  const unsigned lane_off = 16u * lane;
  auto W = [&](const float* p) { return make_ws(p, lane_off); };
  const int nslots = gridDim.x * 4;
  const int slot = blockIdx.x * 4 + wave;
  const int per = (nunits + nslots - 1) / nslots;
  const int ubeg = slot * per, uend = min(nunits, ubeg + per);
  if (ubeg >= uend) return;
  WRing ring;
  ring_prime(ring, W(a.Wp));
#pragma unroll 1
  for (int unit = ubeg; unit < uend; ++unit) {
    int q = q0;
    asm volatile("" : "+v"(q));
    int row[1];
    bool valid[1];
    row[0] = unit * 16 + c;
    valid[0] = row[0] < a.M;
    if (!valid[0]) row[0] = a.M - 1;
    f32x4 x[KG][1], y[FT][1];
    // (requesting the next unit's rows one GEMM ahead was measured: no gain -- what a 16-row unit of a single layer waits for is
    // its own store burst: memory operations retire in order, so the weight ring of the next unit queues behind 16 stores)
    row_gather<KG, 1>(x, a.X, row, a.ldx, q);
    if (a.bias) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        const int col = 16 * ft + 4 * q;
        f32x4 b = splat4(0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (col + r < a.N) b[r] = a.bias[col + r];  // (parameters sit at any 4-byte offset of the flat buffer)
        y[ft][0] = b;
      }
    } else {
      row_zero<FT, 1>(y);
    }
    rgemm<KG, FT, 1>(y, x, W(a.Wp), ring, W(a.Wp));
    if (valid[0]) {
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) {
        const int col = 16 * ft + 4 * q;
        if (col < a.N) {  // N is a multiple of 4
          f32x4 v = y[ft][0];
          if (a.addend) v = v + ldg4(a.addend + (size_t)row[0] * a.ldd + col);
          stg4(a.Y + (size_t)row[0] * a.ldy + col, v);
        }
      }
    }
  }
}

template <int KG, int FT>
void launch_lin(const LinArgs& a, hipStream_t s) {
  const int nunits = (a.M + 15) / 16;
  const int grid = std::min((nunits + 3) / 4, mdx_num_cus() * MDX_WPS);
  hipLaunchKernelGGL((linear_rows_kernel<KG, FT>), dim3(grid), dim3(MDX_WG), 0, s, a, nunits);
}

inline bool shape_ok(int64_t N, int64_t K) {
  const int kg = (int)(K / 16), ft = (int)((N + 31) / 32 * 2);
  if (K % 16 || N % 4 || N < 1) return false;
  return (kg == 1 || kg == 2 || kg == 4 || kg == 5 || kg == 8 || kg == 16) && (ft == 2 || ft == 4 || ft == 8 || ft == 16);
}

}  // namespace

extern "C" int mdx_op_linear_rows_supported(int64_t N, int64_t K) { return shape_ok(N, K) ? 1 : 0; }
extern "C" size_t mdx_op_linear_rows_ws(int64_t N, int64_t K) {
  const size_t ftp = (size_t)(N + 31) / 32, kg = (size_t)(K + 15) / 16;
  return (ftp * kg + MDX_RING + 1) * 512 * sizeof(float);
}

extern "C" int mdx_op_linear_rows(const float* X, int64_t ldx, const float* Wt, int64_t ldw, int32_t transW, const float* bias,
                                  const float* addend, int64_t ldd, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                                  float* pack_ws, void* stream) {
  if (M <= 0) return MDX_OK;
  if (!X || !Wt || !Y || !pack_ws) return mdx_set_error(MDX_ERR_ARG, "linear_rows: null operand");
  if (!shape_ok(N, K)) return mdx_set_error(MDX_ERR_UNSUPPORTED, "linear_rows: shape not built (use sgemm_nt)");
  if ((ldx & 3) || (ldy & 3) || (addend && (ldd & 3)) || ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) |
                                                          reinterpret_cast<uintptr_t>(addend) | reinterpret_cast<uintptr_t>(pack_ws)) & 15))
    return mdx_set_error(MDX_ERR_ARG, "linear_rows: rows must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int KG = (int)(K / 16), FTP = (int)((N + 31) / 32);
  const int total = (FTP * KG + MDX_RING + 1) * 512;
  hipLaunchKernelGGL(pack_stream_kernel, dim3((total + 255) / 256), dim3(256), 0, s, Wt, (int)ldw, (int)transW, (int)N, (int)K, KG, FTP,
                     pack_ws, total);
  LinArgs a{X, pack_ws, bias, addend, Y, (int)ldx, (int)ldd, (int)ldy, (int)M, (int)N};
#define MDX_LIN_FT(KGv)                                        \
  switch (2 * FTP) {                                           \
    case 2: launch_lin<KGv, 2>(a, s); break;                   \
    case 4: launch_lin<KGv, 4>(a, s); break;                   \
    case 8: launch_lin<KGv, 8>(a, s); break;                   \
    default: launch_lin<KGv, 16>(a, s); break;                 \
  }
  switch (KG) {
    case 1: MDX_LIN_FT(1) break;
    case 2: MDX_LIN_FT(2) break;
    case 4: MDX_LIN_FT(4) break;
    case 5: MDX_LIN_FT(5) break;
    case 8: MDX_LIN_FT(8) break;
    default: MDX_LIN_FT(16) break;
  }
#undef MDX_LIN_FT
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? MDX_OK : mdx_set_error(MDX_ERR_HIP, hipGetErrorString(e));
}
