// Micro-benchmark of the fused-kernel building block gemm_tile<4,ET,256> (csrc/mdx_tile.h):
// how close to the fp32-MFMA peak does the inner loop alone get, with the same LDS tile / packed-weight stream /
// workgroup shape as edge kernel A, and 1..3 workgroups per CU?   hipcc --offload-arch=gfx950 -O3 -I moldiff_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "mdx_tile.h"

template <int ET, int WPS, int K, int FTW>
__global__ __launch_bounds__(256, WPS) void k(const float* __restrict__ W, float* out, int reps, int nmat) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int LD = mdx_ld(256);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * ET * LD; i += 256) smem[i] = (float)((i * 7 + blockIdx.x) % 13) * 0.01f;
  __syncthreads();
  f32x4 acc[FTW][ET];
  acc_zero<FTW, ET>(acc);
  for (int r = 0; r < reps; ++r) {
    const float* Wp = W + (size_t)((r + blockIdx.x) % nmat) * 65536;
    gemm_tile<FTW, ET, K>(acc, Wp, 4 * FTW, FTW * wave, smem, LD, lane);
    __syncthreads();
  }
  f32x4 s = splat4(0.f);
  for (int ft = 0; ft < FTW; ++ft)
    for (int et = 0; et < ET; ++et) s = s + acc[ft][et];
  out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

template <int ET, int WPS, int K = 256, int FTW = 4>
void run(const float* W, float* out, int nmat) {
  const int reps = 64, grid = 256 * WPS * 4;
  const size_t lds = (size_t)16 * ET * mdx_ld(256) * 4 + (WPS == 2 ? 20000 : WPS == 1 ? 60000 : 0);  // pad so that exactly WPS WGs fit
  hipFuncSetAttribute((const void*)k<ET, WPS, K, FTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<ET, WPS, K, FTW>), dim3(grid), dim3(256), lds, 0, W, out, reps, nmat);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<ET, WPS, K, FTW>), dim3(grid), dim3(256), lds, 0, W, out, reps, nmat);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flop = (double)grid * reps * 2.0 * K * (64 * FTW) * 16 * ET;
  printf("ET=%d WG/CU=%d K=%d NOUT=%d lds=%zu : %.3f ms  %.1f TFLOP/s\n", ET, WPS, K, 64 * FTW, lds, ms, flop / ms / 1e9);
}

int main() {
  const int nmat = 20;
  std::vector<float> h((size_t)nmat * 65536);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
  float *W, *out;
  hipMalloc(&W, h.size() * 4); hipMalloc(&out, 256 * 12 * 256 * 4 * 4);
  hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<3, 1>(W, out, nmat); run<3, 2>(W, out, nmat); run<3, 2, 128, 4>(W, out, nmat); run<3, 2, 64, 4>(W, out, nmat); run<3, 2, 128, 2>(W, out, nmat);
  run<3, 2, 64, 2>(W, out, nmat); run<3, 2, 128, 1>(W, out, nmat); run<3, 2, 64, 1>(W, out, nmat); run<3, 2, 32, 1>(W, out, nmat);
  return 0;
}
