// How does v_mfma_f32_16x16x32_f16 round when it adds its 32 products to the fp32 accumulator?  (round 4: the split-precision path's
// accuracy hinges on it.)  C = 1536 (ulp 2^-13); the products of one output sum to f * ulp for f in {+-0.25, +-0.5, +-0.75, +-1.5}.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_rounding mfma_rounding.hip
// Round-to-nearest-even gives 0, 0(tie->even), 1, 2(tie: 1.5 -> 2) ulps;  truncation toward zero gives 0, 0, 0, 1 for f > 0 and -1, -1, -1, -2 for f < 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__global__ void k(const float* fr, float* out, int n, int spread) {
  const int lane = threadIdx.x;
  for (int i = 0; i < n; ++i) {
    // A[i][k] = a (all rows), B[k][j] = b: every output = C + 32 a b.  spread = 1: the mass sits in ONE product (the others 0)
    const float f = fr[i];
    h8 a, b;
    for (int t = 0; t < 8; ++t) {
      const bool on = !spread || (lane / 16 == 0 && t == 0);
      a[t] = (_Float16)(on ? f * (spread ? 32.f : 1.f) * 0.0009765625f : 0.f);  // f * 2^-10 (x32 when a single product carries it)
      b[t] = (_Float16)(on ? 0.0009765625f * 0.00390625f : 0.f);                 // 2^-18: 32 * f 2^-10 * 2^-18 = f * 2^-23 ... scaled below
    }
    // ulp(1536) = 2^-13: we want 32 a b = f 2^-13  ->  a = f 2^-4, b = 2^-14 (normal float16)
    for (int t = 0; t < 8; ++t) {
      const bool on = !spread || (lane / 16 == 0 && t == 0);
      a[t] = (_Float16)(on ? f * (spread ? 32.f : 1.f) * 0.0625f : 0.f);
      b[t] = (_Float16)(on ? 6.103515625e-05f : 0.f);
    }
    f32x4 c = {1536.f, 1536.f, 1536.f, 1536.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[i] = (c[0] - 1536.f) * 8192.f;  // in ulps
  }
}

int main() {
  // (total addend in ulps; in the "32 equal products" mode each product is 1/32 of it)
  const float fr[] = {0.25f, 0.5f, 0.75f, 1.5f, -0.25f, -0.5f, -0.75f, -1.5f, 0.46875f, 0.53125f, -0.46875f, -0.53125f,
                      2.f, 3.f, 4.f, 6.f, 8.f, 12.f, 16.f, 24.f, 32.f, 48.f, -4.f, -8.f, -16.f, -24.f, -48.f};
  const int n = sizeof(fr) / 4;
  float *d, *o, h[64];
  hipMalloc(&d, sizeof(fr));
  hipMalloc(&o, sizeof(fr));
  hipMemcpy(d, fr, sizeof(fr), hipMemcpyHostToDevice);
  for (int spread = 0; spread < 2; ++spread) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n, spread);
    hipMemcpy(h, o, sizeof(fr), hipMemcpyDeviceToHost);
    printf("%s\n", spread ? "one product carries the whole addend:" : "32 equal products:");
    for (int i = 0; i < n; ++i) printf("  C + %+.5f ulp -> C %+g ulp\n", fr[i], h[i]);
  }
  return 0;
}
