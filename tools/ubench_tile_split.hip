// Micro-benchmark for the NEXT design of the split-precision edge kernels (round 4): the 256 -> 256 (LayerNorm,
// ReLU) chain of tools/ubench_split.hip as a WORKGROUP TILE instead of a row-owner wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_tile_split ubench_tile_split.hip
//
// Why: the row-owner split kernels are bound by the per-wave weight stream through the CU's L1 port (profiles/r4_weight_ablation.txt):
// a wave that owns 16 R rows and ALL 256 output features pulls the whole 256 KiB matrix for 16 R rows -- 170 / R bytes per clock
// and CU against ~41 measured.  R is capped at 2 by the accumulators (16 tiles x R).  Here a workgroup of NW waves owns 16 RT rows;
// wave w computes feature tiles [FTW w, FTW w + FTW) (FTW = 16 / NW) for ALL RT row tiles, so the same accumulator budget buys
// RT = 4 or 8 rows tiles per weight fragment: 42 / 21 bytes per clock and CU.  The price: the activations go through LDS between
// layers (as float16 hi / lo fragments in B-operand layout -- a lane's OWN accumulators are its 8 halves, so the write is one
// ds_write_b128 per fragment and lane with no cross-lane traffic) and LayerNorm needs one exchange of per-row partial statistics:
// two barriers per layer.
//
//   LDS:  X[g 0..7][rt][h][lane][8 halves]   (RT x 16 KiB),  stats[NW][16 RT] x (mean, M2)
//   stream of wave w, layer m: for g: for ftp < FTW/2: half-step (hi fragments of tiles FTW w + 2 ftp, +1), half-step (lo, x 2^11)
//   per k-group g:  2 RT ds_read_b128 (B operand hi, lo of every row tile), FTW x RT x 3 MFMAs
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}
__device__ __forceinline__ float red_q(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
__device__ __forceinline__ void wg_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int NROWS = 256 * 768;  // input rows (a multiple of every tile height)
constexpr float LO_UP = 2048.0f, LO_DOWN = 1.0f / 2048.0f;

template <int RT, int NW, int DEPTH, bool LN, int WPS>
__global__ __launch_bounds__(64 * NW, WPS) void tile_chain_kernel(const float* __restrict__ X, const void* __restrict__ Wp,
                                                             const float* __restrict__ gb, float* __restrict__ out, int reps, int nl,
                                                             int nmat, float eps, int nrows) {
  constexpr int FTW = 16 / NW, FP = FTW / 2, HS = 8 * FP * 2;  // feature tiles / pairs per wave, half-steps per wave and layer
  constexpr int ROWS = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_float* xs = (lds_float*)smem;                      // RT * 4096 floats
  lds_float* st = (lds_float*)smem + RT * 4096;          // [NW][ROWS][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  const unsigned lane_off = 16u * lane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Wp), 0, -1, 0x00020000);
  lds_float* xl = xs + 4 * lane;  // this lane's 16 bytes inside a fragment
  auto frag = [&](int g, int rt, int h) { return xl + ((g * RT + rt) * 2 + h) * 256; };

  for (int rep = 0; rep < reps; ++rep) {
    const size_t row0 = (((size_t)blockIdx.x * reps + rep) * ROWS) % (size_t)nrows;  // the timing runs wrap around the input
    // ---- input rows -> split fragments of this wave's k-groups (features FTW*16*wave ...)
    {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int gp = 0; gp < FP; ++gp) {
          const int g = FP * wave + gp;
          const float* src = X + (row0 + 16 * rt + c) * 256 + 32 * g + 4 * q;
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 16);
          h8 hi, lo;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float v = t < 4 ? v0[t] : v1[t - 4];
            const _Float16 hh = (_Float16)v;
            hi[t] = hh;
            lo[t] = (_Float16)((v - (float)hh) * LO_UP);
          }
          *(lds_f32x4*)frag(g, rt, 0) = __builtin_bit_cast(f32x4, hi);
          *(lds_f32x4*)frag(g, rt, 1) = __builtin_bit_cast(f32x4, lo);
        }
    }
    wg_sync();
    int m = 0;
    f32x4 y[FTW][RT];
    for (int l = 0; l < nl; ++l) {
      f32x4 t[FTW][RT];
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) y[ft][rt] = t[ft][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      // stream of (layer m, wave): HS half-steps of 2 KiB
      const int sbase = (__builtin_amdgcn_readfirstlane(m) * NW + wave) * HS * 2048;
      f32x4 ring[DEPTH][2];
#pragma unroll
      for (int p = 0; p < DEPTH; ++p) {
        ring[p][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, sbase + (2 * p) * 1024, 0));
        ring[p][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, sbase + (2 * p + 1) * 1024, 0));
      }
      h8 bh[RT], bl[RT];
      static_for<0, HS>([&](auto pc) {
        constexpr int p = decltype(pc)::value, g = p / (2 * FP), ftp = (p / 2) % FP, h = p % 2;
        if constexpr (ftp == 0 && h == 0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            bh[rt] = __builtin_bit_cast(h8, *(lds_f32x4*)frag(g, rt, 0));
            bl[rt] = __builtin_bit_cast(h8, *(lds_f32x4*)frag(g, rt, 1));
          }
        }
        const h8 a0 = __builtin_bit_cast(h8, ring[p % DEPTH][0]), a1 = __builtin_bit_cast(h8, ring[p % DEPTH][1]);
        if constexpr (p + DEPTH < HS) {
          ring[p % DEPTH][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, sbase + (2 * (p + DEPTH)) * 1024, 0));
          ring[p % DEPTH][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane_off, sbase + (2 * (p + DEPTH) + 1) * 1024, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (h == 0) {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            y[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bh[rt], y[2 * ftp][rt], 0, 0, 0);
            y[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh[rt], y[2 * ftp + 1][rt], 0, 0, 0);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            t[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bl[rt], t[2 * ftp][rt], 0, 0, 0);
            t[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bl[rt], t[2 * ftp + 1][rt], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            t[2 * ftp][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bh[rt], t[2 * ftp][rt], 0, 0, 0);
            t[2 * ftp + 1][rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bh[rt], t[2 * ftp + 1][rt], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) y[ft][rt] = y[ft][rt] + t[ft][rt] * f32x4{LO_DOWN, LO_DOWN, LO_DOWN, LO_DOWN};
      if (LN) {
        // per-row partial statistics over this wave's FTW*16 features: (mean, M2); combined across waves with Chan's formula
        float mw[RT], m2w[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float s = 0.f;
#pragma unroll
          for (int ft = 0; ft < FTW; ++ft) s += (y[ft][rt][0] + y[ft][rt][1]) + (y[ft][rt][2] + y[ft][rt][3]);
          mw[rt] = red_q(s) * (1.0f / (16 * FTW));
          float d2 = 0.f;
#pragma unroll
          for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float d = y[ft][rt][r] - mw[rt];
              d2 = fmaf(d, d, d2);
            }
          m2w[rt] = red_q(d2);
          if (q == 0) {
            st[(wave * ROWS + 16 * rt + c) * 2] = mw[rt];
            st[(wave * ROWS + 16 * rt + c) * 2 + 1] = m2w[rt];
          }
        }
        wg_sync();  // all waves are past their GEMM: X may be overwritten after this, and the partials are visible
        const float* gbp = gb + (size_t)m * 512;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float mean = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) mean += st[(w * ROWS + 16 * rt + c) * 2];
          mean *= 1.0f / NW;
          float M2 = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            const float d = st[(w * ROWS + 16 * rt + c) * 2] - mean;
            M2 += st[(w * ROWS + 16 * rt + c) * 2 + 1] + (16 * FTW) * d * d;
          }
          const float rstd = 1.0f / sqrtf(M2 * (1.0f / 256) + eps);
#pragma unroll
          for (int ft = 0; ft < FTW; ++ft) {
            const int f = 16 * (FTW * wave + ft) + 4 * q;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gbp + f), bt = *reinterpret_cast<const f32x4*>(gbp + 256 + f);
#pragma unroll
            for (int r = 0; r < 4; ++r) y[ft][rt][r] = fmaxf((y[ft][rt][r] - mean) * rstd * gm[r] + bt[r], 0.f);
          }
        }
      } else {
        wg_sync();
      }
      // ---- own features -> split fragments of the next layer's operand
#pragma unroll
      for (int gp = 0; gp < FP; ++gp)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          h8 hi, lo;
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) {
            const float v = y[2 * gp + tt / 4][rt][tt % 4];
            const _Float16 hh = (_Float16)v;
            hi[tt] = hh;
            lo[tt] = (_Float16)((v - (float)hh) * LO_UP);
          }
          *(lds_f32x4*)frag(FP * wave + gp, rt, 0) = __builtin_bit_cast(f32x4, hi);
          *(lds_f32x4*)frag(FP * wave + gp, rt, 1) = __builtin_bit_cast(f32x4, lo);
        }
      wg_sync();
      m = (m + 1 == nmat) ? 0 : m + 1;
    }
#pragma unroll
    for (int ft = 0; ft < FTW; ++ft)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
        *reinterpret_cast<f32x4*>(out + (row0 + 16 * rt + c) * 256 + 16 * (FTW * wave + ft) + 4 * q) = y[ft][rt];
  }
}

// ---------------------------------------------------------------- host ----------------------------------------------------
static uint16_t f2h(float f) {
  _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
static float h2f(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}
struct Problem {
  int nmat;
  std::vector<float> W, gb;
};
template <int NW>
static void pack_tile(const Problem& P, std::vector<uint16_t>& out) {
  constexpr int FTW = 16 / NW, FP = FTW / 2, HS = 8 * FP * 2;
  out.assign((size_t)P.nmat * NW * HS * 1024, 0);
  for (int m = 0; m < P.nmat; ++m)
    for (int w = 0; w < NW; ++w)
      for (int g = 0; g < 8; ++g)
        for (int ftp = 0; ftp < FP; ++ftp)
          for (int j = 0; j < 2; ++j)
            for (int lane = 0; lane < 64; ++lane)
              for (int t = 0; t < 8; ++t) {
                const int q = lane >> 4, c = lane & 15, f = 16 * (FTW * w + 2 * ftp + j) + c, k = 32 * g + 16 * (t / 4) + 4 * q + t % 4;
                const float wt = P.W[(size_t)m * 65536 + f * 256 + k];
                const uint16_t hi = f2h(wt), lo = f2h((wt - h2f(hi)) * 2048.0f);
                const size_t hs = ((size_t)(m * NW + w) * HS) + (g * FP + ftp) * 2;
                out[((hs * 2 + j) * 64 + lane) * 8 + t] = hi;
                out[(((hs + 1) * 2 + j) * 64 + lane) * 8 + t] = lo;
              }
}
static void ref_chain(const Problem& P, const float* x0, int nl, bool ln, double* y) {
  std::vector<double> x(x0, x0 + 256), t(256);
  for (int l = 0; l < nl; ++l) {
    const int m = l % P.nmat;
    for (int f = 0; f < 256; ++f) {
      double s = 0;
      for (int k = 0; k < 256; ++k) s += (double)P.W[(size_t)m * 65536 + f * 256 + k] * x[k];
      t[f] = s;
    }
    if (ln) {
      double mean = 0, var = 0;
      for (int f = 0; f < 256; ++f) mean += t[f];
      mean /= 256;
      for (int f = 0; f < 256; ++f) var += (t[f] - mean) * (t[f] - mean);
      var /= 256;
      const double rstd = 1.0 / sqrt(var + 1e-5);
      for (int f = 0; f < 256; ++f) {
        const double v = (t[f] - mean) * rstd * P.gb[(size_t)m * 512 + f] + P.gb[(size_t)m * 512 + 256 + f];
        t[f] = v > 0 ? v : 0;
      }
    }
    x = t;
  }
  for (int f = 0; f < 256; ++f) y[f] = x[f];
}

template <int RT, int NW, int DEPTH, bool LN, int WPC>
static void run(const char* name, const float* dX, const void* dW, const float* dgb, float* dout, const Problem& P,
                const std::vector<double>& ref, int nl_check, int nref) {
  constexpr int wgs_per_cu = WPC;
  auto kern = tile_chain_kernel<RT, NW, DEPTH, LN, WPC * NW / 4>;
  const int ROWS = 16 * RT;
  const size_t lds = (size_t)(RT * 4096 + NW * ROWS * 2) * 4;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void*)kern);
  const int gcheck = (nref + ROWS - 1) / ROWS;
  hipLaunchKernelGGL(kern, dim3(gcheck), dim3(64 * NW), lds, 0, dX, dW, dgb, dout, 1, nl_check, P.nmat, 1e-5f, NROWS);
  std::vector<float> o((size_t)nref * 256);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  double emax = 0, esum = 0;
  for (size_t i = 0; i < o.size(); ++i) {
    const double e = fabs((double)o[i] - ref[i]);
    emax = e > emax ? e : emax;
    esum += e * e;
  }
  const int grid = 256 * wgs_per_cu * 3, nl = 6, reps = 4096 / ROWS / wgs_per_cu * 2;
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, dX, dW, dgb, dout, reps, nl, P.nmat, 1e-5f, NROWS);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, 0, dX, dW, dgb, dout, reps, nl, P.nmat, 1e-5f, NROWS);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
  }
  const double flop = (double)grid * reps * ROWS * nl * 2.0 * 256 * 256;
  printf("%-30s rows/WG=%3d waves=%d depth=%d WGs/CU=%d LN=%d vgpr=%3d scratch=%d lds=%zuK : %7.3f ms %7.1f eq.TFLOP/s (%.2fx of 157.3, %.2f of 833) | max err %.3e rms %.3e\n",
         name, ROWS, NW, DEPTH, wgs_per_cu, (int)LN, fa.numRegs, (int)fa.localSizeBytes, lds / 1024, best, flop / best / 1e9,
         flop / best / 1e9 / 157.3, flop / best / 1e9 / 833.3, emax, sqrt(esum / o.size()));
  fflush(stdout);
}

int main() {
  const int nmat = 6, nl_check = 6, nref = 128;
  Problem P;
  P.nmat = nmat;
  P.W.resize((size_t)nmat * 65536);
  P.gb.resize((size_t)nmat * 512);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  };
  for (auto& w : P.W) w = (float)(rnd() * 0.0625 * 1.7);
  for (int m = 0; m < nmat; ++m)
    for (int f = 0; f < 256; ++f) P.gb[(size_t)m * 512 + f] = (float)(1.0 + 0.3 * rnd()), P.gb[(size_t)m * 512 + 256 + f] = (float)(0.2 * rnd());
  const size_t rows = NROWS;
  std::vector<float> hX(rows * 256);
  for (auto& v : hX) v = (float)(rnd() * 2.0);
  std::vector<double> refLN((size_t)nref * 256), refNo((size_t)nref * 256);
  for (int r = 0; r < nref; ++r) {
    ref_chain(P, hX.data() + (size_t)r * 256, nl_check, true, refLN.data() + (size_t)r * 256);
    ref_chain(P, hX.data() + (size_t)r * 256, 2, false, refNo.data() + (size_t)r * 256);
  }
  std::vector<uint16_t> p2, p4, p8;
  pack_tile<2>(P, p2);
  pack_tile<4>(P, p4);
  pack_tile<8>(P, p8);
  float *dX, *dout, *dgb;
  void *dW2, *dW4, *dW8;
  hipMalloc(&dX, hX.size() * 4);
  hipMalloc(&dout, hX.size() * 4);
  hipMalloc(&dgb, P.gb.size() * 4);
  hipMalloc(&dW2, p2.size() * 2);
  hipMemcpy(dW2, p2.data(), p2.size() * 2, hipMemcpyHostToDevice);
  hipMalloc(&dW4, p4.size() * 2);
  hipMalloc(&dW8, p8.size() * 2);
  hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dgb, P.gb.data(), P.gb.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW4, p4.data(), p4.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dW8, p8.data(), p8.size() * 2, hipMemcpyHostToDevice);
  printf("workgroup-tile split chain (6 layers 256 -> 256, LayerNorm + ReLU); row-owner figures of the same chain: tools/ubench_split.hip\n");
  // wave PAIRS: 32 rows, two waves with half of the output features each (64 accumulator registers per wave): 4 pairs per CU = 2 waves per
  // SIMD at the weight stream of 32 rows x 1 wave
  run<2, 2, 4, true, 4>("pair 32 rows, 2 waves", dX, dW2, dgb, dout, P, refLN, nl_check, nref);
  run<2, 2, 8, true, 4>("pair 32 rows, 2 waves", dX, dW2, dgb, dout, P, refLN, nl_check, nref);
  run<2, 2, 4, false, 4>("pair 32 rows, 2 waves, no LN", dX, dW2, dgb, dout, P, refNo, 2, nref);
  run<4, 2, 4, true, 2>("pair 64 rows, 2 waves", dX, dW2, dgb, dout, P, refLN, nl_check, nref);
  run<4, 4, 4, true, 2>("tile 64 rows, 4 waves", dX, dW4, dgb, dout, P, refLN, nl_check, nref);
  run<4, 4, 8, true, 2>("tile 64 rows, 4 waves", dX, dW4, dgb, dout, P, refLN, nl_check, nref);
  run<2, 4, 4, true, 4>("tile 32 rows, 4 waves", dX, dW4, dgb, dout, P, refLN, nl_check, nref);
  run<8, 4, 4, true, 1>("tile 128 rows, 4 waves", dX, dW4, dgb, dout, P, refLN, nl_check, nref);
  run<4, 8, 4, true, 2>("tile 64 rows, 8 waves", dX, dW8, dgb, dout, P, refLN, nl_check, nref);
  run<8, 8, 4, true, 1>("tile 128 rows, 8 waves", dX, dW8, dgb, dout, P, refLN, nl_check, nref);
  run<4, 4, 4, false, 2>("tile 64 rows, 4 waves, no LN", dX, dW4, dgb, dout, P, refNo, 2, nref);
  return 0;
}
