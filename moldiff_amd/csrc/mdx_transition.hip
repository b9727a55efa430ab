// Per-step transition math + counter-based noise (gfx950).  One thread per row (K <= 8).
//   pos_posterior      models/transition.py:44-63   (ContigousTransition.get_prev_from_recon)
//   cat_posterior      models/transition.py:285-315 (GeneralCategoricalTransition.q_v_posterior, v0_prob=True)
//                      optionally fused with the log_softmax of models/model.py:291,297
//   gumbel_argmax      models/diffusion.py:79-85    (log_sample_categorical) + one-hot (transition.py:255)
//   philox_noise       replaces torch.rand_like / randn_like (diffusion.py:80, transition.py:60): Philox4x32-10
//                      keyed by (seed; global molecule id, local element, draw index, stream) so that a
//                      molecule's noise does not depend on how the batch is sharded over GPUs.
// Compiled with -ffp-contract=off: the reference evaluates these formulas with separate mul/add.
#include "mdx_kernels.h"
#include <algorithm>

#define MDX_MAXK 8

namespace {

__device__ __forceinline__ void pos_posterior_elem(const float* __restrict__ c0, const float* __restrict__ ct,
                                                   const float* __restrict__ sd, const float* __restrict__ xt,
                                                   const float* __restrict__ x0, const float* __restrict__ eps,
                                                   const int64_t* __restrict__ t, const int64_t* __restrict__ batch, int n,
                                                   float* __restrict__ out, int i) {
  if (i >= 3 * n) return;
  const int v = i / 3;
  const int64_t tv = t[batch[v]];
  const float mu = c0[tv] * x0[i] + ct[tv] * xt[i];
  const float x = mu + sd[tv] * eps[i];
  out[i] = (tv == 0) ? mu : x;
}
__global__ void pos_posterior_kernel(const float* __restrict__ c0, const float* __restrict__ ct,
                                     const float* __restrict__ sd, const float* __restrict__ xt,
                                     const float* __restrict__ x0, const float* __restrict__ eps,
                                     const int64_t* __restrict__ t, const int64_t* __restrict__ batch, int n,
                                     float* __restrict__ out) {
  pos_posterior_elem(c0, ct, sd, xt, x0, eps, t, batch, n, out, blockIdx.x * blockDim.x + threadIdx.x);
}

// the same posterior for rows of any width C (continuous categorical space: atom / bond features are real vectors, model.py:301-304)
__global__ void gauss_posterior_kernel(const float* __restrict__ c0, const float* __restrict__ ct, const float* __restrict__ sd,
                                       const float* __restrict__ xt, const float* __restrict__ x0, const float* __restrict__ eps,
                                       const int64_t* __restrict__ t, const int64_t* __restrict__ batch, int n, int C,
                                       float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * C) return;
  const int64_t tv = t[batch[i / C]];
  const float mu = c0[tv] * x0[i] + ct[tv] * xt[i];
  const float x = mu + sd[tv] * eps[i];
  out[i] = (tv == 0) ? mu : x;
}

template <int K>
__device__ __forceinline__ void cat_posterior_row(const float* __restrict__ qmats, const float* __restrict__ qT1, int T,
                                                  const float* __restrict__ in0, int is_logits, const float* __restrict__ log_vt,
                                                  const int64_t* __restrict__ t, const int64_t* __restrict__ batch, int n,
                                                  float* __restrict__ out, int i) {
  if (i >= n) return;
  const int64_t tv = t[batch[i]];
  const int64_t tm1 = tv > 0 ? tv - 1 : 0;
  float l0[K], lt[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    l0[k] = in0[(size_t)i * K + k];
    lt[k] = log_vt[(size_t)i * K + k];
  }
  if (is_logits) {  // log_softmax
    float m = l0[0];
#pragma unroll
    for (int k = 1; k < K; ++k) m = fmaxf(m, l0[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += expf(l0[k] - m);
    const float ls = logf(s);
#pragma unroll
    for (int k = 0; k < K; ++k) l0[k] = (l0[k] - m) - ls;
  }
  const float* Q1 = qT1 + (size_t)tv * K * K;
  const float* Q0 = qmats + (size_t)tm1 * K * K;
  float e0[K], et[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    e0[k] = expf(l0[k]);
    et[k] = expf(lt[k]);
  }
  float o[K];
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float f1 = 0.f, f2 = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      f1 += et[j] * Q1[j * K + k];
      f2 += e0[j] * Q0[j * K + k];
    }
    o[k] = fmaxf(logf(f1 + 1e-30f), -32.f) + fmaxf(logf(f2 + 1e-30f), -32.f);
    m = fmaxf(m, o[k]);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) s += expf(o[k] - m);
  const float lse = m + logf(s);
#pragma unroll
  for (int k = 0; k < K; ++k) out[(size_t)i * K + k] = (tv == 0) ? l0[k] : (o[k] - lse);
}
template <int K>
__global__ void cat_posterior_kernel(const float* __restrict__ qmats, const float* __restrict__ qT1, int T,
                                     const float* __restrict__ in0, int is_logits, const float* __restrict__ log_vt,
                                     const int64_t* __restrict__ t, const int64_t* __restrict__ batch, int n,
                                     float* __restrict__ out) {
  cat_posterior_row<K>(qmats, qT1, T, in0, is_logits, log_vt, t, batch, n, out, blockIdx.x * blockDim.x + threadIdx.x);
}

__device__ __forceinline__ void gumbel_argmax_row(const float* __restrict__ logits, const float* __restrict__ u, int K, int n,
                                                  int64_t* __restrict__ cls, float* __restrict__ onehot,
                                                  uint8_t* __restrict__ cls8, int i) {
  if (i >= n) return;
  int best = 0;
  float bv = -INFINITY;
  for (int k = 0; k < K; ++k) {
    const float g = -logf(-logf(u[(size_t)i * K + k] + 1e-30f) + 1e-30f);
    const float z = g + logits[(size_t)i * K + k];
    if (k == 0 || z > bv) {  // first maximum wins, like torch.argmax
      bv = z;
      best = k;
    }
  }
  if (cls) cls[i] = best;
  if (cls8) cls8[i] = (uint8_t)best;  // compact trajectory frame (one byte per atom / half-edge)
  if (onehot)
    for (int k = 0; k < K; ++k) onehot[(size_t)i * K + k] = (k == best) ? 1.f : 0.f;
}
__global__ void gumbel_argmax_kernel(const float* __restrict__ logits, const float* __restrict__ u, int K, int n,
                                     int64_t* __restrict__ cls, float* __restrict__ onehot, uint8_t* __restrict__ cls8) {
  gumbel_argmax_row(logits, u, K, n, cls, onehot, cls8, blockIdx.x * blockDim.x + threadIdx.x);
}

// Prior draw of the categorical chain (reference models/transition.py:331-339 sample_init): Gumbel-max over the SAME K float64
// logits log(init_prob + 1e-30).clamp_min(-32) for every row, the whole expression of diffusion.py:79-85 evaluated in float64
// as the reference does at this one place (its logits tensor is float64, so rand_like / log / argmax are too).  The uniforms
// are float64 (explicit, tests) or the float32 Philox draws of mdx_noise widened exactly.
struct PriorLogits { double v[8]; };
template <class U>
__global__ void prior_draw_kernel(const PriorLogits lg, const U* __restrict__ u, int K, int n, int64_t* __restrict__ cls,
                                  float* __restrict__ onehot, float* __restrict__ log_onehot, float log_off,
                                  uint8_t* __restrict__ cls8) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int best = 0;
  double bv = 0.0;
  for (int k = 0; k < K; ++k) {
    const double g = -log(-log((double)u[(size_t)i * K + k] + 1e-30) + 1e-30);
    const double z = g + lg.v[k];
    if (k == 0 || z > bv) {  // first maximum wins, like torch.argmax
      bv = z;
      best = k;
    }
  }
  if (cls) cls[i] = best;
  if (cls8) cls8[i] = (uint8_t)best;
  for (int k = 0; k < K; ++k) {
    if (onehot) onehot[(size_t)i * K + k] = (k == best) ? 1.f : 0.f;
    if (log_onehot) log_onehot[(size_t)i * K + k] = (k == best) ? 0.f : log_off;  // log(onehot.clamp(min=1e-30)), transition.py:338
  }
}

// The five transition launches of a sampling step (models/model.py:287-307) in one: thread i does position component i, atom
// row i and half-edge row i through the same row functions as the stand-alone kernels (a row's posterior is written and read
// back by the same thread).  MolDiff's class counts (8 atom types, 6 bond types) only; other counts take the separate kernels.
__global__ void step_transition_kernel(const StepTransArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  pos_posterior_elem(a.c0, a.ct, a.sd, a.pos, a.pred_pos, a.eps, a.t, a.batch_node, a.N, a.pos_next, i);
  if (i < a.N) {
    cat_posterior_row<8>(a.node_q, a.node_qT1, a.T, a.pred_node, 1, a.log_node, a.t, a.batch_node, a.N, a.log_node_next, i);
    gumbel_argmax_row(a.log_node_next, a.u_node, 8, a.N, nullptr, a.h_node_next, a.node_cls, i);
  }
  if (i < a.Eh) {
    cat_posterior_row<6>(a.edge_q, a.edge_qT1, a.T, a.pred_half, 1, a.log_half, a.t, a.batch_half, a.Eh, a.log_half_next, i);
    gumbel_argmax_row(a.log_half_next, a.u_half, 6, a.Eh, nullptr, a.h_half_next, a.half_cls, i);
  }
}

// 'uncertainty' guidance objective (models/model.py:322-324): U = sum_h log sigmoid(-logsumexp_k logits[h,k]);
// dU/dlogits[h,k] = -sigmoid(s_h) * softmax_k.   One thread per half-edge.
__global__ void uncertainty_grad_kernel(const float* __restrict__ logits, int K, int n, float* __restrict__ glogits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m = -INFINITY;
  for (int k = 0; k < K; ++k) m = fmaxf(m, logits[(size_t)i * K + k]);
  float sum = 0.f;
  for (int k = 0; k < K; ++k) sum += expf(logits[(size_t)i * K + k] - m);
  const float s = m + logf(sum);
  const float sig = 1.0f / (1.0f + expf(-s));
  for (int k = 0; k < K; ++k) glogits[(size_t)i * K + k] = -sig * (expf(logits[(size_t)i * K + k] - m) / sum);
}

__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = dst[i] + src[i];
}

struct u4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(u4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    u4 n = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }  // [0,1)

__global__ void philox_noise_kernel(uint64_t seed, int draw, const int* __restrict__ node_graph,
                                    const int* __restrict__ node_local, const int* __restrict__ he_graph,
                                    const int* __restrict__ he_local, const int64_t* __restrict__ mol_ids, int N, int Eh,
                                    int Kn, int Ke, float* __restrict__ eps_pos, float* __restrict__ u_node,
                                    float* __restrict__ u_half, int64_t* __restrict__ t_buf, int64_t t_val, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_buf && i < B) t_buf[i] = t_val;  // the step's time tensor rides along (one launch less per sampling step)
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  if (i < N) {
    const uint64_t mol = (uint64_t)mol_ids[node_graph[i]];
    const uint32_t loc = (uint32_t)node_local[i];
    if (eps_pos) {
      u4 c = {loc, (uint32_t)draw, (uint32_t)mol, (uint32_t)(mol >> 32) << 4 | 0u};
      const u4 r = philox4x32_10(c, k0, k1);
      const float u1 = ((float)(r.x >> 8) + 1.0f) * 5.9604644775390625e-08f;  // (0,1]
      const float u3 = ((float)(r.z >> 8) + 1.0f) * 5.9604644775390625e-08f;
      const float r1 = sqrtf(-2.0f * logf(u1)), r2 = sqrtf(-2.0f * logf(u3));
      const float a1 = 6.283185307179586f * u01(r.y), a2 = 6.283185307179586f * u01(r.w);
      eps_pos[3 * (size_t)i + 0] = r1 * cosf(a1);
      eps_pos[3 * (size_t)i + 1] = r1 * sinf(a1);
      eps_pos[3 * (size_t)i + 2] = r2 * cosf(a2);
    }
    if (u_node) {
      for (int b = 0; 4 * b < Kn; ++b) {
        u4 c = {loc * 2u + (uint32_t)b, (uint32_t)draw, (uint32_t)mol, (uint32_t)(mol >> 32) << 4 | 1u};
        const u4 r = philox4x32_10(c, k0, k1);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
        for (int j = 0; j < 4 && 4 * b + j < Kn; ++j) u_node[(size_t)i * Kn + 4 * b + j] = u01(rr[j]);
      }
    }
  }
  if (i < Eh && u_half) {
    const uint64_t mol = (uint64_t)mol_ids[he_graph[i]];
    const uint32_t loc = (uint32_t)he_local[i];
    for (int b = 0; 4 * b < Ke; ++b) {
      u4 c = {loc * 2u + (uint32_t)b, (uint32_t)draw, (uint32_t)mol, (uint32_t)(mol >> 32) << 4 | 2u};
      const u4 r = philox4x32_10(c, k0, k1);
      const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
      for (int j = 0; j < 4 && 4 * b + j < Ke; ++j) u_half[(size_t)i * Ke + 4 * b + j] = u01(rr[j]);
    }
  }
}

}  // namespace

void launch_gauss_posterior(const float* c0, const float* ct, const float* sd, const float* xt, const float* x0, const float* eps,
                            const int64_t* t, const int64_t* batch, int n, int C, float* out, hipStream_t s) {
  if (n <= 0 || C <= 0) return;
  hipLaunchKernelGGL(gauss_posterior_kernel, dim3(((size_t)n * C + 255) / 256), dim3(256), 0, s, c0, ct, sd, xt, x0, eps, t, batch, n,
                     C, out);
}

void launch_pos_posterior(const float* c0, const float* ct, const float* sd, const float* xt, const float* x0,
                          const float* eps, const int64_t* t, const int64_t* batch, int n, float* out, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pos_posterior_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, s, c0, ct, sd, xt, x0, eps, t, batch,
                     n, out);
}

void launch_cat_posterior(const float* qmats, const float* qT1, int K, int T, const float* in0, int is_logits,
                          const float* log_vt, const int64_t* t, const int64_t* batch, int n, float* out, hipStream_t s) {
  if (n <= 0) return;
  dim3 g((n + 255) / 256), b(256);
#define MDX_CP(KK)                                                                                                     \
  case KK:                                                                                                             \
    hipLaunchKernelGGL(cat_posterior_kernel<KK>, g, b, 0, s, qmats, qT1, T, in0, is_logits, log_vt, t, batch, n, out); \
    break;
  switch (K) {
    MDX_CP(2) MDX_CP(3) MDX_CP(4) MDX_CP(5) MDX_CP(6) MDX_CP(7) MDX_CP(8)
    default: break;
  }
#undef MDX_CP
}

// ---- training: q(v_t | v_0) draw of a clean batch in one launch (round 6) -------------------------------------------------------------
// Reference models/transition.py:266-283 (add_noise -> q_vt_sample -> q_vt_pred, index_to_log_onehot models/diffusion.py:53-57,
// log_sample_categorical :79-85): log_v0 = log(clamp(onehot(v), 1e-30)); logits = log(exp(log_v0) Q[t] + 1e-30).clamp_min(-32);
// cls = argmax(logits + Gumbel(u)); outputs onehot(cls), log(clamp(onehot(cls), 1e-30)), log_v0.  One thread per row, the sum over the K
// source classes in index order; log_off = log(1e-30) as torch computes it in fp32 (the caller passes torch's value).  Class ids >= K are
// clamped to K - 1 (the caller's range check reports them: diffusion.deferred_class_checks).
template <int K>
__global__ void cat_add_noise_kernel(const float* __restrict__ qmats, const int64_t* __restrict__ v, const int64_t* __restrict__ t,
                                     const int64_t* __restrict__ batch, const float* __restrict__ u, int n, float log_off,
                                     float* __restrict__ onehot, float* __restrict__ log_vt, float* __restrict__ log_v0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t cv = v[i];
  cv = cv < 0 ? 0 : (cv >= K ? K - 1 : cv);
  const float* Q = qmats + (size_t)t[batch[i]] * K * K;
  float l0[K], e0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    l0[k] = (k == (int)cv) ? 0.f : log_off;
    e0[k] = expf(l0[k]);
    log_v0[(size_t)i * K + k] = l0[k];
  }
  int best = 0;
  float bv = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float f = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) f += e0[j] * Q[j * K + k];
    const float lg = fmaxf(logf(f + 1e-30f), -32.f);
    const float g = -logf(-logf(u[(size_t)i * K + k] + 1e-30f) + 1e-30f);
    const float z = g + lg;
    if (k == 0 || z > bv) {  // first maximum wins, like torch.argmax
      bv = z;
      best = k;
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    onehot[(size_t)i * K + k] = (k == best) ? 1.f : 0.f;
    log_vt[(size_t)i * K + k] = (k == best) ? 0.f : log_off;
  }
}
void launch_cat_add_noise(const float* qmats, int K, const int64_t* v, const int64_t* t, const int64_t* batch, const float* u, int n,
                          float log_off, float* onehot, float* log_vt, float* log_v0, hipStream_t s) {
  if (n <= 0) return;
  dim3 g((n + 127) / 128), b(128);
#define MDX_CAN(KK)                                                                                                                       \
  case KK:                                                                                                                                \
    hipLaunchKernelGGL(cat_add_noise_kernel<KK>, g, b, 0, s, qmats, v, t, batch, u, n, log_off, onehot, log_vt, log_v0);                   \
    break;
  switch (K) {
    MDX_CAN(2) MDX_CAN(3) MDX_CAN(4) MDX_CAN(5) MDX_CAN(6) MDX_CAN(7) MDX_CAN(8)
    default: break;
  }
#undef MDX_CAN
}

// ---- training: the categorical loss rows and their gradient in one launch (round 6) ---------------------------------------------------
// Reference models/model.py:170-189: log_recon = log_softmax(logits); post_true = q_v_posterior(log_v0, log_vt); post_pred =
// q_v_posterior(log_recon, log_vt) (models/transition.py:285-315); row term = KL(post_true || post_pred) for t > 0, -sum exp(log_v0)
// post_pred at t == 0 (compute_v_Lt :317-327 with models/diffusion.py categorical_kl / log_categorical).  The layer-by-layer torch
// evaluation of that tail and of its autograd costs ~190 launches of (rows x K <= 8) element-wise kernels per training step; here a
// thread owns a row, evaluates both posteriors with the arithmetic of cat_posterior_row, and back-propagates by hand:
//   g_pp = d row / d post_pred = -(mask exp(log_v0) + (1 - mask) exp(post_true))
//   t > 0: post_pred = o - logsumexp(o), o = A + B, B = max(log(f2 + eps), -32), f2 = exp(log_recon) Q[t-1]   (A: no logits inside)
//          g_o = g_pp - exp(post_pred) sum(g_pp); g_f2 = g_o [log(f2 + eps) >= -32] / (f2 + eps); g_lr = exp(log_recon) (Q[t-1] g_f2)
//   t = 0: post_pred = log_recon (torch.where routes the whole gradient there): g_lr = g_pp
//   log_recon = x - logsumexp(x): g_x = g_lr - softmax(x) sum(g_lr)
// (what torch.autograd computes through transition.q_v_posterior_autograd, clamp_min passing the gradient where its input >= the bound).
template <int K>
__global__ void cat_loss_kernel(const float* __restrict__ qmats, const float* __restrict__ qT1, const float* __restrict__ logits,
                                const float* __restrict__ log_vt, const float* __restrict__ log_v0, const int64_t* __restrict__ t,
                                const int64_t* __restrict__ batch, int n, float* __restrict__ row_loss, float* __restrict__ dlogits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t tv = t[batch[i]];
  const int64_t tm1 = tv > 0 ? tv - 1 : 0;
  const float* Q1 = qT1 + (size_t)tv * K * K;
  const float* Q0 = qmats + (size_t)tm1 * K * K;
  float x[K], lt[K], l0[K], lr[K], er[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    x[k] = logits[(size_t)i * K + k];
    lt[k] = log_vt[(size_t)i * K + k];
    l0[k] = log_v0[(size_t)i * K + k];
  }
  {  // log_softmax
    float m = x[0];
#pragma unroll
    for (int k = 1; k < K; ++k) m = fmaxf(m, x[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += expf(x[k] - m);
    const float ls = logf(s);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      lr[k] = (x[k] - m) - ls;
      er[k] = expf(lr[k]);
    }
  }
  float et[K], e0[K], A[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    et[k] = expf(lt[k]);
    e0[k] = expf(l0[k]);
  }
  // the two posteriors share A = max(log(exp(log_vt) Q1 + eps), -32)
  float pt[K], pp[K], f2p[K];
  {
    float ot[K], op[K], mt = -INFINITY, mp = -INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float f1 = 0.f, f2t = 0.f, f2 = 0.f;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        f1 += et[j] * Q1[j * K + k];
        f2t += e0[j] * Q0[j * K + k];
        f2 += er[j] * Q0[j * K + k];
      }
      A[k] = fmaxf(logf(f1 + 1e-30f), -32.f);
      ot[k] = A[k] + fmaxf(logf(f2t + 1e-30f), -32.f);
      op[k] = A[k] + fmaxf(logf(f2 + 1e-30f), -32.f);
      f2p[k] = f2;
      mt = fmaxf(mt, ot[k]);
      mp = fmaxf(mp, op[k]);
    }
    float st = 0.f, sp = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      st += expf(ot[k] - mt);
      sp += expf(op[k] - mp);
    }
    const float lset = mt + logf(st), lsep = mp + logf(sp);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      pt[k] = (tv == 0) ? l0[k] : (ot[k] - lset);
      pp[k] = (tv == 0) ? lr[k] : (op[k] - lsep);
    }
  }
  const float mask = (tv == 0) ? 1.f : 0.f;
  float kl = 0.f, nll = 0.f, gpp[K], gsum = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float ept = expf(pt[k]);
    kl += ept * (pt[k] - pp[k]);
    nll += e0[k] * pp[k];
    gpp[k] = -(mask * e0[k] + (1.f - mask) * ept);
    gsum += gpp[k];
  }
  row_loss[i] = mask * (-nll) + (1.f - mask) * kl;
  float glr[K];
  if (tv == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) glr[k] = gpp[k];
  } else {
    float gf2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float go = gpp[k] - expf(pp[k]) * gsum;
      const float u = f2p[k] + 1e-30f;
      gf2[k] = (logf(u) >= -32.f) ? go / u : 0.f;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < K; ++k) acc += gf2[k] * Q0[j * K + k];
      glr[j] = er[j] * acc;
    }
  }
  float gl = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) gl += glr[k];
#pragma unroll
  for (int k = 0; k < K; ++k) dlogits[(size_t)i * K + k] = glr[k] - er[k] * gl;
}

void launch_cat_loss(const float* qmats, const float* qT1, int K, const float* logits, const float* log_vt, const float* log_v0,
                     const int64_t* t, const int64_t* batch, int n, float* row_loss, float* dlogits, hipStream_t s) {
  if (n <= 0) return;
  dim3 g((n + 127) / 128), b(128);
#define MDX_CL(KK)                                                                                                                  \
  case KK:                                                                                                                          \
    hipLaunchKernelGGL(cat_loss_kernel<KK>, g, b, 0, s, qmats, qT1, logits, log_vt, log_v0, t, batch, n, row_loss, dlogits);         \
    break;
  switch (K) {
    MDX_CL(2) MDX_CL(3) MDX_CL(4) MDX_CL(5) MDX_CL(6) MDX_CL(7) MDX_CL(8)
    default: break;
  }
#undef MDX_CL
}

void launch_step_transition(const StepTransArgs& a, hipStream_t s) {
  const int n = std::max(3 * a.N, a.Eh);
  if (n > 0) hipLaunchKernelGGL(step_transition_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
}

void launch_uncertainty_grad(const float* logits, int K, int n, float* glogits, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(uncertainty_grad_kernel, dim3((n + 255) / 256), dim3(256), 0, s, logits, K, n, glogits);
}

void launch_add_inplace(float* dst, const float* src, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n);
}

void launch_gumbel_argmax(const float* logits, const float* u, int K, int n, int64_t* cls, float* onehot, hipStream_t s,
                          uint8_t* cls8) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gumbel_argmax_kernel, dim3((n + 255) / 256), dim3(256), 0, s, logits, u, K, n, cls, onehot, cls8);
}

void launch_prior_draw(const double* logits64, int K, const void* u, bool u_f64, int n, int64_t* cls, float* onehot, float* log_onehot,
                       float log_off, uint8_t* cls8, hipStream_t s) {
  if (n <= 0) return;
  PriorLogits lg{};
  for (int k = 0; k < K && k < 8; ++k) lg.v[k] = logits64[k];
  const dim3 g((n + 255) / 256), b(256);
  if (u_f64) hipLaunchKernelGGL(prior_draw_kernel<double>, g, b, 0, s, lg, (const double*)u, K, n, cls, onehot, log_onehot, log_off, cls8);
  else hipLaunchKernelGGL(prior_draw_kernel<float>, g, b, 0, s, lg, (const float*)u, K, n, cls, onehot, log_onehot, log_off, cls8);
}

namespace {
__global__ void fill_i64_kernel(int64_t* __restrict__ p, int64_t v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
}  // namespace
void launch_fill_i64(int64_t* p, int64_t v, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(fill_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
}

void launch_philox_noise(uint64_t seed, int draw, const int* node_graph, const int* node_local, const int* he_graph,
                         const int* he_local, const int64_t* mol_ids, int N, int Eh, int Kn, int Ke, float* eps_pos,
                         float* u_node, float* u_half, hipStream_t s, int64_t* t_buf, int64_t t_val, int B) {
  const int n = std::max(N > Eh ? N : Eh, t_buf ? B : 0);
  if (n <= 0) return;
  hipLaunchKernelGGL(philox_noise_kernel, dim3((n + 255) / 256), dim3(256), 0, s, seed, draw, node_graph, node_local,
                     he_graph, he_local, mol_ids, N, Eh, Kn, Ke, eps_pos, u_node, u_half, t_buf, t_val, B);
}
