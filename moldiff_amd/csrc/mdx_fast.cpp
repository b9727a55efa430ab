// Host-side fast path of the training step (round 6): the operator BODIES of moldiff_amd/train_ops.py in C++.
//
// A training step issues ~850 launches through ~190 autograd nodes; with the bodies in Python (tensor normalisation, output allocation,
// ctypes marshalling, gradient-sink bookkeeping) the host needs ~20 ms per step -- as long as the GPU does (profiles/r6_train_host_profile.txt:
// a Linear node 32 us, a fused BondFFN node 110 us forward).  This module is the same logic against the same C ABI (include/moldiff_hip.h,
// libmoldiff_hip.so), a pybind11 extension of the torch build in this image:
//   * the gradient sink and the weight-gradient queue (train_ops.grad_sink / _flush_wgrads / flush_grad_sink) as C++ state,
//   * Linear, Linear+LayerNorm+ReLU and the element-wise operators as torch::autograd::Function nodes,
//   * forward / backward bodies of the four fused row-owner operators (argument structs, buffers, queue entries, segment sums).
// It covers the float16 autocast mode with float16 containers inside a gradient sink (Trainer.step with precision='fp16', the reference's
// use_amp: True); every other mode keeps the Python bodies.  torch is plumbing here too: allocation, autograd edges, the current stream.
// Reference lines replaced are those the Python bodies cite (models/common.py:181-201, models/graph.py:29-55,133-141,268-295,384-396).
#include <torch/extension.h>
#include <torch/csrc/autograd/custom_function.h>
#include <c10/hip/HIPStream.h>

#include <array>
#include <unordered_set>
#include <vector>

#include "../../include/moldiff_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline void chk(int rc) {
  if (rc != 0) throw std::runtime_error(std::string("libmoldiff_hip error ") + std::to_string(rc) + ": " + mdx_last_error());
}
inline void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
inline void* P(const Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
inline int64_t A(const Tensor& t) { return t.defined() ? (int64_t)(uintptr_t)t.data_ptr() : 0; }
inline int H(const Tensor& t) { return (t.defined() && t.scalar_type() == at::kHalf) ? 1 : 0; }
inline void need_gpu(const Tensor& t) {
  TORCH_CHECK(!t.defined() || t.is_cuda(), "moldiff_amd runs on a ROCm device only (got a CPU tensor); there is no CPU fallback.");
}
// train_ops._rows: fp32 or float16 container, unit column stride (row stride arbitrary)
inline Tensor rows(const Tensor& t) {
  need_gpu(t);
  Tensor r = t;
  if (r.scalar_type() != at::kFloat && r.scalar_type() != at::kHalf) r = r.to(at::kFloat);
  return r.stride(-1) == 1 ? r : r.contiguous();
}
// train_ops._t: own container type, contiguous
inline Tensor tc(const Tensor& t) {
  need_gpu(t);
  Tensor r = t;
  if (r.scalar_type() != at::kFloat && r.scalar_type() != at::kHalf) r = r.to(at::kFloat);
  return r.is_contiguous() ? r : r.contiguous();
}
// train_ops._c: fp32, contiguous
inline Tensor fc(const Tensor& t) {
  need_gpu(t);
  Tensor r = t.scalar_type() == at::kFloat ? t : t.to(at::kFloat);
  return r.is_contiguous() ? r : r.contiguous();
}
inline Tensor aligned_rows(const Tensor& t, int64_t ld_mult, int64_t ptr_mult) {
  Tensor r = rows(t);
  if (r.stride(0) % ld_mult || ((uintptr_t)r.data_ptr()) % ptr_mult) r = r.contiguous();
  return r;
}
inline Tensor half_empty(int64_t E, int64_t F, const Tensor& like) { return at::empty({E, F}, like.options().dtype(at::kHalf)); }

constexpr int64_t RED_CHUNK = 256;  // == csrc/mdx_train.hip

struct Job {
  Tensor g, x;
  int64_t plan[8];
  int64_t M, N, K, dt, dst_w, ldw, dst_b, rk;
};

struct State {
  // precision of the running step (train_ops._AMP): half kind (2 = float16), autocast, float16 containers
  int amp0 = 0, amp_auto = 0, amp_store = 0;
  // gradient sink
  bool sink = false;
  int64_t s_data = 0, s_grad = 0, s_nbytes = 0;
  std::vector<int64_t> recs, recs2;
  int64_t blocks = 0, blocks2 = 0;
  std::unordered_set<int64_t> seen;
  std::vector<Tensor> keep;
  Tensor like;  // any tensor of the sink's device (allocation options)
  // weight-gradient queue
  std::vector<Job> wq;
  int64_t wq_bytes = 0, wgrad_rows = 2048, wq_cap = (int64_t)24 << 30;
  bool wq_on = true;
  // transposed parameters (train_ops.TransposedParams)
  int64_t wt_base = 0, wt_nbytes = 0, wt_buf = 0;
  std::vector<int64_t> wt_off;                       // sorted byte offsets
  std::vector<std::array<int64_t, 4>> wt_ent;        // (element offset in the parameter buffer, R, C, element offset of W^T) per entry
  int64_t n_launch = 0;
} S;

// a node's backward runs in the precision of its forward (train_ops: `with precision(ctx.prec)`), whatever the mode is by then
inline int64_t amp_pack() { return S.amp0 | (S.amp_auto << 4) | (S.amp_store << 5); }
struct AmpGuard {
  int a0, a1, a2;
  explicit AmpGuard(int64_t packed) : a0(S.amp0), a1(S.amp_auto), a2(S.amp_store) {
    S.amp0 = (int)(packed & 15), S.amp_auto = (int)((packed >> 4) & 1), S.amp_store = (int)((packed >> 5) & 1);
  }
  ~AmpGuard() { S.amp0 = a0, S.amp_auto = a1, S.amp_store = a2; }
};

bool fast_mode() { return S.sink && S.amp0 == 2 && S.amp_auto && S.amp_store; }

int64_t sink_dst(const Tensor& t) {
  if (!S.sink || !t.defined()) return 0;
  const int64_t off = A(t) - S.s_data;
  return (off >= 0 && off < S.s_nbytes) ? S.s_grad + off : 0;
}

void flush_sink();

void sink_add(std::vector<int64_t>& recs, int64_t& counter, int64_t Pp, int64_t dst, int64_t Sn, int64_t rows_, int64_t cols, int64_t ld,
              int64_t pstride, int64_t rkind, int64_t nchunks = 1) {
  const int64_t r[8] = {Pp, dst, Sn, rows_, cols, ld, pstride, rkind | (counter << 8)};
  recs.insert(recs.end(), r, r + 8);
  counter += nchunks * ((rows_ * cols + 127) / 128);
}

// train_ops._sink_record
void sink_record(int64_t Pp, int64_t dst, int64_t Sn, int64_t rows_, int64_t cols, int64_t ld, int64_t pstride, int64_t rkind,
                 const Tensor& keep) {
  TORCH_CHECK(S.sink, "sink_record outside a gradient sink");
  if (S.seen.count(dst)) flush_sink();
  S.seen.insert(dst);
  if (Sn > RED_CHUNK) {
    const int64_t nc = (Sn + RED_CHUNK - 1) / RED_CHUNK;
    sink_add(S.recs, S.blocks, Pp, dst, Sn, rows_, cols, ld, pstride, 64, nc);
    sink_add(S.recs2, S.blocks2, Pp + 4 * Sn * pstride, dst, nc, rows_, cols, ld, pstride, rkind);
  } else {
    sink_add(S.recs, S.blocks, Pp, dst, Sn, rows_, cols, ld, pstride, rkind);
  }
  if (keep.defined()) S.keep.push_back(keep);
}

Tensor to_device_i64(const std::vector<int64_t>& v) {
  Tensor host = at::empty({(int64_t)v.size()}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
  std::memcpy(host.data_ptr(), v.data(), v.size() * sizeof(int64_t));
  Tensor dev = at::empty({(int64_t)v.size()}, S.like.options().dtype(at::kLong));
  dev.copy_(host, /*non_blocking=*/true);
  S.keep.push_back(host);   // (until the next flush: the copy is asynchronous)
  return dev;
}

// train_ops._flush_wgrads
void flush_wgrads() {
  if (!S.sink || S.wq.empty()) return;
  std::vector<Job> jobs;
  jobs.swap(S.wq);
  S.wq_bytes = 0;
  int64_t total = 0;
  for (auto& j : jobs) total += (j.plan[6] + 3) / 4 * 4;
  Tensor part = at::empty({total}, S.like.options().dtype(at::kFloat));
  const int64_t base = A(part);
  struct Placed { int64_t Pp, Sn, N, K, boff, dst_w, ldw, dst_b, rk; };
  std::vector<Placed> placed;
  placed.reserve(jobs.size());
  std::vector<std::array<int64_t, 16>> by_kind[8];
  int64_t off = 0;
  for (auto& j : jobs) {
    const int64_t kind = j.plan[0], gx = j.plan[1], gy = j.plan[2], Sn = j.plan[3], mper = j.plan[4], boff = j.plan[5], psize = j.plan[6],
                  blocks = j.plan[7];
    const int64_t Pp = base + 4 * off;
    by_kind[kind].push_back({A(j.g), A(j.x), Pp, j.dst_b ? Pp + 4 * boff : 0, j.g.stride(0), j.x.stride(0), j.M, j.N, j.K, mper, gx, gy, Sn,
                             j.dt, 0, blocks});
    placed.push_back({Pp, Sn, j.N, j.K, boff, j.dst_w, j.ldw, j.dst_b, j.rk});
    off += (psize + 3) / 4 * 4;
  }
  std::vector<int64_t> rows_;
  struct Launch { int kind; int64_t start, n, tb; };
  std::vector<Launch> launches;
  for (int kind = 0; kind < 8; ++kind) {
    auto& recs = by_kind[kind];
    if (recs.empty()) continue;
    std::stable_sort(recs.begin(), recs.end(), [](const std::array<int64_t, 16>& a, const std::array<int64_t, 16>& b) { return a[9] > b[9]; });
    int64_t fb = 0;
    for (auto& r : recs) {
      r[14] = fb;
      fb += r[15];
    }
    launches.push_back({kind, (int64_t)rows_.size() / 16, (int64_t)recs.size(), fb});
    for (auto& r : recs) rows_.insert(rows_.end(), r.begin(), r.end());
  }
  Tensor desc = to_device_i64(rows_);
  void* st = cur_stream();
  for (auto& l : launches) {
    chk(mdx_op_wgrad_grouped(reinterpret_cast<const int64_t*>(desc.data_ptr()) + 16 * l.start, (int32_t)l.n, l.tb, l.kind, st));
    ++S.n_launch;
  }
  S.keep.push_back(part);
  S.keep.push_back(desc);
  for (auto& p : placed) {
    sink_record(p.Pp, p.dst_w, p.Sn, p.N, p.K, p.ldw, p.N * p.K, p.rk, Tensor());
    if (p.dst_b) sink_record(p.Pp + 4 * p.boff, p.dst_b, p.Sn, 1, p.N, p.N, p.N, p.rk, Tensor());
  }
  S.keep.push_back(part);   // (again: a repeated destination above flushes the sink, which drops its references)
  S.keep.push_back(desc);
  // the operands (jobs) are released here: the launches above are enqueued, the allocator reuses memory in stream order
}

// train_ops.flush_grad_sink
void flush_sink() {
  if (!S.sink) return;
  flush_wgrads();
  if (S.recs.empty()) return;
  void* st = cur_stream();
  std::vector<Tensor> descs;
  if (!S.recs.empty()) {
    Tensor d = to_device_i64(S.recs);
    chk(mdx_op_reduce_deferred(reinterpret_cast<const int64_t*>(d.data_ptr()), (int32_t)(S.recs.size() / 8), S.blocks, st));
    descs.push_back(d);
    ++S.n_launch;
  }
  if (!S.recs2.empty()) {
    Tensor d = to_device_i64(S.recs2);
    chk(mdx_op_reduce_deferred(reinterpret_cast<const int64_t*>(d.data_ptr()), (int32_t)(S.recs2.size() / 8), S.blocks2, st));
    descs.push_back(d);
    ++S.n_launch;
  }
  // the partial buffers may be reused once the launches above are enqueued (stream order).  The pinned staging tensors of THIS flush
  // stay referenced until the next one (their copies are asynchronous).
  std::vector<Tensor> staging;
  for (auto& t : S.keep)
    if (t.defined() && !t.is_cuda()) staging.push_back(t);
  S.keep.clear();
  for (auto& t : staging) S.keep.push_back(t);
  for (auto& t : descs) S.keep.push_back(t);
  S.recs.clear();
  S.recs2.clear();
  S.blocks = S.blocks2 = 0;
  S.seen.clear();
}

// queue one weight gradient dW = g^T x (+ bias column sums): train_ops.sgemm_tn(defer=...) in the float16 mode
void wq_append(const Tensor& g, const Tensor& x, int64_t dst_w, int64_t ldw, int64_t dst_b, int64_t rk) {
  TORCH_CHECK(S.sink && dst_w, "wq_append: no sink destination");
  TORCH_CHECK(g.stride(1) == 1 && x.stride(1) == 1, "wq_append: unit column stride expected");
  Job j;
  j.g = g, j.x = x;
  j.M = g.size(0), j.N = g.size(1), j.K = x.size(1);
  j.dt = H(g) | (H(x) << 1);
  const int aligned = (((uintptr_t)g.data_ptr()) % 16 == 0 && ((uintptr_t)x.data_ptr()) % 16 == 0) ? 1 : 0;
  const int64_t splits = std::max<int64_t>(1, (j.M + S.wgrad_rows - 1) / S.wgrad_rows);
  chk(mdx_op_wgrad_plan(j.M, j.N, j.K, (int32_t)splits, (int32_t)j.dt, g.stride(0), x.stride(0), aligned, j.plan));
  j.dst_w = dst_w, j.ldw = ldw, j.dst_b = dst_b, j.rk = rk;
  S.wq_bytes += g.numel() * g.element_size() + x.numel() * x.element_size();
  S.wq.push_back(std::move(j));
  if (S.wq_bytes > S.wq_cap) flush_wgrads();
}
inline int64_t rk_now() { return S.amp_auto ? S.amp0 : 0; }

// W^T of a parameter matrix or of a column slice of one as (pointer, leading dimension): train_ops.TransposedParams.view; 0 if none
bool wt_view(const Tensor& w, int64_t& ptr, int64_t& ld) {
  if (!S.wt_base || w.dim() != 2 || w.scalar_type() != at::kFloat || w.stride(1) != 1) return false;
  const int64_t rel = A(w) - S.wt_base;
  if (rel < 0 || rel >= S.wt_nbytes) return false;
  auto it = std::upper_bound(S.wt_off.begin(), S.wt_off.end(), rel);
  if (it == S.wt_off.begin()) return false;
  const auto& e = S.wt_ent[(it - S.wt_off.begin()) - 1];
  const int64_t off = e[0], R = e[1], C = e[2], doff = e[3];
  const int64_t col = rel / 4 - off;
  if (col < 0 || col >= C || w.stride(0) != C || w.size(0) != R || col + w.size(1) > C) return false;
  const int64_t p = S.wt_buf + 4 * (doff + col * R);
  if (p % 16) return false;
  ptr = p, ld = R;
  return true;
}

// ---- Linear ------------------------------------------------------------------------------------------------------------------------
// y = x w^T + b + addend in the float16 autocast arithmetic (train_ops._Linear + sgemm_nt, AMP branch)
Tensor xgemm_nt(const Tensor& a, const void* B, int64_t ldb, int64_t N, int64_t K, const Tensor& bias, const Tensor& addend, bool keep32,
                c10::optional<at::ScalarType> out_dtype) {
  const int64_t M = a.size(0);
  const at::ScalarType od = out_dtype ? *out_dtype : (keep32 ? at::kFloat : (S.amp_store ? at::kHalf : at::kFloat));
  Tensor out = at::empty({M, N}, a.options().dtype(od));
  const int rnd = ((S.amp_auto && !keep32) || od == at::kHalf) ? 1 : 0;
  chk(mdx_op_xgemm_nt_t(a.data_ptr(), a.stride(0), reinterpret_cast<const float*>(B), ldb, reinterpret_cast<const float*>(P(bias)), P(addend),
                        addend.defined() ? addend.stride(0) : 0, out.data_ptr(), N, M, N, K, S.amp0, rnd, H(a) | (H(addend) << 1) | (H(out) << 2),
                        cur_stream()));
  ++S.n_launch;
  return out;
}

struct LinCtx {
  Tensor x, w, b;            // normalised operands (b: the parameter view, for its sink address)
  bool has_bias = false, has_addend = false, need_x = false, need_w = false, need_b = false, need_add = false;
  at::ScalarType x_dtype = at::kFloat, add_dtype = at::kFloat;
};

// data gradient, queued weight (+ bias) gradient, addend gradient of one Linear
void linear_backward(const LinCtx& c, const Tensor& gy_in, Tensor& gx, Tensor& ga) {
  Tensor gy = tc(gy_in);
  if (c.need_x) {
    int64_t wtp = 0, wtld = 0;
    const int64_t N = c.w.size(0), K = c.w.size(1);
    if (wt_view(c.w, wtp, wtld)) {
      gx = xgemm_nt(gy, (const void*)wtp, wtld, K, N, Tensor(), Tensor(), false, c.x.scalar_type());
    } else {  // train_ops.transpose: W^T into a padded buffer
      const int64_t ld = (N + 3) / 4 * 4;
      Tensor buf = at::empty({K, ld}, c.w.options());
      chk(mdx_op_transpose(reinterpret_cast<const float*>(c.w.data_ptr()), c.w.stride(0), N, K, reinterpret_cast<float*>(buf.data_ptr()), ld,
                           cur_stream()));
      ++S.n_launch;
      gx = xgemm_nt(gy, buf.data_ptr(), ld, K, N, Tensor(), Tensor(), false, c.x.scalar_type());
    }
    if (gx.scalar_type() != c.x_dtype) gx = gx.to(c.x_dtype);
  }
  if (c.need_w) {
    const int64_t dst_w = sink_dst(c.w), dst_b = (c.has_bias && c.need_b) ? sink_dst(c.b) : 0;
    TORCH_CHECK(dst_w && (!(c.has_bias && c.need_b) || dst_b), "fast Linear: parameter left the gradient sink between forward and backward");
    wq_append(gy, c.x, dst_w, c.w.stride(0), dst_b, rk_now());
  }
  if (c.has_addend && c.need_add) ga = gy.scalar_type() == c.add_dtype ? gy : gy.to(c.add_dtype);
}

// can this Linear take the fast node?  (float16-container mode inside a sink, parameters in the sink, weight + bias trained together)
bool linear_fast_ok(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b) {
  if (!fast_mode() || !S.wq_on || x.dim() != 2 || w.dim() != 2 || !x.is_cuda()) return false;
  if (w.scalar_type() != at::kFloat || w.stride(1) != 1 || !sink_dst(w) || !w.requires_grad()) return false;
  if (b && b->defined() && (!sink_dst(*b) || !b->requires_grad())) return false;
  return true;
}

class LinearFn : public torch::autograd::Function<LinearFn> {
 public:
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b,
                        const c10::optional<Tensor>& addend, bool keep32) {
    Tensor xc = rows(x), wc = rows(w);
    Tensor bc = (b && b->defined()) ? fc(*b) : Tensor();
    Tensor ac = (addend && addend->defined()) ? rows(*addend) : Tensor();
    Tensor out = xgemm_nt(xc, wc.data_ptr(), wc.stride(0), wc.size(0), wc.size(1), bc, ac, keep32, c10::nullopt);
    ctx->save_for_backward({xc, wc, (b && b->defined()) ? *b : Tensor()});
    ctx->saved_data["f"] = (int64_t)((b && b->defined()) | ((addend && addend->defined()) << 1));
    ctx->saved_data["xd"] = (int64_t)x.scalar_type();
    ctx->saved_data["amp"] = amp_pack();
    ctx->saved_data["ad"] = (int64_t)((addend && addend->defined()) ? addend->scalar_type() : at::kFloat);
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    AmpGuard guard(ctx->saved_data["amp"].toInt());
    auto sv = ctx->get_saved_variables();
    LinCtx c;
    c.x = sv[0], c.w = sv[1], c.b = sv[2];
    const int64_t f = ctx->saved_data["f"].toInt();
    c.has_bias = f & 1, c.has_addend = f & 2;
    // (needs_input_grad counts the TENSOR arguments that were present: x, w, [b], [addend], ...)
    c.need_x = ctx->needs_input_grad(0), c.need_w = ctx->needs_input_grad(1);
    c.need_b = c.has_bias && ctx->needs_input_grad(2);
    c.need_add = c.has_addend && ctx->needs_input_grad(2 + (c.has_bias ? 1 : 0));
    c.x_dtype = (at::ScalarType)ctx->saved_data["xd"].toInt();
    c.add_dtype = (at::ScalarType)ctx->saved_data["ad"].toInt();
    Tensor gx, ga;
    linear_backward(c, grads[0], gx, ga);
    return {gx, Tensor(), Tensor(), ga, Tensor()};
  }
};

// ---- Linear + LayerNorm + ReLU (train_ops._LinearLnRelu) ------------------------------------------------------------------------------
void ln_backward(const Tensor& gy_in, const Tensor& x, const Tensor& stats, const Tensor& g, const Tensor& b, const Tensor& gamma_ref,
                 const Tensor& beta_ref, int relu, Tensor& dx) {
  Tensor gy = tc(gy_in);
  const int64_t M = x.size(0), F = x.size(1);
  dx = at::empty_like(x);
  Tensor ws = at::empty({(int64_t)(mdx_op_ln_relu_bwd_ws(M, (int32_t)F) / 4 + 1)}, x.options().dtype(at::kFloat));
  const int64_t dst_g = sink_dst(gamma_ref), dst_b = sink_dst(beta_ref);
  TORCH_CHECK(dst_g && dst_b && M > 0 && ((uintptr_t)ws.data_ptr()) % 16 == 0, "fast LayerNorm backward: parameters not in the gradient sink");
  chk(mdx_op_ln_relu_bwd_t(gy.data_ptr(), x.data_ptr(), reinterpret_cast<const float*>(stats.data_ptr()),
                           reinterpret_cast<const float*>(g.data_ptr()), reinterpret_cast<const float*>(b.data_ptr()), M, (int32_t)F, relu,
                           dx.data_ptr(), nullptr, reinterpret_cast<float*>(ws.data_ptr()), H(gy) | (H(x) << 1) | (H(dx) << 2), cur_stream()));
  ++S.n_launch;
  const int64_t nrows = mdx_op_ln_relu_bwd_rows(M);
  sink_record(A(ws), dst_g, nrows, 1, F, F, 2 * F, 0, ws);
  sink_record(A(ws) + 4 * F, dst_b, nrows, 1, F, F, 2 * F, 0, ws);
}

class LinearLnReluFn : public torch::autograd::Function<LinearLnReluFn> {
 public:
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b,
                        const c10::optional<Tensor>& addend, const Tensor& gamma, const Tensor& beta) {
    Tensor xc = rows(x), wc = rows(w);
    const int64_t M = xc.size(0), K = xc.size(1), N = wc.size(0);
    Tensor bc = (b && b->defined()) ? fc(*b) : Tensor();
    Tensor ac = (addend && addend->defined()) ? rows(*addend) : Tensor();
    Tensor g = fc(gamma), bt = fc(beta);
    Tensor pre = half_empty(M, N, xc), post = half_empty(M, N, xc);
    Tensor stats = at::empty({M, 2}, xc.options().dtype(at::kFloat));
    chk(mdx_op_xgemm_nt_ln_t(xc.data_ptr(), xc.stride(0), reinterpret_cast<const float*>(wc.data_ptr()), wc.stride(0),
                             reinterpret_cast<const float*>(P(bc)), P(ac), ac.defined() ? ac.stride(0) : 0, pre.data_ptr(), N,
                             reinterpret_cast<const float*>(g.data_ptr()), reinterpret_cast<const float*>(bt.data_ptr()), post.data_ptr(), N,
                             reinterpret_cast<float*>(stats.data_ptr()), 1, M, N, K, S.amp0, 1, 1 | (H(ac) << 1) | 4 | 8, cur_stream()));
    ++S.n_launch;
    ctx->save_for_backward({xc, wc, (b && b->defined()) ? *b : Tensor(), pre, g, bt, stats, gamma, beta});
    ctx->saved_data["f"] = (int64_t)((b && b->defined()) | ((addend && addend->defined()) << 1));
    ctx->saved_data["xd"] = (int64_t)x.scalar_type();
    ctx->saved_data["amp"] = amp_pack();
    ctx->saved_data["ad"] = (int64_t)((addend && addend->defined()) ? addend->scalar_type() : at::kFloat);
    return post;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    AmpGuard guard(ctx->saved_data["amp"].toInt());
    auto sv = ctx->get_saved_variables();
    Tensor gpre;
    ln_backward(grads[0], sv[3], sv[6], sv[4], sv[5], sv[7], sv[8], 1, gpre);
    LinCtx c;
    c.x = sv[0], c.w = sv[1], c.b = sv[2];
    const int64_t f = ctx->saved_data["f"].toInt();
    c.has_bias = f & 1, c.has_addend = f & 2;
    // (needs_input_grad counts the TENSOR arguments that were present: x, w, [b], [addend], ...)
    c.need_x = ctx->needs_input_grad(0), c.need_w = ctx->needs_input_grad(1);
    c.need_b = c.has_bias && ctx->needs_input_grad(2);
    c.need_add = c.has_addend && ctx->needs_input_grad(2 + (c.has_bias ? 1 : 0));
    c.x_dtype = (at::ScalarType)ctx->saved_data["xd"].toInt();
    c.add_dtype = (at::ScalarType)ctx->saved_data["ad"].toInt();
    Tensor gx, ga;
    linear_backward(c, gpre, gx, ga);
    return {gx, Tensor(), Tensor(), ga, Tensor(), Tensor()};
  }
};

// ---- Linear + LayerNorm + ReLU + Linear to ONE output (common.MLP of PosUpdate's inter module: 256 -> 256 -> 1) -----------------------------
// forward: the fused Linear+LayerNorm launch, then the row dot as a GEMM with N = 1.  backward: the second Linear's weight gradient is
// queued, its data gradient is the rank-1 product g1[row] * w2[c], formed INSIDE the LayerNorm backward (mdx_op_ln_relu_bwd_r1_t): the
// (E,256) gradient tensor, the K = 1 GEMM and the weight transpose of the per-operator path are gone.
class LinearLnReluDotFn : public torch::autograd::Function<LinearLnReluDotFn> {
 public:
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const Tensor& b, const Tensor& gamma, const Tensor& beta,
                        const Tensor& w2, const Tensor& b2) {
    Tensor xc = rows(x), wc = rows(w), bc = fc(b), g = fc(gamma), bt = fc(beta), w2c = rows(w2), b2c = fc(b2);
    const int64_t M = xc.size(0), K = xc.size(1), N = wc.size(0);
    TORCH_CHECK(w2c.size(0) == 1 && w2c.size(1) == N, "linear_ln_relu_dot: second weight must be (1, N)");
    Tensor pre = half_empty(M, N, xc), post = half_empty(M, N, xc);
    Tensor stats = at::empty({M, 2}, xc.options().dtype(at::kFloat));
    chk(mdx_op_xgemm_nt_ln_t(xc.data_ptr(), xc.stride(0), reinterpret_cast<const float*>(wc.data_ptr()), wc.stride(0),
                             reinterpret_cast<const float*>(bc.data_ptr()), nullptr, 0, pre.data_ptr(), N,
                             reinterpret_cast<const float*>(g.data_ptr()), reinterpret_cast<const float*>(bt.data_ptr()), post.data_ptr(), N,
                             reinterpret_cast<float*>(stats.data_ptr()), 1, M, N, K, S.amp0, 1, 1 | 4 | 8, cur_stream()));
    ++S.n_launch;
    Tensor y = xgemm_nt(post, w2c.data_ptr(), w2c.stride(0), 1, N, b2c, Tensor(), false, c10::nullopt);
    ctx->save_for_backward({xc, wc, b, pre, g, bt, stats, gamma, beta, post, w2c, b2, w2});
    ctx->saved_data["xd"] = (int64_t)x.scalar_type();
    ctx->saved_data["amp"] = amp_pack();
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    AmpGuard guard(ctx->saved_data["amp"].toInt());
    auto sv = ctx->get_saved_variables();
    const Tensor &pre = sv[3], &g = sv[4], &bt = sv[5], &stats = sv[6], &post = sv[9], &w2c = sv[10];
    Tensor gy = tc(grads[0]);                                  // (M,1)
    const int64_t M = pre.size(0), F = pre.size(1);
    // second Linear: weight (+ bias) gradient queued
    const int64_t dst_w2 = sink_dst(sv[12]), dst_b2 = sink_dst(sv[11]);
    TORCH_CHECK(dst_w2 && dst_b2, "linear_ln_relu_dot: parameters left the gradient sink");
    wq_append(gy, post, dst_w2, sv[12].stride(0), dst_b2, rk_now());
    // LayerNorm backward with the rank-1 upstream gradient
    Tensor gpre = at::empty_like(pre);
    Tensor ws = at::empty({(int64_t)(mdx_op_ln_relu_bwd_ws(M, (int32_t)F) / 4 + 1)}, pre.options().dtype(at::kFloat));
    const int64_t dst_g = sink_dst(sv[7]), dst_bt = sink_dst(sv[8]);
    TORCH_CHECK(dst_g && dst_bt && ((uintptr_t)ws.data_ptr()) % 16 == 0, "linear_ln_relu_dot: LayerNorm parameters not in the gradient sink");
    chk(mdx_op_ln_relu_bwd_r1_t(gy.data_ptr(), reinterpret_cast<const float*>(w2c.data_ptr()), pre.data_ptr(),
                                reinterpret_cast<const float*>(stats.data_ptr()), reinterpret_cast<const float*>(g.data_ptr()),
                                reinterpret_cast<const float*>(bt.data_ptr()), M, (int32_t)F, 1, gpre.data_ptr(),
                                reinterpret_cast<float*>(ws.data_ptr()), H(gy) | (H(pre) << 1) | (H(gpre) << 2), cur_stream()));
    ++S.n_launch;
    const int64_t nrows = mdx_op_ln_relu_bwd_rows(M);
    sink_record(A(ws), dst_g, nrows, 1, F, F, 2 * F, 0, ws);
    sink_record(A(ws) + 4 * F, dst_bt, nrows, 1, F, F, 2 * F, 0, ws);
    // first Linear
    LinCtx c;
    c.x = sv[0], c.w = sv[1], c.b = sv[2];
    c.has_bias = true, c.has_addend = false;
    c.need_x = ctx->needs_input_grad(0), c.need_w = ctx->needs_input_grad(1), c.need_b = ctx->needs_input_grad(2);
    c.x_dtype = (at::ScalarType)ctx->saved_data["xd"].toInt();
    Tensor gx, ga;
    linear_backward(c, gpre, gx, ga);
    return {gx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};
bool linear_ln_dot_fast_ok(const Tensor& x, const Tensor& w, const Tensor& b, const Tensor& gamma, const Tensor& beta, const Tensor& w2,
                           const Tensor& b2) {
  if (!linear_fast_ok(x, w, b)) return false;
  for (const Tensor* p : {&gamma, &beta, &w2, &b2})
    if (!sink_dst(*p) || !p->requires_grad()) return false;
  const int64_t F = w.size(0);
  return w2.dim() == 2 && w2.size(0) == 1 && w2.size(1) == F && w2.stride(1) == 1 && (F == 32 || F == 64 || F == 128 || F == 256) &&
         x.scalar_type() == at::kHalf;
}

// ---- element-wise (train_ops._Ew): add 0, sub 1, mul 2, gate 3 ------------------------------------------------------------------------
class EwFn : public torch::autograd::Function<EwFn> {
 public:
  static Tensor forward(AutogradContext* ctx, int64_t op, const Tensor& a, const Tensor& b) {
    Tensor ac = tc(a), bc = tc(b);
    if (ac.scalar_type() != bc.scalar_type()) ac = ac.to(at::kFloat), bc = bc.to(at::kFloat);
    TORCH_CHECK(ac.sizes() == bc.sizes(), "element-wise operands differ in shape");
    Tensor out = at::empty_like(ac);
    const int64_t rk = (op == 2 || op == 3) ? rk_now() : 0;
    chk(mdx_op_ew_fwd_t((int32_t)(op | (rk << 8)), ac.data_ptr(), bc.data_ptr(), out.data_ptr(), ac.numel(), H(ac) | (H(bc) << 1) | (H(out) << 2),
                        cur_stream()));
    ++S.n_launch;
    ctx->save_for_backward({ac, bc});
    ctx->saved_data["op"] = op;
    ctx->saved_data["amp"] = amp_pack();
    ctx->saved_data["da"] = (int64_t)a.scalar_type();
    ctx->saved_data["db"] = (int64_t)b.scalar_type();
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    AmpGuard guard(ctx->saved_data["amp"].toInt());
    auto sv = ctx->get_saved_variables();
    const Tensor &a = sv[0], &b = sv[1];
    Tensor g = tc(grads[0]);
    Tensor da = ctx->needs_input_grad(0) ? at::empty_like(a) : Tensor();     // (indices over the tensor arguments: a, b)
    Tensor db = ctx->needs_input_grad(1) ? at::empty_like(b) : Tensor();
    chk(mdx_op_ew_bwd_t((int32_t)ctx->saved_data["op"].toInt(), a.data_ptr(), b.data_ptr(), g.data_ptr(), P(da), P(db), a.numel(),
                        H(a) | (H(b) << 1) | (H(g) << 2) | (H(da) << 3) | (H(db) << 4), cur_stream()));
    ++S.n_launch;
    const auto ta = (at::ScalarType)ctx->saved_data["da"].toInt(), tb = (at::ScalarType)ctx->saved_data["db"].toInt();
    if (da.defined() && da.scalar_type() != ta) da = da.to(ta);
    if (db.defined() && db.scalar_type() != tb) db = db.to(tb);
    return {Tensor(), da, db};
  }
};

// ---- segment sum / gather bodies (train_ops._segsum_raw / _gather_raw) ------------------------------------------------------------------
Tensor segsum_raw(const Tensor& src, const Tensor& order, const Tensor& ptr, int64_t n, bool out_half) {
  const int64_t F = src.size(1);
  Tensor out = at::empty({n, F}, src.options().dtype(out_half ? at::kHalf : at::kFloat));
  chk(mdx_op_segsum_rows_t(src.data_ptr(), reinterpret_cast<const int64_t*>(order.data_ptr()), reinterpret_cast<const int64_t*>(ptr.data_ptr()), n,
                           (int32_t)F, out.data_ptr(), H(src) | (H(out) << 1), cur_stream()));
  ++S.n_launch;
  return out;
}

// LayerNorm-parameter partial rows of a fused backward -> sink records
void lnp_records(const Tensor& lnp, int64_t nwg, int64_t lnf, const std::vector<std::pair<const Tensor*, std::pair<int64_t, int64_t>>>& items) {
  for (auto& it : items) {
    const int64_t dst = sink_dst(*it.first);
    TORCH_CHECK(dst, "fused backward: LayerNorm parameter not in the gradient sink");
    sink_record(A(lnp) + 4 * it.second.first, dst, nwg, 1, it.second.second, it.second.second, lnf, 0, lnp);
  }
}
inline void wg(const Tensor& gy, const Tensor& xin, const Tensor& w, const Tensor* b) {
  const int64_t dst_w = sink_dst(w), dst_b = b ? sink_dst(*b) : 0;
  TORCH_CHECK(dst_w && (!b || dst_b), "fused backward: weight not in the gradient sink");
  wq_append(gy, xin, dst_w, w.stride(0), dst_b, rk_now());
}
bool all_in_sink(const std::vector<Tensor>& ps) {
  if (!fast_mode() || !S.wq_on) return false;
  for (auto& p : ps)
    if (!sink_dst(p) || !p.requires_grad()) return false;
  return true;
}
inline const float* F32(const Tensor& t) { return reinterpret_cast<const float*>(t.data_ptr()); }

// ---- fused BondFFN + scatter_sum (train_ops._BondFfnScatter) -----------------------------------------------------------------------------
// params: Wb, Wi1, bi1, g1, be1, Wi2, bi2, Wg1, bg1, gg, gbe, Wt, Wg2, bg2
enum { B_Wb, B_Wi1, B_bi1, B_g1, B_be1, B_Wi2, B_bi2, B_Wg1, B_bg1, B_gg, B_gbe, B_Wt, B_Wg2, B_bg2 };
void bondffn_fill(mdx_bondffn_args& a, const Tensor& x, const Tensor& NL, const Tensor& GN, const Tensor& te, const Tensor& idx,
                  const std::vector<Tensor>& p, const std::vector<Tensor>& bufs) {
  a.X = x.data_ptr(), a.ldx = x.stride(0);
  a.Wb = F32(p[B_Wb]), a.ldwb = p[B_Wb].stride(0);
  a.Wi1 = F32(p[B_Wi1]), a.ldwi1 = p[B_Wi1].stride(0), a.bi1 = F32(p[B_bi1]), a.g1 = F32(p[B_g1]), a.be1 = F32(p[B_be1]);
  a.Wi2 = F32(p[B_Wi2]), a.ldwi2 = p[B_Wi2].stride(0), a.bi2 = F32(p[B_bi2]);
  a.Wg1 = F32(p[B_Wg1]), a.ldwg1 = p[B_Wg1].stride(0), a.bg1 = F32(p[B_bg1]), a.gg = F32(p[B_gg]), a.gbe = F32(p[B_gbe]);
  a.Wt = F32(p[B_Wt]), a.ldwt = p[B_Wt].stride(0);
  a.Wg2 = F32(p[B_Wg2]), a.ldwg2 = p[B_Wg2].stride(0), a.bg2 = F32(p[B_bg2]);
  a.NL = NL.data_ptr(), a.ldnl = NL.stride(0), a.GN = F32(GN), a.ldgn = GN.stride(0);
  a.idx = reinterpret_cast<const int64_t*>(idx.data_ptr()), a.te = F32(te);
  a.prod = bufs[0].data_ptr(), a.pre1 = bufs[1].data_ptr(), a.post1 = bufs[2].data_ptr(), a.inter = bufs[3].data_ptr();
  a.gpre = bufs[4].data_ptr(), a.gpost = bufs[5].data_ptr(), a.gate = bufs[6].data_ptr(), a.out = bufs[7].data_ptr();
  a.E = x.size(0);
}
void check_params(const std::vector<Tensor>& p) {
  for (auto& t : p) TORCH_CHECK(t.scalar_type() == at::kFloat && t.stride(-1) == 1, "fused operator: fp32 parameters with unit column stride expected");
}
// -> [S (n_out,64) fp32, x, NL, GN, te, prod, pre1, post1, inter, gpre, gpost, gate, out]
std::vector<Tensor> bondffn_fwd(const Tensor& bond_in, const Tensor& NL, const Tensor& GN, const Tensor& time, const Tensor& idx,
                                const Tensor& o_order, const Tensor& o_ptr, int64_t n_out, const std::vector<Tensor>& params) {
  check_params(params);
  Tensor x = aligned_rows(bond_in, 8, 16), NLc = rows(NL), GNc = rows(GN), te = fc(time).reshape({-1});
  const int64_t E = x.size(0);
  std::vector<Tensor> bufs = {half_empty(E, 128, x), half_empty(E, 128, x), half_empty(E, 128, x), half_empty(E, 64, x),
                              half_empty(E, 32, x),  half_empty(E, 32, x),  half_empty(E, 64, x),  half_empty(E, 64, x)};
  mdx_bondffn_args a;
  bondffn_fill(a, x, NLc, GNc, te, idx, params, bufs);
  chk(mdx_op_bondffn_fwd(&a, cur_stream()));
  ++S.n_launch;
  std::vector<Tensor> r = {segsum_raw(bufs[7], o_order, o_ptr, n_out, false), x, NLc, GNc, te};
  r.insert(r.end(), bufs.begin(), bufs.end());
  return r;
}
// saved = what bondffn_fwd returned after S; -> [g_x, g_NL, g_GN]
std::vector<Tensor> bondffn_bwd(const Tensor& gS_in, const std::vector<Tensor>& saved, const Tensor& idx, const Tensor& i_order,
                                const Tensor& i_ptr, int64_t n_in, const Tensor& oidx, const Tensor& time2d, const std::vector<Tensor>& params,
                                bool need_x, bool need_nl, bool need_gn) {
  const Tensor &x = saved[0], &NL = saved[1], &GN = saved[2], &te = saved[3];
  std::vector<Tensor> bufs(saved.begin() + 4, saved.begin() + 12);
  const int64_t E = x.size(0);
  Tensor gS = fc(gS_in);
  Tensor g_inter = half_empty(E, 64, x), g_gate = half_empty(E, 64, x), g_pre1 = half_empty(E, 128, x), g_bf = half_empty(E, 128, x),
         g_nl = half_empty(E, 128, x), g_gpre = half_empty(E, 32, x), g_x = half_empty(E, 64, x);
  const int64_t nwg = mdx_op_bondffn_workgroups(), lnf = mdx_op_bondffn_lnp_floats();
  Tensor lnp = at::empty({nwg, lnf}, x.options().dtype(at::kFloat));
  mdx_bondffn_bwd_args b;
  bondffn_fill(b.f, x, NL, GN, te, idx, params, bufs);
  b.gS = F32(gS), b.ldgs = gS.stride(0), b.oidx = reinterpret_cast<const int64_t*>(oidx.data_ptr());
  b.g_inter = g_inter.data_ptr(), b.g_gate = g_gate.data_ptr(), b.g_pre1 = g_pre1.data_ptr(), b.g_bf = g_bf.data_ptr();
  b.g_nl = g_nl.data_ptr(), b.g_gpre = g_gpre.data_ptr(), b.g_x = g_x.data_ptr(), b.lnp = reinterpret_cast<float*>(lnp.data_ptr());
  chk(mdx_op_bondffn_bwd(&b, cur_stream()));
  ++S.n_launch;
  const auto& p = params;
  wg(g_inter, bufs[2], p[B_Wi2], &p[B_bi2]);
  wg(g_gate, bufs[5], p[B_Wg2], &p[B_bg2]);
  wg(g_pre1, bufs[0], p[B_Wi1], &p[B_bi1]);
  wg(g_bf, x, p[B_Wb], nullptr);
  wg(g_gpre, x, p[B_Wg1], &p[B_bg1]);
  wg(g_gpre, time2d, p[B_Wt], nullptr);
  lnp_records(lnp, nwg, lnf, {{&p[B_g1], {0, 128}}, {&p[B_be1], {128, 128}}, {&p[B_gg], {256, 32}}, {&p[B_gbe], {288, 32}}});
  return {need_x ? g_x : Tensor(), need_nl ? segsum_raw(g_nl, i_order, i_ptr, n_in, true) : Tensor(),
          need_gn ? segsum_raw(g_gpre, i_order, i_ptr, n_in, false) : Tensor()};
}

// ---- fused EdgeBlock tail (train_ops._EdgeTail) ---------------------------------------------------------------------------------------------
// params: Ws, bs, lng, lnb, Wo, bo
void edge_tail_fill(mdx_edge_tail_args& a, const Tensor& x, const Tensor& BL, const Tensor& BR, const Tensor& il, const Tensor& ir,
                    const std::vector<Tensor>& p, const Tensor& pre, const Tensor& post, const Tensor& out) {
  a.H = x.data_ptr(), a.ldh = x.stride(0), a.BL = BL.data_ptr(), a.ldbl = BL.stride(0), a.BR = BR.data_ptr(), a.ldbr = BR.stride(0);
  a.il = reinterpret_cast<const int64_t*>(il.data_ptr()), a.ir = reinterpret_cast<const int64_t*>(ir.data_ptr());
  a.Ws = F32(p[0]), a.ldws = p[0].stride(0), a.bs = F32(p[1]), a.lng = F32(p[2]), a.lnb = F32(p[3]);
  a.Wo = F32(p[4]), a.ldwo = p[4].stride(0), a.bo = F32(p[5]);
  a.pre = pre.data_ptr(), a.post = post.data_ptr(), a.out = out.defined() ? out.data_ptr() : nullptr;
  a.E = x.size(0);
}
// -> [out, x, BL, BR, pre, post]
std::vector<Tensor> edge_tail_fwd(const Tensor& h, const Tensor& BL, const Tensor& BR, const Tensor& il, const Tensor& ir,
                                  const std::vector<Tensor>& params) {
  check_params(params);
  Tensor x = aligned_rows(h, 8, 16), BLc = rows(BL), BRc = rows(BR);
  const int64_t E = x.size(0);
  Tensor pre = half_empty(E, 64, x), post = half_empty(E, 64, x), out = half_empty(E, 64, x);
  mdx_edge_tail_args a;
  edge_tail_fill(a, x, BLc, BRc, il, ir, params, pre, post, out);
  chk(mdx_op_edge_tail_fwd(&a, cur_stream()));
  ++S.n_launch;
  return {out, x, BLc, BRc, pre, post};
}
// -> [g_h, g_BL, g_BR]
std::vector<Tensor> edge_tail_bwd(const Tensor& g_out_in, const std::vector<Tensor>& saved, const Tensor& il, const Tensor& ir,
                                  const Tensor& l_order, const Tensor& l_ptr, const Tensor& r_order, const Tensor& r_ptr, int64_t n,
                                  const std::vector<Tensor>& params, bool need_h, bool need_bl, bool need_br) {
  const Tensor &x = saved[0], &BL = saved[1], &BR = saved[2], &pre = saved[3], &post = saved[4];
  const int64_t E = x.size(0);
  Tensor g_out = rows(g_out_in);
  if (g_out.scalar_type() != at::kHalf) g_out = g_out.to(at::kHalf);
  if (g_out.stride(0) % 4 || ((uintptr_t)g_out.data_ptr()) % 8) g_out = g_out.contiguous();
  Tensor g_pre = half_empty(E, 64, x), g_h = half_empty(E, 64, x);
  const int64_t nwg = mdx_op_bondffn_workgroups(), lnf = mdx_op_edge_tail_lnp_floats();
  Tensor lnp = at::empty({nwg, lnf}, x.options().dtype(at::kFloat));
  mdx_edge_tail_bwd_args b;
  edge_tail_fill(b.f, x, BL, BR, il, ir, params, pre, post, Tensor());
  b.g_out = g_out.data_ptr(), b.ldg = g_out.stride(0), b.g_pre = g_pre.data_ptr(), b.g_h = g_h.data_ptr();
  b.lnp = reinterpret_cast<float*>(lnp.data_ptr());
  chk(mdx_op_edge_tail_bwd(&b, cur_stream()));
  ++S.n_launch;
  wg(g_out, post, params[4], &params[5]);
  wg(g_pre, x, params[0], &params[1]);
  lnp_records(lnp, nwg, lnf, {{&params[2], {0, 64}}, {&params[3], {64, 64}}});
  return {need_h ? g_h : Tensor(), need_bl ? segsum_raw(g_pre, l_order, l_ptr, n, true) : Tensor(),
          need_br ? segsum_raw(g_pre, r_order, r_ptr, n, true) : Tensor()};
}

// ---- fused PosUpdate front (train_ops._PosFfnFront) -----------------------------------------------------------------------------------------
// params: Wb, Wn, Wg1x, Wg1a, Wt, bg1, gg, gbe, Wg2, bg2
void posffn_fill(mdx_posffn_args& a, const Tensor& x, const Tensor& LF, const Tensor& RF, const Tensor& te, const Tensor& il, const Tensor& ir,
                 const std::vector<Tensor>& p, const std::vector<Tensor>& bufs) {
  a.X = x.data_ptr(), a.ldx = x.stride(0), a.LF = LF.data_ptr(), a.ldlf = LF.stride(0), a.RF = RF.data_ptr(), a.ldrf = RF.stride(0);
  a.il = reinterpret_cast<const int64_t*>(il.data_ptr()), a.ir = reinterpret_cast<const int64_t*>(ir.data_ptr()), a.te = F32(te);
  a.Wb = F32(p[0]), a.ldwb = p[0].stride(0), a.Wn = F32(p[1]), a.ldwn = p[1].stride(0);
  a.Wg1x = F32(p[2]), a.ldwg1x = p[2].stride(0), a.Wg1a = F32(p[3]), a.ldwg1a = p[3].stride(0), a.Wt = F32(p[4]), a.ldwt = p[4].stride(0);
  a.bg1 = F32(p[5]), a.gg = F32(p[6]), a.gbe = F32(p[7]), a.Wg2 = F32(p[8]), a.bg2 = F32(p[9]);
  a.a = bufs[0].data_ptr(), a.prod = bufs[1].data_ptr(), a.gpre = bufs[2].data_ptr(), a.gpost = bufs[3].data_ptr(), a.gate = bufs[4].data_ptr();
  a.E = x.size(0);
}
// -> [prod, gate, x, LF, RF, te, a, gpre, gpost]
std::vector<Tensor> posffn_fwd(const Tensor& h_edge, const Tensor& LF, const Tensor& RF, const Tensor& time, const Tensor& il, const Tensor& ir,
                               const std::vector<Tensor>& params) {
  check_params(params);
  TORCH_CHECK(params[8].is_contiguous() && params[9].is_contiguous(), "posffn: gate.net.3 parameters must be contiguous");
  Tensor x = aligned_rows(h_edge, 8, 16), LFc = aligned_rows(LF, 8, 16), RFc = aligned_rows(RF, 8, 16), te = fc(time).reshape({-1});
  const int64_t E = x.size(0);
  std::vector<Tensor> bufs = {half_empty(E, 64, x), half_empty(E, 256, x), half_empty(E, 32, x), half_empty(E, 32, x), half_empty(E, 1, x)};
  mdx_posffn_args a;
  posffn_fill(a, x, LFc, RFc, te, il, ir, params, bufs);
  chk(mdx_op_posffn_fwd(&a, cur_stream()));
  ++S.n_launch;
  return {bufs[1], bufs[4], x, LFc, RFc, te, bufs[0], bufs[2], bufs[3]};
}
// saved = [x, LF, RF, te, a, gpre, gpost, prod, gate]; -> [g_x, g_LF, g_RF]
std::vector<Tensor> posffn_bwd(const c10::optional<Tensor>& g_prod_in, const c10::optional<Tensor>& g_gate_in, const std::vector<Tensor>& saved,
                               const Tensor& il, const Tensor& ir, const Tensor& l_order, const Tensor& l_ptr, const Tensor& r_order,
                               const Tensor& r_ptr, int64_t n, const Tensor& time2d, const std::vector<Tensor>& params, bool need_x, bool need_lf,
                               bool need_rf) {
  const Tensor &x = saved[0], &LF = saved[1], &RF = saved[2], &te = saved[3];
  std::vector<Tensor> bufs = {saved[4], saved[7], saved[5], saved[6], saved[8]};
  const int64_t E = x.size(0);
  auto f16 = [&](const c10::optional<Tensor>& t, int64_t f) {
    if (!t || !t->defined()) return at::zeros({E, f}, x.options().dtype(at::kHalf));
    Tensor r = t->scalar_type() == at::kHalf ? *t : t->to(at::kHalf);
    return r.contiguous();
  };
  Tensor g_prod = f16(g_prod_in, 256), g_gate = f16(g_gate_in, 1);
  Tensor g_bf = half_empty(E, 256, x), g_nf = half_empty(E, 256, x), g_gpre = half_empty(E, 32, x), g_x = half_empty(E, 64, x),
         g_lf = half_empty(E, 64, x), g_rf = half_empty(E, 64, x);
  const int64_t nwg = mdx_op_bondffn_workgroups(), lnf = mdx_op_posffn_lnp_floats();
  Tensor lnp = at::empty({nwg, lnf}, x.options().dtype(at::kFloat));
  mdx_posffn_bwd_args b;
  posffn_fill(b.f, x, LF, RF, te, il, ir, params, bufs);
  b.g_prod = g_prod.data_ptr(), b.ldgp = g_prod.stride(0), b.g_gate = g_gate.data_ptr(), b.lnp = reinterpret_cast<float*>(lnp.data_ptr());
  b.g_bf = g_bf.data_ptr(), b.g_nf = g_nf.data_ptr(), b.g_gpre = g_gpre.data_ptr(), b.g_x = g_x.data_ptr(), b.g_lf = g_lf.data_ptr(),
  b.g_rf = g_rf.data_ptr();
  chk(mdx_op_posffn_bwd(&b, cur_stream()));
  ++S.n_launch;
  const auto& p = params;
  wg(g_bf, x, p[0], nullptr);
  wg(g_nf, bufs[0], p[1], nullptr);
  wg(g_gpre, x, p[2], &p[5]);
  wg(g_gpre, bufs[0], p[3], nullptr);
  wg(g_gpre, time2d, p[4], nullptr);
  wg(g_gate, bufs[3], p[8], &p[9]);
  lnp_records(lnp, nwg, lnf, {{&p[6], {0, 32}}, {&p[7], {32, 32}}});
  return {need_x ? g_x : Tensor(), need_lf ? segsum_raw(g_lf, l_order, l_ptr, n, true) : Tensor(),
          need_rf ? segsum_raw(g_rf, r_order, r_ptr, n, true) : Tensor()};
}

// ---- fused NodeBlock message path (train_ops._NodeMsg) ----------------------------------------------------------------------------------------
// params: W1e, b1e, lng_e, lnb_e, W2e, b2e, Wm, bm, Wg1, bg1, lng_g, lnb_g, Wg2, bg2
enum { N_W1e, N_b1e, N_lng_e, N_lnb_e, N_W2e, N_b2e, N_Wm, N_bm, N_Wg1, N_bg1, N_lng_g, N_lnb_g, N_Wg2, N_bg2 };
struct PackSpec { int w, n_out, n_in, perm, trans; };
constexpr PackSpec PACKS[10] = {{N_W1e, 256, 64, 0, 0},  {N_W2e, 256, 256, 1, 0}, {N_Wm, 256, 256, 1, 0},  {N_Wg1, 256, 64, 0, 0},
                                {N_Wg2, 256, 256, 1, 0}, {N_Wg2, 256, 256, 1, 1}, {N_Wg1, 64, 256, 1, 1},  {N_Wm, 256, 256, 1, 1},
                                {N_W2e, 256, 256, 1, 1}, {N_W1e, 64, 256, 1, 1}};
void nodemsg_fill(mdx_nodemsg_args& a, const Tensor& x, const Tensor& HN, const Tensor& PN, const Tensor& col, const std::vector<Tensor>& p,
                  const int64_t* packs, const std::vector<Tensor>& bufs) {
  a.X = x.data_ptr(), a.ldx = x.stride(0), a.HN = HN.data_ptr(), a.ldhn = HN.stride(0), a.PN = F32(PN), a.ldpn = PN.stride(0);
  a.col = reinterpret_cast<const int64_t*>(col.data_ptr());
  a.pk_w1e = (const void*)packs[0], a.pk_w2e = (const void*)packs[1], a.pk_wm = (const void*)packs[2], a.pk_wg1 = (const void*)packs[3],
  a.pk_wg2 = (const void*)packs[4];
  a.b1e = F32(p[N_b1e]), a.lng_e = F32(p[N_lng_e]), a.lnb_e = F32(p[N_lnb_e]), a.b2e = F32(p[N_b2e]), a.bm = F32(p[N_bm]), a.bg1 = F32(p[N_bg1]);
  a.lng_g = F32(p[N_lng_g]), a.lnb_g = F32(p[N_lnb_g]), a.bg2 = F32(p[N_bg2]);
  a.he_pre = bufs[0].data_ptr(), a.he_post = bufs[1].data_ptr(), a.he = bufs[2].data_ptr(), a.p = bufs[3].data_ptr(), a.m0 = bufs[4].data_ptr();
  a.g_pre = bufs[5].data_ptr(), a.g_post = bufs[6].data_ptr(), a.gt = bufs[7].data_ptr(), a.msg = bufs[8].data_ptr();
  a.E = x.size(0);
}
void pack_ptrs(const Tensor& buf, int64_t* packs) {
  int64_t off = 0;
  for (int i = 0; i < 10; ++i) {
    packs[i] = A(buf) + 2 * off;
    off += (int64_t)PACKS[i].n_out * PACKS[i].n_in;
  }
}
// -> [out (n,256) fp32, x, HN, PN, packbuf, he_pre, he_post, he, p, m0, g_pre, g_post, gt]
std::vector<Tensor> nodemsg_fwd(const Tensor& edge_attr, const Tensor& HN, const Tensor& PN, const Tensor& col, const Tensor& r_order,
                                const Tensor& r_ptr, int64_t n, const std::vector<Tensor>& params) {
  check_params(params);
  Tensor x = aligned_rows(edge_attr, 8, 16), HNc = rows(HN), PNc = rows(PN);
  const int64_t E = x.size(0);
  int64_t total = 0;
  for (auto& s : PACKS) total += (int64_t)s.n_out * s.n_in;
  Tensor buf = at::empty({total}, x.options().dtype(at::kHalf));
  int64_t packs[10];
  pack_ptrs(buf, packs);
  mdx_pack_jobs jobs;
  for (int i = 0; i < 10; ++i) {
    const Tensor& w = params[PACKS[i].w];
    jobs.job[i] = {F32(w), w.stride(0), PACKS[i].n_out, PACKS[i].n_in, PACKS[i].perm, PACKS[i].trans, (void*)packs[i]};
  }
  jobs.n = 10;
  chk(mdx_op_pack_a(&jobs, cur_stream()));
  ++S.n_launch;
  std::vector<Tensor> bufs(9);
  for (auto& b : bufs) b = half_empty(E, 256, x);
  mdx_nodemsg_args a;
  nodemsg_fill(a, x, HNc, PNc, col, params, packs, bufs);
  chk(mdx_op_nodemsg_fwd(&a, cur_stream()));
  ++S.n_launch;
  std::vector<Tensor> r = {segsum_raw(bufs[8], r_order, r_ptr, n, false), x, HNc, PNc, buf};
  r.insert(r.end(), bufs.begin(), bufs.begin() + 8);   // (msg is not kept)
  return r;
}
// saved = [x, HN, PN, packbuf, he_pre, he_post, he, p, m0, g_pre, g_post, gt]; -> [g_x, g_HN, g_PN]
std::vector<Tensor> nodemsg_bwd(const Tensor& gA_in, const std::vector<Tensor>& saved, const Tensor& col, const Tensor& c_order, const Tensor& c_ptr,
                                int64_t n, const Tensor& row, const std::vector<Tensor>& params, bool need_x, bool need_hn, bool need_pn) {
  const Tensor &x = saved[0], &HN = saved[1], &PN = saved[2], &buf = saved[3];
  std::vector<Tensor> bufs(saved.begin() + 4, saved.begin() + 12);
  bufs.push_back(saved[8]);   // (the forward's msg buffer is gone; the backward does not read it)
  const int64_t E = x.size(0);
  Tensor gA = fc(gA_in);
  Tensor g_m0 = half_empty(E, 256, x), g_gt = half_empty(E, 256, x), g_gpre = half_empty(E, 256, x), g_hne = half_empty(E, 256, x),
         g_he = half_empty(E, 256, x), g_pre = half_empty(E, 256, x), g_x = half_empty(E, 64, x);
  const int64_t nwg = mdx_op_bondffn_workgroups(), lnf = mdx_op_nodemsg_lnp_floats();
  Tensor lnp = at::empty({nwg, lnf}, x.options().dtype(at::kFloat));
  int64_t packs[10];
  pack_ptrs(buf, packs);
  mdx_nodemsg_bwd_args b;
  nodemsg_fill(b.f, x, HN, PN, col, params, packs, bufs);
  b.gA = F32(gA), b.ldga = gA.stride(0), b.row = reinterpret_cast<const int64_t*>(row.data_ptr());
  b.pk_wg2t = (const void*)packs[5], b.pk_wg1t = (const void*)packs[6], b.pk_wmt = (const void*)packs[7], b.pk_w2et = (const void*)packs[8],
  b.pk_w1et = (const void*)packs[9];
  b.g_m0 = g_m0.data_ptr(), b.g_gt = g_gt.data_ptr(), b.g_gpre = g_gpre.data_ptr(), b.g_hne = g_hne.data_ptr(), b.g_he = g_he.data_ptr(),
  b.g_pre = g_pre.data_ptr(), b.g_x = g_x.data_ptr(), b.lnp = reinterpret_cast<float*>(lnp.data_ptr());
  chk(mdx_op_nodemsg_bwd(&b, cur_stream()));
  ++S.n_launch;
  const auto& p = params;
  wg(g_m0, bufs[3], p[N_Wm], &p[N_bm]);
  wg(g_gt, bufs[6], p[N_Wg2], &p[N_bg2]);
  wg(g_gpre, x, p[N_Wg1], &p[N_bg1]);
  wg(g_he, bufs[1], p[N_W2e], &p[N_b2e]);
  wg(g_pre, x, p[N_W1e], &p[N_b1e]);
  lnp_records(lnp, nwg, lnf, {{&p[N_lng_e], {0, 256}}, {&p[N_lnb_e], {256, 256}}, {&p[N_lng_g], {512, 256}}, {&p[N_lnb_g], {768, 256}}});
  return {need_x ? g_x : Tensor(), need_hn ? segsum_raw(g_hne, c_order, c_ptr, n, true) : Tensor(),
          need_pn ? segsum_raw(g_gpre, c_order, c_ptr, n, false) : Tensor()};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "moldiff_amd training fast path: operator bodies, gradient sink and weight-gradient queue in C++ (csrc/mdx_fast.cpp)";
  m.def("set_precision", [](int a0, int autoc, int store) { S.amp0 = a0, S.amp_auto = autoc, S.amp_store = store; });
  m.def("set_options", [](bool wq_on, int64_t rows_, int64_t cap_bytes) { S.wq_on = wq_on, S.wgrad_rows = rows_, S.wq_cap = cap_bytes; });
  m.def("sink_begin", [](const Tensor& data, const Tensor& grad) {
    TORCH_CHECK(!S.sink, "gradient sinks do not nest in the fast path");
    S.sink = true, S.s_data = A(data), S.s_grad = A(grad), S.s_nbytes = data.numel() * 4, S.like = data;
    S.recs.clear(), S.recs2.clear(), S.keep.clear(), S.seen.clear(), S.wq.clear();
    S.blocks = S.blocks2 = S.wq_bytes = 0;
  });
  m.def("sink_end", []() {
    if (S.sink && (!S.recs.empty() || !S.wq.empty())) flush_sink();
    S.sink = false;
    S.wq.clear(), S.keep.clear(), S.like = Tensor();
  });
  m.def("sink_active", []() { return S.sink; });
  m.def("fast_mode", &fast_mode);
  m.def("sink_dst", &sink_dst);
  m.def("sink_record", [](int64_t Pp, int64_t dst, int64_t Sn, int64_t rows_, int64_t cols, int64_t ld, int64_t pstride, int64_t rkind,
                          const c10::optional<Tensor>& keep) {
    sink_record(Pp, dst, Sn, rows_, cols, ld, pstride, rkind, (keep && keep->defined()) ? *keep : Tensor());
  });
  m.def("wq_append", &wq_append);
  m.def("flush", &flush_sink);
  m.def("set_wt", [](int64_t base, int64_t nbytes, int64_t buf, std::vector<int64_t> offs, std::vector<std::array<int64_t, 4>> ents) {
    S.wt_base = base, S.wt_nbytes = nbytes, S.wt_buf = buf, S.wt_off = std::move(offs), S.wt_ent = std::move(ents);
  });
  m.def("launches", []() { return S.n_launch; });
  m.def("linear_fast_ok", &linear_fast_ok);
  m.def("linear_ln_fast_ok", [](const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, const Tensor& gamma, const Tensor& beta) {
    return linear_fast_ok(x, w, b) && sink_dst(gamma) && sink_dst(beta) && gamma.requires_grad() && beta.requires_grad();
  });
  m.def("linear", [](const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, const c10::optional<Tensor>& addend, bool keep32) {
    return LinearFn::apply(x, w, b, addend, keep32);
  });
  m.def("linear_ln_relu", [](const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, const c10::optional<Tensor>& addend,
                             const Tensor& gamma, const Tensor& beta) { return LinearLnReluFn::apply(x, w, b, addend, gamma, beta); });
  m.def("ew", [](int64_t op, const Tensor& a, const Tensor& b) { return EwFn::apply(op, a, b); });
  m.def("linear_ln_dot_fast_ok", &linear_ln_dot_fast_ok);
  m.def("linear_ln_relu_dot", [](const Tensor& x, const Tensor& w, const Tensor& b, const Tensor& gamma, const Tensor& beta, const Tensor& w2,
                                 const Tensor& b2) { return LinearLnReluDotFn::apply(x, w, b, gamma, beta, w2, b2); });
  m.def("all_in_sink", &all_in_sink);
  m.def("bondffn_fwd", &bondffn_fwd);
  m.def("bondffn_bwd", &bondffn_bwd);
  m.def("edge_tail_fwd", &edge_tail_fwd);
  m.def("edge_tail_bwd", &edge_tail_bwd);
  m.def("posffn_fwd", &posffn_fwd);
  m.def("posffn_bwd", &posffn_bwd);
  m.def("nodemsg_fwd", &nodemsg_fwd);
  m.def("nodemsg_bwd", &nodemsg_bwd);
}
