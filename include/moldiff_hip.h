/* moldiff_hip.h -- C ABI of libmoldiff_hip.so: the MI355X (gfx950) implementation of MolDiff's
 * denoising hot path.  Plain C: opaque handles, raw pointers, sizes; no C++/torch types.
 *
 * The reference (pengxingang/MolDiff @ 2024_08_07) has no FFI layer of its own -- its boundary is
 * the Python model API plus one third-party op -- so each entry point below names the reference
 * function (file:line) whose arithmetic it replaces.  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns MDX_OK (0) or an error code; mdx_last_error() gives the message
 *     (thread-local).  No C++ exception crosses this boundary.
 *   - pointers named h_* / *_host are HOST pointers; everything else is a DEVICE pointer valid on
 *     the current HIP device.  `stream` is a hipStream_t passed as void*.
 *   - no hidden per-call device allocation: forward-type calls take a caller-owned workspace of
 *     at least mdx_workspace_bytes() bytes (256-byte aligned).  Handles own only their packed
 *     weights / graph index arrays.
 *   - edge-indexed tensors crossing this boundary use the REFERENCE edge order
 *     (edge_index = cat[halfedge_index, flip(halfedge_index)], models/model.py:269); the library
 *     re-orders internally.  float = IEEE fp32, indices = int64 like the reference's LongTensors.
 */
#ifndef MOLDIFF_HIP_H
#define MOLDIFF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_OK 0
#define MDX_ERR_ARG 1         /* bad argument / shape */
#define MDX_ERR_HIP 2         /* HIP runtime error (no device, launch failure, OOM) */
#define MDX_ERR_STATE 3       /* missing parameters, workspace too small, call order */
#define MDX_ERR_UNSUPPORTED 4 /* configuration outside what the kernels are built for */

#define MDX_KIND_MOLDIFF 0  /* models/model.py:13-46         (embedders + denoiser + 2 decoders) */
#define MDX_KIND_BONDPRED 1 /* models/bond_predictor.py:12-37 (embedders + encoder + edge decoder) */
#define MDX_KIND_NET 2      /* a bare NodeEdgeNet, models/graph.py:298-346 */

typedef struct mdx_model_s* mdx_model_t;
typedef struct mdx_graph_s* mdx_graph_t;

typedef struct mdx_config {
  int32_t kind;           /* MDX_KIND_* */
  int32_t node_dim;       /* 256 */
  int32_t edge_dim;       /* 64  */
  int32_t num_blocks;     /* 6 (denoiser) / 8 (bond predictor encoder) */
  float cutoff;           /* distance smearing stop: 15 / 20 */
  int32_t num_gaussians;  /* 16 */
  int32_t update_pos;     /* 1 / 0 */
  int32_t time_dim;       /* 10 / 20 (ignored for MDX_KIND_NET) */
  int32_t num_timesteps;  /* 1000 */
  int32_t num_node_types; /* 8 */
  int32_t num_edge_types; /* 6 (MolDiff) / 5 (bond predictor) */
} mdx_config;

const char* mdx_last_error(void);
int mdx_version(void);
int mdx_device_count(int* count); /* 0 devices is MDX_OK with *count = 0 */

/* ---- model handle: parameters enter by state_dict key (Appendix B of SURVEY.md; the strict
 * load_state_dict contract of scripts/sample_drug3d.py:79) ---------------------------------- */
int mdx_model_create(const mdx_config* cfg, mdx_model_t* out);
int mdx_model_destroy(mdx_model_t m);
/* h_data: host fp32, row-major, `ndim` dims in `shape`.  Key prefix for MDX_KIND_NET is "" (keys as
 * inside NodeEdgeNet, e.g. "edge_embs.0.weight"), otherwise the full model key ("denoiser.edge_embs.0.weight"). */
int mdx_model_set_param(mdx_model_t m, const char* key, const float* h_data, const int64_t* shape, int32_t ndim);
/* Packs all weights into MFMA fragment order and uploads them.  Fails with MDX_ERR_STATE listing the
 * first missing key.  Must be called after the last set_param and before any forward. */
int mdx_model_finalize(mdx_model_t m);
/* Matrix path of the per-edge Linear layers (reference models/common.py:181-201, models/graph.py:29-55,133-141,268-295; the
 * reference computes them in fp32).  MDX_MATRIX_EXACT_F32 (default): fp32-input MFMA, bit-for-bit an fmaf chain.
 * MDX_MATRIX_SPLIT_F16 (opt-in): every operand split into float16 hi + lo halves (22 significand bits), three float16 MFMAs
 * per k-group with fp32 accumulation -- same parity tests, ~2x the matrix throughput.  Needs a finalized model; refused with
 * MDX_ERR_UNSUPPORTED if a weight lies outside float16's range (|w| >= 65504); a handle that is re-finalized with such weights
 * drops back to the exact path.  ACTIVATION RANGE: operands of the per-edge layers must stay below 65504 in magnitude too -- only the
 * weights can be checked at pack time.  LayerNorm outputs are bounded by their gains; the un-normalised operands are the pairwise
 * products bond_linear(b) * node_linear(n) and edge_net(e) * node_net(h[col]) (graph.py:45,139) and the raw inputs of a bare
 * NodeEdgeNet: beyond the range they become inf / NaN on this path where the exact path stays finite (NaN/inf are propagated
 * silently, like the reference does).  tests/golden/stress.npz exercises 3.7e4.  Takes effect on the next forward / step. */
#define MDX_MATRIX_EXACT_F32 0
#define MDX_MATRIX_SPLIT_F16 1
int mdx_model_set_matrix_path(mdx_model_t m, int32_t path);
int mdx_model_get_matrix_path(mdx_model_t m, int32_t* path);
/* Lower clamp of the distance smearing, GaussianSmearing(start, stop = cutoff) of models/graph.py:330-333 / common.py:233-235.
 * 0 (the default) in every shipped config; the offset / coeff tables always come from the state_dict.  Any time before a forward. */
int mdx_model_set_smear_start(mdx_model_t m, float start);

/* ---- graph handle: CSR plan of one packed batch (utils/transforms.py:125-156 produces the inputs) */
/* h_edge_index: (2,E) int64 row-major, row 0 = left/row, row 1 = right/col; h_batch_node: (N) int64,
 * non-decreasing.  h_mol_ids: (n_graphs) int64 global molecule ids for noise keying, or NULL = 0..n_graphs-1. */
int mdx_graph_create(int64_t n_nodes, int64_t n_edges, const int64_t* h_edge_index, const int64_t* h_batch_node,
                     int64_t n_graphs, const int64_t* h_mol_ids, mdx_graph_t* out);
int mdx_graph_destroy(mdx_graph_t g);
/* Host-only plan (no GPU needed; used by the CPU tests).  All outputs int32, caller-allocated:
 * left/right/int2ref/col_eids: E, row_ptr/col_ptr: N+1. */
int mdx_graph_plan_host(int64_t n_nodes, int64_t n_edges, const int64_t* h_edge_index, int32_t* left, int32_t* right,
                        int32_t* int2ref, int32_t* row_ptr, int32_t* col_ptr, int32_t* col_eids);

size_t mdx_workspace_bytes(int64_t n_nodes, int64_t n_edges);

/* ---- network (models/graph.py) ---------------------------------------------------------------- */
/* NodeEdgeNet.forward, graph.py:348-367.  node_time (N), edge_time (E, reference order) are t/T floats. */
int mdx_net_forward(mdx_model_t m, mdx_graph_t g, const float* h_node, const float* pos, const float* h_edge,
                    const float* node_time, const float* edge_time, float* h_node_out, float* pos_out,
                    float* h_edge_out, void* ws, size_t ws_bytes, void* stream);
/* NodeBlock.forward, graph.py:29-55: out (N,256) = block `i`'s node update (no residual). */
int mdx_node_block(mdx_model_t m, mdx_graph_t g, int32_t i, const float* x, const float* edge_attr,
                   const float* node_time, float* out, void* ws, size_t ws_bytes, void* stream);
/* EdgeBlock.forward, graph.py:268-295: out (E,64) reference order (no residual). */
int mdx_edge_block(mdx_model_t m, mdx_graph_t g, int32_t i, const float* h_bond, const float* h_node,
                   const float* bond_time, float* out, void* ws, size_t ws_bytes, void* stream);
/* BondFFN.forward of an EdgeBlock (models/graph.py:133-141) on its own:
 *   out[e] = inter_module((W_bl bond[e]) * (W_nl node[e])) * sigmoid(gate([bond[e] | node[e] | time[e]]))
 * side 0 = edge_blocks[i].bond_ffn_left, 1 = bond_ffn_right.  node_feat holds one ALREADY GATHERED row per edge (the
 * reference calls it with h_node[left] / h_node[right]), so `g` must be the identity graph of E nodes and E edges
 * (edge e = (e, e)); bond_feat (E,64), node_feat (E,256), time (E), out (E,64). */
int mdx_bond_ffn(mdx_model_t m, mdx_graph_t g, int32_t i, int32_t side, const float* bond_feat, const float* node_feat,
                 const float* time, float* out, void* ws, size_t ws_bytes, void* stream);
/* PosUpdate.forward, graph.py:384-396: rel (E,3), dist (E) reference order; out (N,3) = delta_pos. */
int mdx_pos_update(mdx_model_t m, mdx_graph_t g, int32_t i, const float* h_node, const float* h_edge,
                   const float* rel, const float* dist, const float* edge_time, float* out, void* ws,
                   size_t ws_bytes, void* stream);
/* torch_scatter.scatter_sum(src, index, dim=0, dim_size=N) for index = left (by_right = 0) or right
 * (by_right = 1) of the graph; src (E,C) reference order, C in {3, 64, 256}; call sites graph.py:50,279,283,394.
 * by_right | 2: src is already in the plan's internal edge order (mdx_graph_plan_host) -- the segment-sum kernel alone, without the
 * boundary permutation (what bench.py times as the scatter/gather primitive). */
int mdx_segment_sum(mdx_graph_t g, const float* src, int32_t C, int32_t by_right, float* out, void* ws,
                    size_t ws_bytes, void* stream);

/* ---- model forward ------------------------------------------------------------------------------ */
/* MolDiff.forward, model.py:204-234.  h_node_pert (N,Kn), h_edge_pert (E,Ke) reference order (may be NULL
 * when h_halfedge_pert (Eh,Ke) is given: then both directions use it, model.py:273), t (n_graphs) int64. */
int mdx_moldiff_forward(mdx_model_t m, mdx_graph_t g, const float* h_node_pert, const float* pos_pert,
                        const float* h_edge_pert, const float* h_halfedge_pert, const int64_t* t, float* pred_node,
                        float* pred_pos, float* pred_halfedge, void* ws, size_t ws_bytes, void* stream);
/* BondPredictor.forward, bond_predictor.py:128-162: logits (Eh, num_edge_types).  `tape` (caller-owned,
 * mdx_bondpred_tape_bytes(), 256-byte aligned) records the per-block state the backward needs; NULL = inference only. */
size_t mdx_bondpred_tape_bytes(int64_t n_nodes, int64_t n_edges, int32_t num_blocks);
int mdx_bondpred_forward(mdx_model_t m, mdx_graph_t g, const float* h_node, const float* pos, const int64_t* t,
                         float* logits, void* ws, size_t ws_bytes, void* tape, size_t tape_bytes, void* stream);
/* The autograd call of the guidance block, model.py:312-325 (torch.autograd.grad(scalar(logits), pos_in)):
 * gpos (N,3) = scale * dL/dpos given glogits = dL/dlogits (Eh, num_edge_types) and the tape of the matching
 * forward (same model, graph, pos).  Hand-written data-gradient backward through all encoder blocks. */
int mdx_bondpred_backward(mdx_model_t m, mdx_graph_t g, const float* pos, const float* glogits, float scale,
                          float* gpos, void* ws, size_t ws_bytes, void* tape, size_t tape_bytes, void* stream);

/* ---- transitions (models/transition.py, models/diffusion.py) -------------------------------------- */
/* ContigousTransition.get_prev_from_recon, transition.py:44-63 (eps passed in). x (n,3). */
int mdx_pos_posterior(const float* coef_x0, const float* coef_xt, const float* std, const float* x_t,
                      const float* x_recon, const float* eps, const int64_t* t, const int64_t* batch, int64_t n,
                      float* out, void* stream);
/* The same for rows of any width C: x (n,C).  categorical_space = 'continuous' runs the atom (C = num_node_types) and bond
 * (C = num_edge_types) features through it with their own schedules, models/model.py:301-304. */
int mdx_gauss_posterior(const float* coef_x0, const float* coef_xt, const float* std, const float* x_t,
                        const float* x_recon, const float* eps, const int64_t* t, const int64_t* batch, int64_t n, int32_t C,
                        float* out, void* stream);
/* GeneralCategoricalTransition.q_v_posterior(v0_prob=True), transition.py:285-315.  in0 = log_v0, or raw
 * logits when is_logits != 0 (fuses F.log_softmax of model.py:291,297).  (n,K), K <= 8. */
int mdx_cat_posterior(const float* q_mats, const float* qT_onestep, int32_t K, int32_t T, const float* in0,
                      int32_t is_logits, const float* log_vt, const int64_t* t, const int64_t* batch, int64_t n,
                      float* out, void* stream);
/* Training / add_noise: GeneralCategoricalTransition.add_noise (models/transition.py:266-283: index_to_log_onehot, q_vt_pred, the
 * Gumbel-max draw of q_vt_sample with the uniforms u (n,K) passed in, onehot_encode) in one launch: v (n) class ids of the clean batch
 * -> onehot (n,K) of the drawn classes, log_vt = log(clamp(onehot, 1e-30)), log_v0 = log(clamp(onehot(v), 1e-30)).  log_off = log(1e-30)
 * as the caller's torch evaluates it in fp32.  Ids outside [0, K) are clamped (the caller reports them, models/diffusion.py:54). */
int mdx_op_cat_add_noise(const float* q_mats, int32_t K, int32_t T, const int64_t* v, const int64_t* t, const int64_t* batch, const float* u,
                         int64_t n, float log_off, float* onehot, float* log_vt, float* log_v0, void* stream);
/* Training: the categorical loss rows of models/model.py:170-189 and their gradient w.r.t. the decoder logits in one launch --
 * log_softmax, q_v_posterior (transition.py:285-315) of the true and the predicted classes, compute_v_Lt (transition.py:317-327:
 * KL for t > 0, decoder NLL at t == 0) and the backward torch.autograd would run through them.  logits / log_vt / log_v0 (n,K), K <= 8;
 * row_loss (n); dlogits (n,K) = d row_loss[i] / d logits[i][:] (the caller scales by 100 / n x the upstream gradient). */
int mdx_op_cat_loss(const float* q_mats, const float* qT_onestep, int32_t K, int32_t T, const float* logits, const float* log_vt,
                    const float* log_v0, const int64_t* t, const int64_t* batch, int64_t n, float* row_loss, float* dlogits,
                    void* stream);
/* log_sample_categorical, diffusion.py:79-85 (u passed in) + onehot_encode, transition.py:255. */
int mdx_gumbel_argmax(const float* logits, const float* u, int32_t K, int64_t n, int64_t* cls, float* onehot,
                      void* stream);
/* GeneralCategoricalTransition.sample_init, transition.py:331-339 (called at model.py:245-246): the prior draw.  The
 * reference's logits there are FLOAT64 (log(init_prob + 1e-30).clamp_min(-32) of a float64 numpy array), so rand_like,
 * the Gumbel transform of diffusion.py:79-85 and the argmax all run in float64; this entry point does the same.
 * logits64: K (<= 8) doubles on the HOST (one row, the reference repeats it n times); u: (n,K) uniforms on the device,
 * float64 when u_is_f64 else float32 (widened exactly); outputs (any may be NULL): cls (n) int64, onehot (n,K),
 * log_onehot (n,K) = log(onehot.clamp(min=1e-30)) with log_off = the float32 log(1e-30) of the caller's libm,
 * cls8 (n) one byte per row (compact trajectory frame). */
int mdx_prior_draw(const double* logits64, int32_t K, const void* u, int32_t u_is_f64, int64_t n, int64_t* cls,
                   float* onehot, float* log_onehot, float log_off, uint8_t* cls8, void* stream);
/* dU/dlogits of the default 'uncertainty' guidance objective U = sum_h log sigmoid(-logsumexp_k logits[h,k])
 * (model.py:322-324); glogits (n,K).  Feed to mdx_bondpred_backward with scale = -guidance_scale to get delta. */
int mdx_guidance_uncertainty_grad(const float* logits, int32_t K, int64_t n, float* glogits, void* stream);
/* dst[i] += src[i]  (pos_prev = pos_prev + delta, model.py:362). */
int mdx_add_inplace(float* dst, const float* src, int64_t n, void* stream);
/* Philox4x32-10 noise for draw index `draw` (0 = prior, i+1 = loop iteration i): eps_pos (N,3) ~ N(0,1),
 * u_node (N,Kn), u_halfedge (Eh,Ke) ~ U[0,1).  Any output may be NULL.  Replaces torch.randn_like /
 * rand_like at transition.py:60, diffusion.py:80. */
int mdx_noise(mdx_graph_t g, uint64_t seed, int32_t draw, int32_t Kn, int32_t Ke, float* eps_pos, float* u_node,
              float* u_halfedge, void* stream);

/* ---- one iteration of the reverse chain in a single call (models/model.py:272-308: denoiser forward, Gaussian
 * posterior on the positions, categorical posteriors + Gumbel-max draws on atom and bond types).  Guidance, when wanted,
 * is applied by the caller afterwards (mdx_bondpred_forward / mdx_guidance_uncertainty_grad / mdx_bondpred_backward /
 * mdx_add_inplace).  All pointers are device pointers; `cur` is read, `next` and `pred_*` are written.
 *   tables: the frozen schedule tensors of the checkpoint (pos_transition.coef_x0 / coef_xt / std (T);
 *           node/edge_transition.q_mats and transpopse_q_onestep_mats (T,K,K)).
 *   t (B) int64 must hold `step` for every molecule; batch_node (N) / batch_halfedge (Eh) int64 as in the reference.
 *   eps_pos (N,3) ~ N(0,1), u_node (N,Kn), u_halfedge (Eh,Ke) ~ U[0,1): this step's draws (mdx_noise or the caller's own). */
typedef struct {
  const float *pos_coef_x0, *pos_coef_xt, *pos_std;
  const float *node_q_mats, *node_qT_onestep;
  const float *edge_q_mats, *edge_qT_onestep;
} mdx_tables;
typedef struct {
  float *h_node, *pos, *h_halfedge;   /* (N,Kn) one-hot, (N,3), (Eh,Ke) one-hot */
  float *log_node, *log_halfedge;     /* (N,Kn), (Eh,Ke) log-probabilities of the current types */
} mdx_state;
int mdx_sample_step(mdx_model_t m, mdx_graph_t g, const mdx_tables* tables, const int64_t* t, const int64_t* batch_node,
                    const int64_t* batch_halfedge, const mdx_state* cur, const mdx_state* next, float* pred_node, float* pred_pos,
                    float* pred_halfedge, const float* eps_pos, const float* u_node, const float* u_halfedge, void* ws,
                    size_t ws_bytes, void* stream);

/* ---- the WHOLE loop body of MolDiff.sample in one call (models/model.py:271-372): per-graph time tensor (:272), noise
 * draws (transition.py:60, diffusion.py:80), denoiser forward (:274-284), Gaussian posterior (:287-288), categorical
 * posteriors + Gumbel-max draws (:291-307) and, when `guide` is given, the bond-predictor guidance with the default
 * 'uncertainty' objective (:309-362: predictor forward at the step's INPUT state, dU/dlogits, hand-written backward
 * w.r.t. the positions, pos_prev += -scale * grad).  With guide->side_stream set, the guidance chain runs on that stream
 * concurrently with the denoiser of the same step (both only read the input state); the library orders the two streams
 * with events and `stream` ends up holding the complete result.  A C caller runs the chain with
 *     for (i = 0; i < T; ++i) { mdx_sample_step_full(..., T - 1 - i, ..., frames i and i + 1, ...); }
 *   step        : the diffusion step id (the reference's `step`), identical for every molecule of the batch
 *   noise       : draw >= 0: Philox draw index (0 = prior, i + 1 = loop iteration i), the buffers receive this step's noise
 *                 (mdx_noise with `seed`); draw < 0: the buffers already hold it (explicit noise, e.g. parity tests)
 *   t_buf       : (n_graphs) int64 device scratch that receives `step` (the reference's time_step tensor)
 *   node_cls / halfedge_cls : optional (N) / (Eh) uint8 outputs = the sampled class ids of `next` (a frame of the compact
 *                 trajectory: one byte per atom / half-edge instead of a one-hot fp32 row, models/model.py:365-367)
 *   guide       : NULL = no guidance */
typedef struct {
  mdx_model_t predictor;          /* BondPredictor handle (MDX_KIND_BONDPRED) */
  float scale;                    /* sample.guidance[1] */
  void* tape;                     /* mdx_bondpred_tape_bytes(N, E, predictor blocks), 256-byte aligned */
  size_t tape_bytes;
  void* ws2;                      /* a second workspace of mdx_workspace_bytes(N, E): the predictor's own */
  size_t ws2_bytes;
  float *logits, *glogits;        /* (Eh, Kb) scratch: predictor logits and dU/dlogits */
  float* delta;                   /* (N,3) out: the increment added to next->pos */
  void* side_stream;              /* NULL = run the guidance chain in line on `stream` */
} mdx_guidance;
typedef struct {
  uint64_t seed;
  int32_t draw;
  float *eps_pos, *u_node, *u_halfedge;   /* (N,3), (N,Kn), (Eh,Ke) */
} mdx_step_noise;
int mdx_sample_step_full(mdx_model_t m, mdx_graph_t g, const mdx_tables* tables, int32_t step, const int64_t* batch_node,
                         const int64_t* batch_halfedge, const mdx_state* cur, const mdx_state* next, float* pred_node,
                         float* pred_pos, float* pred_halfedge, const mdx_step_noise* noise, int64_t* t_buf, uint8_t* node_cls,
                         uint8_t* halfedge_cls, const mdx_guidance* guide, void* ws, size_t ws_bytes, void* stream);

/* ---- harness consumer of the path's outputs (next-row) ---------------------------------------
 * seperate_outputs (utils/sample.py:4-30) + FeaturizeMol.decode_output (utils/transforms.py:65-122) on the device:
 * arg-max class + soft-max confidence per atom / half-edge, mask-type atoms (class >= num_element) dropped and the
 * survivors re-indexed per molecule, bonds = half-edges with 0 < class <= num_bond_types whose atoms survived.
 * Outputs are compacted IN PLACE at each molecule's original offsets (atoms at node offset, bonds at half-edge
 * offset; order preserved): atom_type/atom_prob (N), atom_pos (N,3), n_atoms (n_graphs); bond_type/bond_prob (Eh),
 * bond_index (2,Eh) molecule-local new atom indices, n_bonds (n_graphs).  One direction per bond (the reference
 * mirrors them on the host). */
int mdx_decode_output(mdx_graph_t g, const float* pred_node, int32_t Kn, const float* pred_pos, const float* pred_halfedge,
                      int32_t Ke, int32_t num_element, int32_t num_bond_types, int32_t* atom_type, float* atom_prob,
                      float* atom_pos, int32_t* n_atoms, int32_t* bond_type, float* bond_prob, int32_t* bond_index,
                      int32_t* n_bonds, void* ws, size_t ws_bytes, void* stream);

/* ---- layer-level operators of the training path (next-row) ---------------------------------
 * The loss forward + backward of MolDiff.get_loss / BondPredictor.get_loss (models/model.py:128-201,
 * models/bond_predictor.py:84-124 + torch.autograd) is composed from these forward/backward pairs, one layer at a
 * time, by moldiff_amd/train_ops.py; activations stay in HBM between them.  All fp32, row-major, device pointers.
 *
 * sgemm_nt: C[M,N] (ldc) = A[M,K] (lda) * B[N,K]^T (ldb) + bias[N] + addend[M,N] (ldd) (bias / addend may be NULL; the
 *   addend carries the per-node part of a layer whose input is a concatenation).  nn.Linear forward (B = weight),
 *   its data gradient (B = weight^T) and, with splits > 1 and `partial` = splits*M*N floats, its weight gradient
 *   (A = dY^T, B = X^T, K = number of rows; partial sums are combined in a fixed order -> deterministic).
 * transpose: out[Cn,R] (ldo) = in[R,Cn]^T (ldi).
 * colreduce: out[N] = sum over M rows of X[M,N] (ld), times Y element-wise if Y != NULL (bias / LayerNorm parameter
 *   gradients); ws = ceil(M/512)*N floats. */
int mdx_op_sgemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, const float* addend, int64_t ldd,
                    float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t splits, float* partial, void* stream);
/* sgemm_tn: dW[N,K] (ldw) = G[M,N]^T (ldg) * X[M,K] (ldx), the weight gradient straight from the stored row-major
 *   tensors; `splits` ranges of rows, partial = (splits + ceil(splits/256))*(N*K + N) floats, fixed-order two-stage
 *   reduction; db (may be NULL) receives the bias gradient = column sums of G from the same pass. */
int mdx_op_sgemm_tn(const float* G, int64_t ldg, const float* X, int64_t ldx, float* dW, int64_t ldw, float* db, int64_t M, int64_t N,
                    int64_t K, int32_t splits, float* partial, void* stream);
/* hgemm_nt / hgemm_tn: the same two products with bf16-rounded operands on v_mfma_f32_16x16x32_bf16 and fp32
 *   accumulation / outputs (mixed-precision training; the reference trains under fp16 autocast).  Tensors stay fp32 in
 *   memory.  hgemm_nt has no split-K form. */
int mdx_op_hgemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, const float* addend, int64_t ldd,
                    float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, void* stream);
int mdx_op_hgemm_tn(const float* G, int64_t ldg, const float* X, int64_t ldx, float* dW, int64_t ldw, float* db, int64_t M, int64_t N,
                    int64_t K, int32_t splits, float* partial, void* stream);
/* The same Linear on MANY rows (the edge rows of a training batch) with the row-owner decomposition of the sampling kernels:
 * Y (M,N) = X (M,K) W^T + bias + addend with W (N,K) [transW = 0], or X W with W (K,N) [transW = 1: the grad_input GEMM, no
 * separate transpose].  K % 16 == 0, K/16 in {1,2,4,5,8,16}, N % 4 == 0, N <= 256 (mdx_op_linear_rows_supported); rows 16-byte
 * aligned.  pack_ws: mdx_op_linear_rows_ws(N, K) bytes of scratch (the weight in streaming order, rebuilt per call). */
int mdx_op_linear_rows_supported(int64_t N, int64_t K);
size_t mdx_op_linear_rows_ws(int64_t N, int64_t K);
int mdx_op_linear_rows(const float* X, int64_t ldx, const float* W, int64_t ldw, int32_t transW, const float* bias, const float* addend,
                       int64_t ldd, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, float* pack_ws, void* stream);
int mdx_op_transpose(const float* in, int64_t ldi, int64_t R, int64_t Cn, float* out, int64_t ldo, void* stream);
/* n contiguous row-major matrices transposed by one launch (every weight of a model once per optimisation step: the grad_input
 * GEMMs read W^T).  desc (device): 4 int64 per matrix {src pointer, dst pointer, R, C}; dst receives (C,R) contiguous.
 * max_tiles = the largest matrix's count of 32 x 32 tiles (sizes the launch). */
int mdx_op_transpose_batch(const int64_t* desc, int64_t n, int64_t max_tiles, void* stream);
int mdx_op_colreduce(const float* X, const float* Y, int64_t ld, int64_t M, int64_t N, float* out, float* ws, void* stream);
/* y = relu?(LayerNorm(x) * gamma + beta) over F <= 1024 features (nn.LayerNorm eps 1e-5, models/common.py MLP);
 * stats (M,2) receives (mean, rstd) for the backward.  Backward: dx (M,F), dgb (2F) = [dgamma | dbeta]; ws =
 * mdx_op_ln_relu_bwd_ws(M, F) bytes. */
int mdx_op_ln_relu_fwd(const float* x, const float* gamma, const float* beta, int64_t M, int32_t F, int32_t relu, float* y,
                       float* stats, void* stream);
int mdx_op_ln_relu_bwd(const float* dy, const float* x, const float* stats, const float* gamma, const float* beta, int64_t M,
                       int32_t F, int32_t relu, float* dx, float* dgb, float* ws, void* stream);
size_t mdx_op_ln_relu_bwd_ws(int64_t M, int32_t F);
/* The same backward when the layer after the LayerNorm is a Linear to ONE output (PosUpdate's inter MLP, 256 -> 1, reference
 * models/graph.py:388-392 through models/common.py:191-198): the upstream gradient dy[row][c] = f16(g1[row] * f16(w1[c])) is formed
 * in the kernel from g1 (M values; dt bit 0: float16) and the Linear's weight row w1 (F, fp32) -- autograd's `grad_input` GEMM with
 * K = 1, its weight transpose and the (M,F) gradient tensor are not materialised.  Leaves the [dgamma | dbeta] partial rows in ws
 * (mdx_op_ln_relu_bwd_rows(M) rows of 2F floats) for mdx_op_reduce_deferred.  F in {32, 64, 128, 256}; dt bit 1: x, bit 2: dx float16. */
int mdx_op_ln_relu_bwd_r1_t(const void* g1, const float* w1, const void* x, const float* stats, const float* gamma, const float* beta,
                            int64_t M, int32_t F, int32_t relu, void* dx, float* ws, int32_t dt, void* stream);
/* element-wise pairs, op: 0 a+b, 1 a-b, 2 a*b, 3 a*sigmoid(b).  Backward writes da / db (either may be NULL). */
int mdx_op_ew_fwd(int32_t op, const float* a, const float* b, float* out, int64_t n, void* stream);
int mdx_op_ew_bwd(int32_t op, const float* a, const float* b, const float* g, float* da, float* db, int64_t n, void* stream);
/* y[i] = x[idx[i]] (rows of F floats) and its adjoint out[r] = sum_{j in [ptr[r],ptr[r+1])} src[order[j]]
 * (torch_scatter.scatter_sum with `order` = stable argsort of the index; sequential per output -> deterministic). */
int mdx_op_gather_rows(const float* x, const int64_t* idx, int64_t M, int32_t F, float* y, void* stream);
int mdx_op_segsum_rows(const float* src, const int64_t* order, const int64_t* ptr, int64_t R, int32_t F, float* out, void* stream);
/* y = a * t[idx] (product with a gathered per-node row, e.g. h_edge * h_node[col], models/graph.py:44) without the
 * gathered (rows x F) intermediate; backward: da = g * t[idx], dt[r] = sum over the rows of segment r of g * a. */
int mdx_op_mul_gather_fwd(const float* a, const float* t, const int64_t* idx, int64_t M, int32_t F, float* y, void* stream);
int mdx_op_mul_gather_bwd(const float* g, const float* a, const float* t, const int64_t* idx, const int64_t* order, const int64_t* ptr,
                          int64_t M, int64_t R, int32_t F, float* da, float* dt, void* stream);
/* rel = pos[l] - pos[r], dist = |rel| (models/graph.py:349-350); backward: g (E,3) = drel + ddist * rel / dist, the
 * caller scatters +g to l and -g to r.  drel / ddist may be NULL. */
int mdx_op_edge_geom_fwd(const float* pos, const int64_t* l, const int64_t* r, int64_t E, float* rel, float* dist, void* stream);
int mdx_op_edge_geom_bwd(const float* rel, const float* dist, const float* drel, const float* ddist, int64_t E, float* g, void* stream);
/* GaussianSmearing (models/common.py): out[e,k] = exp(coef[k] * (clamp(d[e], lo, hi) - off[k])^2) and d/dd. */
int mdx_op_smear_fwd(const float* d, const float* off, const float* coef, int32_t G, float lo, float hi, int64_t E, float* out,
                     void* stream);
int mdx_op_smear_bwd(const float* d, const float* off, const float* coef, int32_t G, float lo, float hi, int64_t E, const float* gout,
                     float* gd, void* stream);
/* PosUpdate force (models/graph.py:393): out[e] = w[e] * rel[e] / d[e] / (d[e] + 1) and its three gradients. */
int mdx_op_force_fwd(const float* w, const float* rel, const float* d, int64_t E, float* out, void* stream);
int mdx_op_force_bwd(const float* w, const float* rel, const float* d, const float* g, int64_t E, float* gw, float* grel, float* gd,
                     void* stream);

/* Optimizer step on flat buffers (torch.optim.AdamW + torch.nn.utils.clip_grad_norm_, reference utils/train.py:64-70,
 * scripts/train_drug3d.py:107-108).  sumsq: out[0] = sum x^2 (fixed-order two-stage; ws = 1024 floats).  adamw: one
 * decoupled-weight-decay Adam step, step = 1, 2, ...; when gnorm2 != NULL the gradient is scaled by
 * min(max_norm / (sqrt(*gnorm2) + 1e-6), 1) on the fly. */
int mdx_op_sumsq(const float* x, int64_t n, float* out, float* ws, void* stream);
int mdx_op_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int64_t step, const float* gnorm2, float max_norm, void* stream);

/* Mixed-precision forms of mdx_op_sgemm_nt / mdx_op_sgemm_tn (the reference trains under torch.autocast(dtype=float16) + GradScaler,
 * scripts/train_drug3d.py:86-109): operands rounded to half_kind (1 = bfloat16, 2 = float16) on their way into LDS, half MFMAs,
 * fp32 accumulation; round_out != 0 rounds the result to the same type before it is stored in its fp32 container -- the value a
 * Linear (and the gradient of a weight autocast cast to half) has there; float16 overflows to +-inf beyond 65504.
 * mdx_op_ew_fwd: op | (kind << 8) and mdx_op_mul_gather_fwd: F | (kind << 16) round their products the same way. */
int mdx_op_xgemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, const float* addend, int64_t ldd,
                    float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t half_kind, int32_t round_out, void* stream);
int mdx_op_xgemm_tn(const float* G, int64_t ldg, const float* X, int64_t ldx, float* dW, int64_t ldw, float* db, int64_t M, int64_t N,
                    int64_t K, int32_t splits, float* partial, int32_t half_kind, int32_t round_out, void* stream);

/* Half storage (round 3): the `_t` forms of the training operators take `dt`, a bit mask of the tensors that are stored as float16
 * instead of fp32 (element strides and feature counts are unchanged).  In the mixed-precision mode every Linear / LayerNorm / product
 * result is a float16 VALUE; keeping it in a float16 container halves the HBM traffic these memory-bound operators are limited by.
 * Bits, in argument order -- xgemm_nt_t: 0 A, 1 addend, 2 C (needs half_kind 2);  xgemm_tn_t: 0 G, 1 X (dW, db stay fp32);
 * ln_relu_fwd_t: 0 x, 1 y;  ln_relu_bwd_t: 0 dy, 1 x, 2 dx;  ew_fwd_t: 0 a, 1 b, 2 out;  ew_bwd_t: 0 a, 1 b, 2 g, 3 da, 4 db;
 * gather_rows_t: 0 x, 1 y;  segsum_rows_t: 0 src, 1 out;  mul_gather_fwd_t: 0 a, 1 t, 2 y;  mul_gather_bwd_t: 0 g, 1 a, 2 t, 3 da, 4 dt.
 * dt = 0 is the fp32 operator of the same name without `_t`.  Half storage needs feature counts that are multiples of 4. */
int mdx_op_xgemm_nt_t(const void* A, int64_t lda, const float* B, int64_t ldb, const float* bias, const void* addend, int64_t ldd,
                      void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int32_t half_kind, int32_t round_out, int32_t dt, void* stream);
int mdx_op_xgemm_tn_t(const void* G, int64_t ldg, const void* X, int64_t ldx, float* dW, int64_t ldw, float* db, int64_t M, int64_t N,
                      int64_t K, int32_t splits, float* partial, int32_t half_kind, int32_t round_out, int32_t dt, void* stream);
/* Linear + LayerNorm(+ReLU) in ONE launch (common.MLP's first two layers, reference models/common.py:191-196; replaces
 * F.linear + F.layer_norm + F.relu there): C = the Linear's result exactly as mdx_op_xgemm_nt_t stores it (the backward needs it),
 * post (M,N; row stride ldp) = relu(LN(C)), stats (M,2) = mean, rstd -- mdx_op_ln_relu_fwd_t's formula on C in another summation order
 * (equal to rounding, not bit for bit) and what mdx_op_ln_relu_bwd_t reads.  A float16 C requires round_out = 1.  dt bits: 0 A, 1 addend, 2 C, 3 post.  Built for float16 rows on the row-owner kernel only (A float16,
 * M >= 1024, K in {32, 64, 128, 256}, N in {32, 64, 128, 256}: mdx_op_xgemm_nt_ln_supported); anything else returns
 * MDX_ERR_UNSUPPORTED and the caller runs the two operators. */
int mdx_op_xgemm_nt_ln_supported(int64_t M, int64_t N, int64_t K);
int mdx_op_xgemm_nt_ln_t(const void* A, int64_t lda, const float* B, int64_t ldb, const float* bias, const void* addend, int64_t ldd,
                         void* C, int64_t ldc, const float* gamma, const float* beta, void* post, int64_t ldp, float* stats, int32_t relu,
                         int64_t M, int64_t N, int64_t K, int32_t half_kind, int32_t round_out, int32_t dt, void* stream);
int mdx_op_ln_relu_fwd_t(const void* x, const float* gamma, const float* beta, int64_t M, int32_t F, int32_t relu, void* y, float* stats,
                         int32_t dt, void* stream);
int mdx_op_ln_relu_bwd_t(const void* dy, const void* x, const float* stats, const float* gamma, const float* beta, int64_t M, int32_t F,
                         int32_t relu, void* dx, float* dgb, float* ws, int32_t dt, void* stream);
int mdx_op_ew_fwd_t(int32_t op, const void* a, const void* b, void* out, int64_t n, int32_t dt, void* stream);
/* out = srcs[0] + ... + srcs[k-1] (k <= 12 tensors of n elements, half[j] != 0: float16 container), summed in fp32 in argument order,
 * stored once: the gradient of a tensor with several consumers in one launch (torch.autograd accumulates them pairwise). */
/* Segment order of the right end points from that of the left ones for a directed edge list [half-edges ; flipped half-edges]
 * (models/model.py:269): order_left = stable argsort of left (E = 2 Eh), ptr (R + 1) its segment starts (shared by both index vectors);
 * order_right receives the stable argsort of right.  Replaces a second sort per training step. */
int mdx_op_plan_flip(const int64_t* order_left, const int64_t* ptr, int64_t R, int64_t Eh, int64_t* order_right, void* stream);
int mdx_op_sum_n(const void* const* srcs, const int32_t* half, int32_t k, int64_t n, void* out, int32_t out_half, void* stream);
int mdx_op_ew_bwd_t(int32_t op, const void* a, const void* b, const void* g, void* da, void* db, int64_t n, int32_t dt, void* stream);
int mdx_op_gather_rows_t(const void* x, const int64_t* idx, int64_t M, int32_t F, void* y, int32_t dt, void* stream);
/* (segsum_rows_t: dt bit 0 / 1 = src / out float16; bit 2 = the caller accepts a summation order other than the sequential CSR order -- fp32
 * rows are then dealt to four threads per element like float16 rows always are; without it fp32 sums keep the order of torch's index_add) */
int mdx_op_segsum_rows_t(const void* src, const int64_t* order, const int64_t* ptr, int64_t R, int32_t F, void* out, int32_t dt,
                         void* stream);
int mdx_op_mul_gather_fwd_t(const void* a, const void* t, const int64_t* idx, int64_t M, int32_t F, void* y, int32_t dt, void* stream);
int mdx_op_mul_gather_bwd_t(const void* g, const void* a, const void* t, const int64_t* idx, const int64_t* order, const int64_t* ptr,
                            int64_t M, int64_t R, int32_t F, void* da, void* dtab, int32_t dt, void* stream);

/* Deferred gradient reduction: mdx_op_sgemm_tn / mdx_op_xgemm_tn(_t) with dW == NULL (db != NULL still requests the bias partials)
 * and mdx_op_ln_relu_bwd(_t) with dgb == NULL leave their split partials in the caller's buffer (layout: mdx_op_wgrad_layout -> number
 * of partials S and the float offset of the [S][N] bias partials; LayerNorm: mdx_op_ln_relu_bwd_rows(M) rows of 2F floats in ws).
 * mdx_op_reduce_deferred sums every record of a device table in ONE launch and ADDS it to its destination (a parameter's slot in a flat
 * gradient buffer): record = 8 x int64 {P, dst, S, rows, cols, ld, pstride, rkind | chunked << 6 | store << 7 | first_block << 8};
 * total_blocks = sum of ceil(rows * cols / 128) (x ceil(S / 256) for a chunked record; first_block counts the same blocks).  A record
 * with S > 256 is given as `chunked`: the launch stores the sum of every 256 partials in the scratch planes behind them
 * (P + (S + c) * pstride, which mdx_op_wgrad_layout / mdx_op_ln_relu_bwd_ws reserve), and the caller sums those ceil(S / 256) planes
 * with a plain record {P + S * pstride, dst, ceil(S / 256), ...} in a second call.  Same fixed summation order as the per-layer reduction kernels. */
int mdx_op_wgrad_layout(int64_t M, int64_t N, int64_t K, int32_t splits, int32_t half, int64_t* S, int64_t* bias_off);
int64_t mdx_op_ln_relu_bwd_rows(int64_t M);
int mdx_op_reduce_deferred(const int64_t* desc, int32_t n, int64_t total_blocks, void* stream);

/* Queued (grouped) weight gradients, float16 autocast mode.  A step's weight-gradient contractions dW = dY^T X (the backward of every
 * nn.Linear of reference models/common.py:181-201 and models/graph.py, i.e. torch.autograd's mm_backward for the weight) feed nothing
 * before the optimizer, so the host may queue them and run a whole table per launch.  mdx_op_wgrad_plan: tile class and layout of one
 * contraction -- out[0..7] = kind (0..3 transpose-read kernel with tiles 128x128 / 128x64 / 64x128 / 64x64 (n x k), 4 converting kernel,
 * 5 / 6 the transpose-read kernel with 32x64 / 64x32 tiles, 7 a scaled column sum for N == 1 or K == 1),
 * gx, gy (tiles along K, N), S (row ranges), mper (rows per range), float offset of the [S][N] bias partials, floats of the whole partial
 * area (the layout of mdx_op_xgemm_tn_t / mdx_op_wgrad_layout), blocks.  dt bit 0 / 1: dY / X stored as float16; `aligned`: both start on
 * 16 bytes.  mdx_op_wgrad_grouped: one launch over a DEVICE table of n records of one kind -- 16 x int64 {dY, X, P, Pb (0 = no bias
 * partials), ldg, ldx, M, N, K, mper, gx, gy, S, dt, first_block, 0}, first_block = running sum of the records' blocks,
 * total_blocks = their sum; every record's partials are bit-identical to mdx_op_xgemm_tn_t(dW = NULL) with the same row ranges and are
 * summed by mdx_op_reduce_deferred. */
int mdx_op_wgrad_plan(int64_t M, int64_t N, int64_t K, int32_t splits, int32_t dt, int64_t ldg, int64_t ldx, int32_t aligned, int64_t* out);
int mdx_op_wgrad_grouped(const int64_t* desc, int32_t n, int64_t total_blocks, int32_t kind, void* stream);

/* ---- fused row-owner training operators (round 6; csrc/mdx_train_fused.hip) ---------------------------------------------------------
 * One forward and one backward launch for a whole BondFFN of the EdgeBlock in the float16 autocast arithmetic
 * (replaces, for training, reference models/graph.py:133-141 -- bond_linear, the product with node_linear(h)[idx], the inter MLP
 * models/common.py:191-198, the gate MLP on [bond | node | time] and the sigmoid product -- as called from models/graph.py:272-279,
 * plus torch.autograd's backward through them).  Widths are those of the shipped networks: bond 64, inter 128, out 64, gate hidden 32.
 * Row tensors are float16, row-major and dense unless a stride is given; weights are fp32 (N outputs x K inputs, row stride ld*, so
 * column slices of a wider matrix are passed as they are); Wt is the gate's time COLUMN (element j at Wt[j * ldwt]).
 * NL = node_linear(h_node) (N,128) float16 and GN = the node columns of the gate's first Linear applied to h_node (N,32) fp32 are
 * computed per NODE by the caller (Linear(h[idx]) == Linear(h)[idx]); idx (E) selects the node whose features enter each row.
 * Forward writes what the backward and the weight gradients read: prod (E,128), pre1 / post1 (E,128: before / after LayerNorm+ReLU
 * of the inter MLP), inter (E,64), gpre / gpost (E,32), gate (E,64), out = inter * sigmoid(gate) (E,64). */
typedef struct {
  const void* X; int64_t ldx;                                            /* (E,64) float16, rows 16-byte aligned */
  const float* Wb; int64_t ldwb;                                         /* bond_linear.weight (128,64) */
  const float* Wi1; int64_t ldwi1; const float* bi1; const float* g1; const float* be1;   /* inter_module.net.0 (128,128) + bias, .1 LayerNorm(128) */
  const float* Wi2; int64_t ldwi2; const float* bi2;                      /* inter_module.net.3 (64,128) + bias */
  const float* Wg1; int64_t ldwg1; const float* bg1; const float* gg; const float* gbe;  /* gate.net.0 bond columns (32,64) + bias, .1 LayerNorm(32) */
  const float* Wt; int64_t ldwt;                                         /* gate.net.0 time column (32) */
  const float* Wg2; int64_t ldwg2; const float* bg2;                      /* gate.net.3 (64,32) + bias */
  const void* NL; int64_t ldnl;                                          /* (N,128) float16 */
  const float* GN; int64_t ldgn;                                         /* (N,32) fp32 */
  const int64_t* idx;                                                    /* (E) */
  const float* te;                                                       /* (E) the time input (already t / T) */
  void *prod, *pre1, *post1, *inter, *gpre, *gpost, *gate, *out;         /* float16 outputs (inputs of the backward) */
  int64_t E;
} mdx_bondffn_args;
/* Backward: gS (N,64; fp32, row stride ldgs) = dL/d scatter_sum(out, oidx); row e takes gS[oidx[e]] rounded to float16 (the gather the
 * per-operator path's scatter_sum backward performs).  Writes the float16 row gradients the weight-gradient contractions read
 * -- g_inter, g_gate (E,64), g_pre1, g_bf (E,128), g_gpre (E,32) -- plus g_x = dL/dX (E,64) and g_nl (E,128) = the per-edge dL/dNL rows
 * (the caller sums them per node with mdx_op_segsum_rows_t; dL/dGN is the same sum of g_gpre).  lnp: mdx_op_bondffn_workgroups() rows
 * of mdx_op_bondffn_lnp_floats() = 320 floats [d gamma1 | d beta1 (128 each) | d gamma_g | d beta_g (32 each)], one per workgroup,
 * to be summed by the caller (mdx_op_reduce_deferred record: S = workgroups, pstride = 320). */
typedef struct {
  mdx_bondffn_args f;
  const float* gS; int64_t ldgs;
  const int64_t* oidx;
  void *g_inter, *g_gate, *g_pre1, *g_bf, *g_nl, *g_gpre, *g_x;
  float* lnp;
} mdx_bondffn_bwd_args;
/* EdgeBlock tail + residual in one launch (replaces, for training, reference models/graph.py:281-294 and the `h_edge + ...` of :360):
 * out = H + out_transform(relu(LN(self_ffn(H) + BL[il] + BR[ir]))), BL / BR (N,64) float16 = the per-node sums S_L + node_ffn_left(h),
 * S_R + node_ffn_right(h).  pre / post (E,64): the LayerNorm's input / relu(output), kept for the backward and out_transform's weight
 * gradient.  Backward: g_out (E,64) float16 -> g_pre (E,64) = dL/d pre (self_ffn's weight gradient operand; summed per node by il / ir it
 * is dL/dBL / dL/dBR), g_h = g_out + self_ffn^T g_pre, lnp: mdx_op_bondffn_workgroups() rows of 128 floats [d gamma | d beta]. */
typedef struct {
  const void* H; int64_t ldh;
  const void* BL; int64_t ldbl; const void* BR; int64_t ldbr;
  const int64_t* il; const int64_t* ir;
  const float* Ws; int64_t ldws; const float* bs; const float* lng; const float* lnb;
  const float* Wo; int64_t ldwo; const float* bo;
  void *pre, *post, *out;
  int64_t E;
} mdx_edge_tail_args;
typedef struct {
  mdx_edge_tail_args f;
  const void* g_out; int64_t ldg;
  void *g_pre, *g_h;
  float* lnp;
} mdx_edge_tail_bwd_args;
/* PosUpdate, front of its BondFFN (replaces, for training, reference models/graph.py:388-390 with :133-141 up to the inter MLP's input):
 * a = LF[il] * RF[ir] (LF / RF (N,64) float16 = left_lin_edge(h), right_lin_edge(h), hoisted), prod = bond_linear(X) * node_linear(a)
 * (E,256), gate = gate MLP on [X | a | t] (E,1).  Weights: Wb, Wn (256,64); Wg1x, Wg1a (32,64) and Wt (32) = the bond / node / time
 * columns of gate.net.0; Wg2 (32) / bg2 (1) = gate.net.3.  Outputs (float16): a (E,64), prod (E,256), gpre / gpost (E,32), gate (E).
 * Backward: g_prod (E,256; row stride ldgp), g_gate (E) -> g_bf, g_nf (E,256) = dL/d bond_linear(X), dL/d node_linear(a); g_gpre (E,32);
 * g_x (E,64) = dL/dX; g_lf, g_rf (E,64) = the per-edge dL/dLF[il], dL/dRF[ir] rows (summed per node by the caller); lnp:
 * mdx_op_bondffn_workgroups() rows of 64 floats [d gamma | d beta] of the gate's LayerNorm. */
typedef struct {
  const void* X; int64_t ldx;
  const void* LF; int64_t ldlf; const void* RF; int64_t ldrf;
  const int64_t* il; const int64_t* ir;
  const float* te;
  const float* Wb; int64_t ldwb; const float* Wn; int64_t ldwn;
  const float* Wg1x; int64_t ldwg1x; const float* Wg1a; int64_t ldwg1a; const float* Wt; int64_t ldwt;
  const float* bg1; const float* gg; const float* gbe;
  const float* Wg2; const float* bg2;
  void *a, *prod, *gpre, *gpost, *gate;
  int64_t E;
} mdx_posffn_args;
typedef struct {
  mdx_posffn_args f;
  const void* g_prod; int64_t ldgp;
  const void* g_gate;
  void *g_bf, *g_nf, *g_gpre, *g_x, *g_lf, *g_rf;
  float* lnp;
} mdx_posffn_bwd_args;
/* NodeBlock message path in one forward and one backward launch (replaces, for training, reference models/graph.py:40-48: edge_net,
 * the product with node_net(x)[col], msg_net, the gate MLP and the sigmoid product; the scatter_sum of :50 stays the caller's).
 * Weights arrive as float16 A-operand packs written by mdx_op_pack_a (one launch for up to 10 matrices): job = {W (fp32, row stride ld),
 * n_out, n_in, perm (1: the layer's input is the previous layer's accumulators), trans (1: pack W^T), out (n_out * n_in halves)}.
 * Forward packs: W1e (256 x 64, perm 0), W2e, Wm (256 x 256, perm 1), Wg1 (the gate's edge columns, 256 x 64, perm 0), Wg2 (256 x 256,
 * perm 1); backward packs (trans 1, perm 1): Wg2^T, Wm^T, W2e^T (256 x 256), Wg1^T, W1e^T (64 x 256).
 * HN = node_net(x) (N,256) float16, PN = the gate's node + time columns applied per node (N,256) fp32, col (E) = right end point.
 * Forward outputs, (E,256) float16 each: he_pre / he_post (edge_net's LayerNorm input / relu(output)), he, p = he * HN[col], m0 =
 * msg_net(p), g_pre / g_post (gate LayerNorm), gt, msg = m0 * sigmoid(gt).  Backward: gA (N,256 fp32) = dL/d scatter_sum(msg, row);
 * writes g_m0, g_gt, g_gpre, g_he, g_pre (weight-gradient operands), g_hne = per-edge dL/dHN[col], g_x (E,64) and
 * mdx_op_bondffn_workgroups() partial rows of 1024 floats [d gamma_e | d beta_e | d gamma_g | d beta_g]. */
typedef struct { const float* W; int64_t ld; int32_t n_out, n_in, perm, trans; void* out; } mdx_pack_job;
typedef struct { mdx_pack_job job[10]; int32_t n; } mdx_pack_jobs;
typedef struct {
  const void* X; int64_t ldx;
  const void* HN; int64_t ldhn;
  const float* PN; int64_t ldpn;
  const int64_t* col;
  const void *pk_w1e, *pk_w2e, *pk_wm, *pk_wg1, *pk_wg2;
  const float *b1e, *lng_e, *lnb_e, *b2e, *bm, *bg1, *lng_g, *lnb_g, *bg2;
  void *he_pre, *he_post, *he, *p, *m0, *g_pre, *g_post, *gt, *msg;
  int64_t E;
} mdx_nodemsg_args;
typedef struct {
  mdx_nodemsg_args f;
  const float* gA; int64_t ldga;
  const int64_t* row;
  const void *pk_wg2t, *pk_wg1t, *pk_wmt, *pk_w2et, *pk_w1et;
  void *g_m0, *g_gt, *g_gpre, *g_hne, *g_he, *g_pre, *g_x;
  float* lnp;
} mdx_nodemsg_bwd_args;
int mdx_op_pack_a(const mdx_pack_jobs* jobs, void* stream);
int mdx_op_nodemsg_fwd(const mdx_nodemsg_args* a, void* stream);
int mdx_op_nodemsg_bwd(const mdx_nodemsg_bwd_args* a, void* stream);
int mdx_op_nodemsg_lnp_floats(void);
int mdx_op_posffn_fwd(const mdx_posffn_args* a, void* stream);
int mdx_op_posffn_bwd(const mdx_posffn_bwd_args* a, void* stream);
int mdx_op_posffn_lnp_floats(void);
int mdx_op_edge_tail_fwd(const mdx_edge_tail_args* a, void* stream);
int mdx_op_edge_tail_bwd(const mdx_edge_tail_bwd_args* a, void* stream);
int mdx_op_edge_tail_lnp_floats(void);
int mdx_op_bondffn_fwd(const mdx_bondffn_args* a, void* stream);
int mdx_op_bondffn_bwd(const mdx_bondffn_bwd_args* a, void* stream);
int mdx_op_bondffn_workgroups(void);
int mdx_op_bondffn_lnp_floats(void);

/* One optimisation step with torch.cuda.amp.GradScaler semantics and ALL of its state on the device (no host round trip):
 * g = gradient of (S x loss).  state (16 floats): [0] loss scale S, [1] growth tracker, [2] optimizer steps taken, [3] steps skipped,
 * [4] unscaled squared gradient norm of this step (inf / nan if a gradient overflowed); [5..8] internal.  Finite: clip to max_norm
 * (<= 0: none), AdamW with t = ++state[2]; after growth_interval consecutive finite steps S *= growth.  Not finite: no update, the
 * step count does not advance, S *= backoff.  S = growth = backoff = 1 is the plain fp32 step.  ws: 1024 floats. */
int mdx_op_amp_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, float max_norm, float* state, float growth, float backoff, int32_t growth_interval, float* ws,
                     void* stream);

/* ---- measurement hooks (bench.py): hipEvent timing of the block kernels on their launch stream.
 * kernel: 0 = fused edge kernel A (MFMA), 1 = fused edge kernel B, 2 = node kernel, 3 = message aggregation
 * (segment sum (E,256)->(N,256), the HBM-bound scatter/gather pass), 4 = the guidance backward's fused edge kernel
 * (edge_bwd2_kernel, MFMA).  enable(0) = off, enable(1) = every kernel, enable(m > 1) = kernel k iff bit (k+1) of m is set
 * (an event pair costs ~3.4 us of stream time, 25 pairs per step: bench.py times only the roofline kernel inside its timed
 * region).  read() drains pending events. */
int mdx_profile_enable(int32_t on);
int mdx_profile_read(int32_t kernel, int64_t* count, double* total_ms);
/* the kernel function name slot `kernel` brackets in this build ("" for an unknown slot); static storage */
const char* mdx_profile_kernel_name(int32_t kernel);

#ifdef __cplusplus
}
#endif
#endif /* MOLDIFF_HIP_H */
