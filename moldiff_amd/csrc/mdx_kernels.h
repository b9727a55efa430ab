// Kernel parameter blocks + launch prototypes shared by the host API (mdx_api.cpp) and the device
// code (*.hip).  All pointers are device pointers.  "internal edge order" = directed edges stably
// sorted by (left, right); see mdx_graph.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define MDX_ND 256  // node feature width (shipped configs; other widths are rejected at create)
#define MDX_ED 64   // edge feature width
#define MDX_NG 16   // distance gaussians
#ifndef MDX_ET
#define MDX_ET 3    // edge tile = 16*MDX_ET rows per workgroup
#endif
#ifndef MDX_EWPS
#define MDX_EWPS 2  // waves per SIMD the edge kernels are compiled for (= workgroups per CU)
#endif
#ifndef MDX_NT
#define MDX_NT 1    // node tile = 16*MDX_NT rows per workgroup (N/16 tiles fill 256 CUs better than N/32)
#endif

// Column layout of the per-node table NT (N, MDX_NTW) written by the node kernel (PRE stage) and
// gathered by the edge kernels:  everything that is Linear(h_node)[idx] in the reference is hoisted
// to one per-node GEMM (exact for a plain Linear; for the first layer of the gate MLPs it splits the
// 321-wide contraction into edge + node + time parts, see DESIGN.md "hoisting").
#define MDX_NT_C 0       // centroid_lin(x) + bias                         (256)
#define MDX_NT_GX 256    // gate.net.0.weight[:, 64:320] x   (no bias)      (256)
#define MDX_NT_NLL 512   // bond_ffn_left.node_linear x                     (128)
#define MDX_NT_NLR 640   // bond_ffn_right.node_linear x                    (128)
#define MDX_NT_NFL 768   // node_ffn_left(x) + bias                         (64)
#define MDX_NT_NFR 832   // node_ffn_right(x) + bias                        (64)
#define MDX_NT_GXL 896   // bond_ffn_left.gate.net.0.weight[:, 64:320] x    (32)
#define MDX_NT_GXR 928   // bond_ffn_right.gate.net.0.weight[:, 64:320] x   (32)
#define MDX_NTW 960

struct MlpW {  // Linear -> LN -> ReLU -> Linear, all packed for gemm_tile
  const float *W1, *b1, *g, *be, *W2, *b2;
};

struct FfnW {  // BondFFN of the EdgeBlock (bond 64, node 256 hoisted, inter 128, gate hidden 32, out 64)
  const float* Wbl;            // bond_linear (128 x 64), no bias
  MlpW inter;                  // 128 -> 128 -> 64
  const float *Wg1e, *bg1, *wtg1, *gg, *gb;  // gate first layer: edge part (32 x 64), bias, time column, LN
  const float *Wg2, *bg2;      // gate second layer (64 x 32)
};

// "stream packs" of the row-owner kernels (mdx_row.h, mdx_edge2.hip): the same matrices in consumption order
struct FfnS {
  const float *Wbl, *Wg1e, *W1, *W2, *Wg2;
};
struct EdgeAS {
  const float *Wemb, *Wg1e, *Wg2, *W1, *W2, *Wm;
  FfnS ffn[2];
};
struct EdgeBS {
  const float *Wself, *Wout, *Wbl, *Wnl, *Wg1h, *Wg1a, *Wi1;
};

struct EdgeAW {  // weights of edge kernel A for one block
  const float *Wemb, *bemb;                        // edge_embs (64 x 80)
  const float *Wg1e, *bg1, *wtg1, *gg, *gb, *Wg2, *bg2;  // NodeBlock gate: edge part (256x64), bias, time col, LN, 256x256
  MlpW en;                                         // edge_net 64 -> 256 -> 256
  const float *Wm, *bm;                            // msg_net 256 x 256
  FfnW ffn[2];                                     // left, right
  EdgeAS s;
  EdgeAS ss;  // the same streams split into float16 hi / lo halves (mdx_split.h; EA_SPLIT launches)
};

struct EdgeBW {  // weights of edge kernel B for one block
  const float *Wself, *bself, *lng, *lnb, *Wout, *bout;  // EdgeBlock tail
  // PosUpdate.edge_lin (BondFFN bond 64, node 64, inter 256, out 1)
  const float *Wbl, *Wnl;          // 256 x 64 each, no bias
  const float *Wi1, *bi1, *ig, *ib;  // inter_module first layer 256x256 + LN
  const float *wi2;                // (256) second layer row
  float bi2;
  const float *Wg1h, *Wg1a, *bg1, *wtg1, *gg, *gb;  // gate first layer split: He part (32x64), a part (32x64), bias, time col, LN
  const float *wg2;                // (32)
  float bg2;
  EdgeBS s;
  EdgeBS ss;  // split float16 streams (EB_SPLIT launches)
};

struct NodeW {  // weights of the node kernel
  // MID stage (block i): NodeBlock tail + PosUpdate per-node MLPs
  const float *lng, *lnb, *Wout, *bout;
  MlpW left, right;  // 256 -> 64 -> 64
  // PRE stage (block i or i+1)
  MlpW nn;                     // node_net 256 -> 256 -> 256
  const float *Wcat, *bcat;    // concatenated (960 x 256) + bias (960, zeros where the reference has none)
};

struct NodeWS {  // the node kernel's matrices as dense split float16 packs (mdx_node_s.hip, PackCtx::pack_dense_split)
  const float *Wout, *leftW1, *leftW2, *rightW1, *rightW2;  // MID stage
  const float *nnW1, *nnW2, *Wcat;                          // PRE stage
};

struct EdgeAArgs {
  int E, flags;
  const int *l, *r;       // internal order
  const float* te;        // per-edge time t/T (internal order): edge_time of EdgeBlock / PosUpdate
  const float* tn_r;      // optional (E): node_time[right] seen by the NodeBlock gate (graph.py:46); nullptr = te
  const float* pos;       // (N,3)
  const float* dist_in;   // optional (E) internal order: use instead of |pos[l]-pos[r]| (unused in product path)
  const float *soff, *scoef;  // distance smearing tables (16)
  float cutoff, smear_start;  // GaussianSmearing clamp [smear_start, cutoff] (common.py:233-235)
  const float* He_in;     // (E,64)
  float* He_out;          // (E,64) = edge_embs([He|D])   (== He_in when !EA_EMB)
  const float* H;         // (N,256) node_net(x)
  const float* NT;        // (N,960)
  float* M;               // (E,256) gated messages
  float* F[2];            // (E,64) bond_ffn_left / right outputs
  float *tSG, *tHE;       // optional tape for the guidance backward: sigmoid(gate) and edge_net output, (E,256) each
  // optional tape of the BondFFNs (round 3; both kernels are bound by the matrix pipe, not by HBM, so the backward reads these back
  // instead of recomputing three GEMMs per side): W_bl He' (E,128), the inter MLP's pre-LayerNorm activation (E,128), its output
  // before the gate (E,64); [0] = left, [1] = right
  float *tBL[2], *tH1[2], *tO[2];
  // EA_AGG (round 3): the segment sums over each left node's edge run happen INSIDE the kernel.  Units are aligned to each
  // graph's first edge (units[2u] = first edge, units[2u+1] = rows of unit u, <= 16) so that where a node's run is cut -- and with
  // it the association of its sum -- depends on the molecule only, never on its position in the batch.  A unit emits one partial
  // row per left node it touches: row epo[e] + u of P (256 wide: gated messages) and of PR (64 wide: BondFFN-right outputs),
  // u = the unit index; a node's pieces are consecutive rows, combined in order by seg_reduce_block2.  M / F[1] are then only
  // written when non-null (the guidance tape wants M).
  const int* units;
  int nunits;
  const int* epo;         // (E) partial-row offset of each edge's left node: row = epo[e] + unit
  float *P, *PR;
  int* wq;                 // work-queue counters of the launch's stream (mdx_row.h WorkQ), nullptr = static unit split
  EdgeAW w;
};
#define EA_EMB 1
#define EA_NODE 2
#define EA_FFN 4
#define EA_AGG 8
#define EA_TAPE_FFN 32  // + the BondFFN intermediates (tBL / tH1 / tO all non-null); implies EA_TAPE
#define EA_SPLIT 64  // matrix products on the split float16 path (mdx_edge2s.hip); not a section flag
#define EA_TAPE 16  // the launch writes the guidance tape (tSG, tHE, M, F[1] with EA_AGG, tBL / tH1 / tO): a template flag of the row-owner kernel

struct EdgeBArgs {
  int E, flags;
  const int *l, *r;
  const float* te;
  const float* pos;        // (N,3) positions at block start (rel/dist source)
  const float *rel_in, *dist_in;  // optional explicit (E,3)/(E) internal order (per-function API)
  const float* Hep;        // (E,64) He' (edge_embs output)
  const float *SL, *SR;    // (N,64) reduced FFN messages
  const float* NT;         // (N,960) (nfl, nfr columns)
  float* He_out;           // (E,64): He' + EdgeBlock(...)  (or just EdgeBlock(...) when EB_DELTA)
  const float *Lf, *Rf;    // (N,64) PosUpdate per-node MLP outputs
  float* Fe;               // (E,3) per-edge force
  int* wq;                 // work-queue counters of the launch's stream (mdx_row.h WorkQ), nullptr = static unit split
  EdgeBW w;
};
#define EB_EDGE 1   // run the EdgeBlock tail
#define EB_POS 2    // run PosUpdate
#define EB_DELTA 4  // He_out = delta only (per-function EdgeBlock API)
#define EB_SPLIT 8  // matrix products on the split float16 path (mdx_edge2bs.hip)

struct NodeArgs {
  int N, flags;
  float* Hn;             // (N,256) in/out (updated in place by MID)
  const float* aggr;     // (N,256) segment-summed messages
  const float* NTin;     // (N,960) table of the block being finished (centroid column)
  float* dHn;            // optional (N,256): write out_transform(...) here instead of updating Hn (ND_DELTA)
  float *Lf, *Rf;        // (N,64)
  float* H;              // (N,256) out: node_net(x) for the next block
  float* NT;             // (N,960) out
  // fused reduction (round 3, P != nullptr): the MID stage sums each node's partial rows of P itself instead of reading `aggr`, and the
  // workgroup also writes its 16 nodes' SR (partial rows of PR) and SL (indexed sum over FL) -- what seg_reduce_block2_kernel did in a
  // launch of its own between edge kernel A and this kernel
  const float *P, *PR, *FL;
  const int *pbase, *col_ptr, *col_eids;
  float *SL, *SR;
  float* aggr_out;       // optional (N,256), fused reduction only: the summed messages, kept for the guidance tape
  NodeW wmid, wpre;
  NodeWS smid, spre;     // split float16 packs of the same matrices (ND_SPLIT launches)
};
#define ND_MID 1
#define ND_PRE 2
#define ND_DELTA 4
#define ND_POSMLP 8
#define ND_SPLIT 16   // matrix products on the split float16 path (mdx_node_s.hip)

// ---- backward (bond-predictor guidance gradient; dgrad only, no weight gradients) ---------------------
struct FfnWT {
  const float *WblT, *Wi1T, *Wi2T, *Wg1eT, *Wg2T;
};
struct FfnTS {  // stream packs (mdx_row.h) of the transposed BondFFN matrices
  const float *WblT, *Wi1T, *Wi2T, *Wg1eT, *Wg2T;
};
struct EdgeBwdS {
  const float *WembHT, *WembDT, *Wg1eT, *Wg2T, *W1T, *W2T, *WmT;
  FfnTS ffn[2];
  const float *WselfT, *WoutT;  // EdgeBlock tail
};
struct EdgeBwdW {  // transposed packs (contraction over the forward's output features)
  const float *WembHT, *WembDT;                 // edge_embs^T: -> 64 (He part), -> 16 (distance part)
  const float *Wg1eT, *Wg2T, *W1T, *W2T, *WmT;  // NodeBlock gate / edge_net / msg_net
  FfnWT ffn[2];
  const float *WselfT, *WoutT;                  // EdgeBlock tail
  EdgeBwdS s;                                   // row-owner kernel (mdx_bwd2.hip)
  EdgeBwdS ss;                                  // the same as split float16 streams (mdx_bwd2s.hip)
};
struct NodeBwdW {
  const float* WoutT;      // NodeBlock out_transform^T
  const float *W1T, *W2T;  // node_net^T
  const float* WcatT[4];   // (960 x 256)^T in 4 K-chunks: 256, 256, 256, 192
};

struct EdgeTailBwdArgs {
  int E;
  const int *l, *r;
  const float* te;
  const float *Hep, *gHe;      // (E,64): He'_i (tape), dL/dHe_{i+1}
  const float *SL, *SR, *NT;   // tape
  float *GU, *GHEP;            // (E,64): dL/du ; gHe + self_ffn^T dL/du
  int* wq;                     // see EdgeAArgs
  EdgeBW w;
  const float *WselfT, *WoutT;
  const float *sWselfT, *sWoutT;  // the same as stream packs (row-owner kernel, mdx_bwd2.hip)
  const float *ssWselfT, *ssWoutT;  // ... and as split float16 stream packs (mdx_bwd2s.hip)
  int split;                       // 1: run the split float16 build
};

struct EdgeBwdArgs {
  int E;
  const int *l, *r;
  const float* te;
  const float* pos;
  const float *soff, *scoef;
  float cutoff, smear_start;  // GaussianSmearing clamp [smear_start, cutoff] (common.py:233-235)
  const float *Hep, *GHEP;     // (E,64)
  const float *H, *NT;         // tape node tables of this block
  const float *SG, *HE, *M;    // tape (E,256): sigmoid(gate), edge_net output, gated message (M = msg_net(he*h[r]) * SG)
  const float *BL[2], *H1[2], *O[2];  // BondFFN tape (EdgeAArgs tBL / tH1 / tO); all null = recompute (tile kernels, A/B)
  const float* GNT;            // (N,960) gradient table: C cols = dL/d(aggr), NFL cols = A_l, NFR cols = A_r
  float* gdist;                // (E) accumulated over blocks
  // Gradient payloads for the node tables.  The kernel walks the edges in BY-RIGHT order (units of 16 positions of col_eids,
  // aligned to each graph's first position like the forward's by-left units), so every payload that is reduced by the right end
  // point -- GH, GGX, GNL[1], GGXS[1]: 672 of the 832 floats per edge -- is summed over each right node's run INSIDE the kernel
  // (seg_sum_store, mdx_row.h) and leaves as one PARTIAL ROW per (node, unit) at row epo_r[j] + unit; the by-left payloads
  // GNL[0], GGXS[0] stay per edge.  seg_reduce_bwd_block_kernel adds a node's ~2.5 partial rows in order (pbase_r).
  float *GH, *GGX;             // (nparts_r,256) partial rows
  float* GNL[2];               // [0]: (E,128) per edge, reduced by left afterwards; [1]: (nparts_r,128) partial rows
  float* GGXS[2];              // [0]: (E,32) per edge; [1]: (nparts_r,32) partial rows
  const int *units_r, *epo_r, *col_eids, *col_left, *col_right;  // by-right plan (mdx_graph_s)
  int nunits_r;
  // EdgeBlock-tail backward of the PREVIOUS block (i - 1), fused behind this block's edge_embs backward for the same rows
  // (round 5; it used to be a launch of its own that re-read dL/dHe'' from memory): tape and weights of block i - 1, outputs
  // tGU (E,64) = dL/du and the rows of GHEP for block i - 1 (written over the rows of GHEP this unit has already consumed).
  // fuse_tail = 0 (block 0): none of this is read, and dL/dHe_0 -- the gradient of the embedding, which nobody needs -- is not formed.
  int fuse_tail;
  const float *tHep, *tSL, *tSR, *tNT;
  const float *tWself, *tWoutT, *tWselfT;   // stream packs in the build's format (BW_S)
  const float *tbself, *tlng, *tlnb;
  float *tGU, *tGHEP;
  int* wq;                 // work-queue counters of the launch's stream (mdx_row.h WorkQ), nullptr = static unit split
  int split;               // 1: run the split float16 build (mdx_bwd2s.hip; needs the BondFFN tape)
  EdgeAW w;
  EdgeBwdW wt;
};

struct NodeBwdWS {  // split float16 dense packs of NodeBwdW (mdx_bondpred.hip node_bwd_s_kernel)
  const float *WoutT, *W1T, *W2T, *WcatT[4];
};

struct NodeBwdArgs {
  int N, flags;
  float* gHn;                  // (N,256) in/out
  const float* Hn;             // tape Hn_i
  const float *NTin, *aggr;    // tape NT_i, aggr_i
  float* GNT;                  // (N,960)
  const float* gH;             // (N,256) dL/d node_net(x)
  NodeW w;
  NodeBwdW wt;
  NodeWS ws;       // split packs (NB_SPLIT): node_net's first layer for the recompute ...
  NodeBwdWS wts;   // ... and the transposed matrices
};
#define NB_TAIL 1
#define NB_PRE 2
#define NB_SPLIT 4  // matrix products on the split float16 path

struct BondDecW {
  const float *W1e, *W1n, *b1, *g1, *be1, *W2, *b2, *g2, *be2, *W3, *b3;
  const float *W1eT, *W1nT, *W2T, *W3T;
};
struct BondDecArgs {
  int Eh, Ke;
  const float *He, *Hn;        // (E,64) internal order, (N,256)
  const int *ref2int, *left, *right;
  float* logits;               // fwd out (Eh,Ke)
  const float* glogits;        // bwd in
  float *gHe, *GBN;            // bwd out: (E,64) internal order, (Eh,256)
  BondDecW w;
};

void launch_edge_bwd2(const EdgeBwdArgs& a, hipStream_t s);  // row-owner guidance backward (mdx_bwd2.hip)
void launch_edge_tail_bwd2(const EdgeTailBwdArgs& a, hipStream_t s);
int launch_edge_bwd2s(const EdgeBwdArgs& a, hipStream_t s);       // split float16 builds (mdx_bwd2s.hip); nonzero = tape missing
void launch_edge_tail_bwd2s(const EdgeTailBwdArgs& a, hipStream_t s);
void launch_node_bwd(const NodeBwdArgs& a, hipStream_t s);
void launch_bond_decode(const BondDecArgs& a, bool backward, hipStream_t s);
// generalized segment sum: C in {32,64,128,256}; out row stride out_ld (floats), column offset already applied to `out`
struct SegBwdArgs {  // the payload reductions after the backward edge kernel (mdx_bondpred.hip)
  int N;
  const int *row_ptr, *col_ptr, *col_eids;
  const int* pbase_r;  // partial-row ranges of the by-right payloads (GH, GGX, GNL1, GGXS1 hold partial rows)
  const float *GH, *GGX, *GNL0, *GNL1, *GGXS0, *GGXS1;
  float *gH, *GNT;
};
void launch_seg_reduce_bwd_block(const SegBwdArgs& a, hipStream_t s);
void launch_seg_reduce_tail_block(const float* GU, const int* row_ptr, const int* col_ptr, const int* col_eids, float* GNT, int N,
                                  hipStream_t s);
void launch_seg_reduce_ld(const float* src, const int* ptr, const int* eids, float* out, int out_ld, int N, int C,
                          hipStream_t s);
// dpos[v] = sum_{l=v} gd_e rel_e/d_e - sum_{r=v} gd_e rel_e/d_e   (two-pass, deterministic)
void launch_dist_to_pos(const float* gdist, const float* pos, const int* l, const int* r, const int* row_ptr,
                        const int* col_ptr, const int* col_eids, float* tmpE3, float* tmpN3, float* gpos, float scale, int N,
                        int E, float cutoff, hipStream_t s);

// all four return MDX_OK or an error code with mdx_last_error set (a section-flag combination that is not built)
int launch_edge_a(const EdgeAArgs& a, hipStream_t s);
int launch_edge_b(const EdgeBArgs& a, hipStream_t s);
// the exact fp32 builds (mdx_edge2.hip, mdx_edge2b.hip); launch_edge_a/b dispatch to them, or to the split builds below
int launch_edge_a2(const EdgeAArgs& a, hipStream_t s);
int launch_edge_b2(const EdgeBArgs& a, hipStream_t s);
// split-precision builds of the two (mdx_edge2s.hip, mdx_edge2bs.hip): taken when the flags carry EA_SPLIT / EB_SPLIT
int launch_edge_a2s(const EdgeAArgs& a, hipStream_t s);
int launch_edge_b2s(const EdgeBArgs& a, hipStream_t s);
int mdx_num_cus();
void launch_node(const NodeArgs& a, hipStream_t s);    // dispatches to launch_node_s when the flags carry ND_SPLIT
void launch_node_s(const NodeArgs& a, hipStream_t s);

// out[v][0..C) (+)= sum_{j in ptr[v]..ptr[v+1]} src[(eids ? eids[j] : j)][0..C)
struct StepTransArgs {  // mdx_transition.hip: the transitions of one sampling step in one launch (Kn = 8, Ke = 6)
  int N, Eh, T;
  const float *c0, *ct, *sd, *node_q, *node_qT1, *edge_q, *edge_qT1;
  const int64_t *t, *batch_node, *batch_half;
  const float *pos, *pred_pos, *eps, *pred_node, *log_node, *u_node, *pred_half, *log_half, *u_half;
  float *pos_next, *log_node_next, *h_node_next, *log_half_next, *h_half_next;
  uint8_t *node_cls, *half_cls;
};
void launch_step_transition(const StepTransArgs& a, hipStream_t s);
// the same three sums after an EA_AGG edge kernel A: aggr / SR combine each node's partial rows pbase[v] .. pbase[v+1] of P / PR
// in order, SL is still the indexed sum over FL
void launch_seg_reduce_block2(const float* P, const float* PR, const float* FL, const int* pbase, const int* col_ptr,
                              const int* col_eids, float* aggr, float* SL, float* SR, int N, hipStream_t s);
void launch_seg_reduce(const float* src, const int* ptr, const int* eids, float* out, const float* addend, int N, int C,
                       hipStream_t s);

struct EmbedArgs {
  int N, E, Kn, Ke, time_dim, T, nd_emb, ed_emb;  // nd_emb = 256 - time_dim, ed_emb = 64 - time_dim
  const float* xn;        // (N,Kn)
  const float* xe;        // (E_ref,Ke) reference edge order (MolDiff)   | nullptr for the bond predictor
  const int* int2ref;     // (E)
  int half_rows;          // > 0: xe holds (half_rows,Ke) half-edge rows, reference row r reads row r mod half_rows (model.py:273)
  const int *l, *r;       // internal
  const int* node_graph;  // (N)
  const int64_t* t;       // (B) device
  int zero_time;          // != 0: the time-free bond predictor (num_timesteps == 0): t is ignored, every row's time is 0 (bond_predictor.py:141-144)
  const float *Wn, *We;   // node_embedder (nd_emb x Kn), edge_embedder (ed_emb x Ke or 2*Kn)
  const float *toff, *tcoef;  // time smearing tables
  float *Hn, *He, *tn, *te;   // outputs: (N,256) (E,64) (N) (E)
};
void launch_embed(const EmbedArgs& a, hipStream_t s);

struct DecodeArgs {
  int N, Eh, Kn, Ke;
  const float* Hn;  // (N,256)
  const float* He;  // (E,64) internal order
  const int* ref2int;  // (E): reference index -> internal index
  MlpW nodedec, edgedec;  // second layers padded to 16 outputs
  float *pred_node, *pred_halfedge;
};
void launch_decode(const DecodeArgs& a, hipStream_t s);

// transitions / noise (mdx_transition.hip)
void launch_gauss_posterior(const float* c0, const float* ct, const float* sd, const float* xt, const float* x0, const float* eps,
                            const int64_t* t, const int64_t* batch, int n, int C, float* out, hipStream_t s);
void launch_pos_posterior(const float* c0, const float* ct, const float* sd, const float* xt, const float* x0,
                          const float* eps, const int64_t* t, const int64_t* batch, int n, float* out, hipStream_t s);
void launch_cat_posterior(const float* qmats, const float* qT1, int K, int T, const float* logits_or_log_v0, int is_logits,
                          const float* log_vt, const int64_t* t, const int64_t* batch, int n, float* out, hipStream_t s);
void launch_uncertainty_grad(const float* logits, int K, int n, float* glogits, hipStream_t s);
void launch_cat_add_noise(const float* qmats, int K, const int64_t* v, const int64_t* t, const int64_t* batch, const float* u, int n,
                          float log_off, float* onehot, float* log_vt, float* log_v0, hipStream_t s);
void launch_cat_loss(const float* qmats, const float* qT1, int K, const float* logits, const float* log_vt, const float* log_v0,
                     const int64_t* t, const int64_t* batch, int n, float* row_loss, float* dlogits, hipStream_t s);
void launch_add_inplace(float* dst, const float* src, int n, hipStream_t s);
void launch_gumbel_argmax(const float* logits, const float* u, int K, int n, int64_t* cls, float* onehot, hipStream_t s,
                          uint8_t* cls8 = nullptr);
void launch_prior_draw(const double* logits64_host, int K, const void* u, bool u_f64, int n, int64_t* cls, float* onehot,
                       float* log_onehot, float log_off, uint8_t* cls8, hipStream_t s);
void launch_fill_i64(int64_t* p, int64_t v, int n, hipStream_t s);
void launch_philox_noise(uint64_t seed, int step, const int* node_graph, const int* node_local, const int* he_graph,
                         const int* he_local, const int64_t* mol_ids, int N, int Eh, int Kn, int Ke, float* eps_pos,
                         float* u_node, float* u_half, hipStream_t s, int64_t* t_buf = nullptr, int64_t t_val = 0,
                         int B = 0);  // t_buf: also fill (B) int64 with t_val
