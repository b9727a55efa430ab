// Host side of libmoldiff_hip.so: handles, weight packing, CSR planning, launch orchestration, C ABI.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/moldiff_hip.h"
#include "mdx_kernels.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int mdx_set_error(int code, const char* msg) { return fail(code, "%s", msg); }  // for the other translation units
#define HIPCHK(x)                                                                           \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) return fail(MDX_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_));   \
  } while (0)

extern "C" const char* mdx_last_error(void) { return g_err; }
extern "C" int mdx_version(void) { return 100; }
extern "C" int mdx_device_count(int* count) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  if (count) *count = n;
  return MDX_OK;
}

// ------------------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------------------
struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

struct BlockW {
  EdgeAW ea;
  EdgeBW eb;
  NodeW nd;  // mid fields = tail of this block, pre fields = pre-stage of this block
  NodeWS nds;  // the same matrices as split float16 packs
};

struct mdx_model_s {
  mdx_config cfg;
  std::map<std::string, HostTensor> params;
  bool finalized = false;
  float smear_start = 0.f;  // lower clamp of the distance smearing (mdx_model_set_smear_start)
  float* arena = nullptr;  // device
  size_t arena_floats = 0;
  std::vector<BlockW> blocks;
  const float *soff = nullptr, *scoef = nullptr;  // distance smearing
  // heads
  const float *Wn = nullptr, *We = nullptr, *toff = nullptr, *tcoef = nullptr;
  MlpW nodedec{}, edgedec{};
  // bond predictor: decoder + transposed packs for the guidance backward
  BondDecW dec{};
  std::vector<EdgeBwdW> ebw;
  std::vector<NodeBwdW> nbw;
  std::vector<NodeBwdWS> nbws;
  // matrix path of the row-owner edge kernels: 0 = exact fp32 MFMA (default), 1 = split float16 (mdx_split.h).  Both weight
  // packs are always built; split_ok is false when a weight would overflow the scaled float16 range (the path is then refused).
  int matrix_path = 0;
  bool split_ok = true;
  float split_wmax = 0.f;
};

namespace {

struct Packer {
  std::vector<float> host;
  std::vector<std::pair<const float**, size_t>> fix;  // pointer slot <- arena offset
  size_t reserve(size_t n) {
    size_t off = (host.size() + 63) & ~size_t(63);
    host.resize(off + n, 0.f);
    return off;
  }
  void bind(const float** slot, size_t off) { fix.emplace_back(slot, off); }
};

struct PackCtx {
  mdx_model_s* m;
  Packer pk;
  std::string missing;
  const HostTensor* get(const std::string& key, std::initializer_list<int64_t> shape) {
    auto it = m->params.find(key);
    if (it == m->params.end()) {
      if (missing.empty()) missing = key;
      return nullptr;
    }
    std::vector<int64_t> s(shape);
    if (it->second.shape != s) {
      if (missing.empty()) missing = key + " (shape mismatch)";
      return nullptr;
    }
    return &it->second;
  }
  // dense row-major (F x ldw) -> fragment order for gemm_tile: float index ((g*FT + ft)*64 + lane)*4 + s
  void pack_dense(const float** slot, const std::vector<float>& W, int F, int ldw, int col0, int K) {
    const int FT = (F + 15) / 16, G = K / 16;
    size_t off = pk.reserve((size_t)G * FT * 256);
    float* o = pk.host.data() + off;
    for (int g = 0; g < G; ++g)
      for (int ft = 0; ft < FT; ++ft)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int f = 16 * ft + (lane & 15), k = 16 * g + 4 * (lane >> 4) + s;
            o[(((size_t)g * FT + ft) * 64 + lane) * 4 + s] = f < F ? W[(size_t)f * ldw + col0 + k] : 0.f;
          }
    pk.bind(slot, off);
  }
  void packA(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    pack_dense(slot, t->data, F, ldw, col0, K);
  }
  // "stream pack" of the row-owner kernels (mdx_row.h rgemm): the same fragments in consumption order --
  // float index ((((ftp*KG + g)*2 + j)*64 + lane)*4 + s) <- W[16*(2 ftp + j) + (lane & 15)][col0 + 16 g + 4 (lane >> 4) + s];
  // F is padded to a multiple of 32 with zero rows, plus MDX_RING steps of zero tail so a ring may over-fetch
  void pack_stream(const float** slot, const std::vector<float>& W, int F, int ldw, int col0, int K) {
    const int FTP = (F + 31) / 32, KG = K / 16;
    size_t off = pk.reserve((size_t)FTP * KG * 512 + 4 * 512);  // + MDX_RING_PAD zero steps
    float* o = pk.host.data() + off;
    for (int ftp = 0; ftp < FTP; ++ftp)
      for (int g = 0; g < KG; ++g)
        for (int j = 0; j < 2; ++j)
          for (int lane = 0; lane < 64; ++lane)
            for (int s = 0; s < 4; ++s) {
              const int f = 16 * (2 * ftp + j) + (lane & 15), k = 16 * g + 4 * (lane >> 4) + s;
              o[((((size_t)ftp * KG + g) * 2 + j) * 64 + lane) * 4 + s] = f < F ? W[(size_t)f * ldw + col0 + k] : 0.f;
            }
    pk.bind(slot, off);
  }
  void packS(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    pack_stream(slot, t->data, F, ldw, col0, K);
  }
  // split float16 stream pack (mdx_split.h rgemm_s): W = hi + lo 2^-11 with hi = fp16(W), lo = fp16((W - hi) 2^11); half-step hs = (ftp*KG + g)*2 + h carries
  // the fragments of feature tiles 2 ftp + j, j = 0,1 (h = 0: hi, h = 1: lo), 64 lanes x 8 halves each:
  // half index ((hs*2 + j)*64 + lane)*8 + t <- W[16 (2 ftp + j) + (lane & 15)][col0 + 32 g + 16 (t / 4) + 4 (lane >> 4) + t % 4];
  // F padded to 32 and K to 32 with zeros, plus 4 zero half-steps so a ring may over-fetch.  Same bytes as pack_stream for K % 32 == 0.
  void pack_stream_split(const float** slot, const std::vector<float>& W, int F, int ldw, int col0, int K) {
    const int FTP = (F + 31) / 32, KG = (K + 31) / 32;
    size_t off = pk.reserve((size_t)FTP * KG * 2 * 512 + 4 * 512);
    const float lo_up = 2048.0f;  // 2^MDX_LO_SHIFT
    std::vector<uint16_t> hbuf((size_t)FTP * KG * 2 * 1024, 0);
    for (int ftp = 0; ftp < FTP; ++ftp)
      for (int g = 0; g < KG; ++g)
        for (int j = 0; j < 2; ++j)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 8; ++t) {
              const int f = 16 * (2 * ftp + j) + (lane & 15), k = 32 * g + 16 * (t / 4) + 4 * (lane >> 4) + t % 4;
              const float w = (f < F && k < K) ? W[(size_t)f * ldw + col0 + k] : 0.f;
              if (!(std::fabs(w) < 65504.0f)) m->split_ok = false;
              m->split_wmax = std::max(m->split_wmax, std::fabs(w));
              const _Float16 hi = (_Float16)w;
              const _Float16 lo = (_Float16)((w - (float)hi) * lo_up);
              uint16_t uh, ul;
              std::memcpy(&uh, &hi, 2);
              std::memcpy(&ul, &lo, 2);
              const size_t hs = ((size_t)ftp * KG + g) * 2;
              hbuf[((hs * 2 + j) * 64 + lane) * 8 + t] = uh;
              hbuf[(((hs + 1) * 2 + j) * 64 + lane) * 8 + t] = ul;
            }
    std::memcpy(pk.host.data() + off, hbuf.data(), hbuf.size() * 2);
    pk.bind(slot, off);
  }
  // dense split pack of the tile kernels (mdx_node_s.hip gemm_tile_s): 1-KiB fragment index (g*FT + ft)*2 + h, g = k/32, h = 0 hi /
  // 1 lo (x 2^11); lane (q, c) half t <- W[16 ft + c][col0 + 32 g + 16 (t / 4) + 4 q + t % 4]
  void pack_dense_split(const float** slot, const std::vector<float>& W, int F, int ldw, int col0, int K) {
    const int FT = (F + 15) / 16, G = (K + 31) / 32;
    size_t off = pk.reserve((size_t)G * FT * 512);
    std::vector<uint16_t> hbuf((size_t)G * FT * 1024, 0);
    for (int g = 0; g < G; ++g)
      for (int ft = 0; ft < FT; ++ft)
        for (int lane = 0; lane < 64; ++lane)
          for (int t = 0; t < 8; ++t) {
            const int f = 16 * ft + (lane & 15), k = 32 * g + 16 * (t / 4) + 4 * (lane >> 4) + t % 4;
            const float w = (f < F && k < K) ? W[(size_t)f * ldw + col0 + k] : 0.f;
            if (!(std::fabs(w) < 65504.0f)) m->split_ok = false;
            m->split_wmax = std::max(m->split_wmax, std::fabs(w));
            const _Float16 hi = (_Float16)w;
            const _Float16 lo = (_Float16)((w - (float)hi) * 2048.0f);
            uint16_t uh, ul;
            std::memcpy(&uh, &hi, 2);
            std::memcpy(&ul, &lo, 2);
            const size_t fr = ((size_t)g * FT + ft) * 2;
            hbuf[(fr * 64 + lane) * 8 + t] = uh;
            hbuf[((fr + 1) * 64 + lane) * 8 + t] = ul;
          }
    std::memcpy(pk.host.data() + off, hbuf.data(), hbuf.size() * 2);
    pk.bind(slot, off);
  }
  void packDS(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    pack_dense_split(slot, t->data, F, ldw, col0, K);
  }
  void packSS(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    pack_stream_split(slot, t->data, F, ldw, col0, K);
  }
  // transpose pack: out[k][f] = W[f][col0 + k]  (contraction over the forward's output features, zero padded to 16)
  void packT(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    const int Fp = (F + 15) / 16 * 16;
    std::vector<float> Wt((size_t)K * Fp, 0.f);
    for (int k = 0; k < K; ++k)
      for (int f = 0; f < F; ++f) Wt[(size_t)k * Fp + f] = t->data[(size_t)f * ldw + col0 + k];
    pack_dense(slot, Wt, K, Fp, 0, Fp);
  }
  // transposed stream pack: out[k][f] = W[f][col0 + k] in rgemm consumption order
  void packTS(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    const int Fp = (F + 15) / 16 * 16;
    std::vector<float> Wt((size_t)K * Fp, 0.f);
    for (int k = 0; k < K; ++k)
      for (int f = 0; f < F; ++f) Wt[(size_t)k * Fp + f] = t->data[(size_t)f * ldw + col0 + k];
    pack_stream(slot, Wt, K, Fp, 0, Fp);
  }
  // transposed dense split pack (node_bwd_s_kernel)
  void packTDS(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    const int Fp = (F + 15) / 16 * 16;
    std::vector<float> Wt((size_t)K * Fp, 0.f);
    for (int k = 0; k < K; ++k)
      for (int f = 0; f < F; ++f) Wt[(size_t)k * Fp + f] = t->data[(size_t)f * ldw + col0 + k];
    pack_dense_split(slot, Wt, K, Fp, 0, Fp);
  }
  // transposed split stream pack (mdx_bwd2s.hip)
  void packTSS(const float** slot, const std::string& key, int F, int ldw, int col0, int K) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    const int Fp = (F + 15) / 16 * 16;
    std::vector<float> Wt((size_t)K * Fp, 0.f);
    for (int k = 0; k < K; ++k)
      for (int f = 0; f < F; ++f) Wt[(size_t)k * Fp + f] = t->data[(size_t)f * ldw + col0 + k];
    pack_stream_split(slot, Wt, K, Fp, 0, Fp);
  }
  std::map<int, std::vector<float>> wcat_dense;  // block -> dense (960 x 256) concatenated node-table weights
  void vec(const float** slot, const std::string& key, int n, int pad_to = 0) {
    const HostTensor* t = get(key, {n});
    if (!t) return;
    size_t off = pk.reserve(std::max(n, pad_to));
    std::copy(t->data.begin(), t->data.end(), pk.host.begin() + off);
    pk.bind(slot, off);
  }
  void raw(const float** slot, const std::vector<float>& v) {
    size_t off = pk.reserve(v.size());
    std::copy(v.begin(), v.end(), pk.host.begin() + off);
    pk.bind(slot, off);
  }
  void column(const float** slot, const std::string& key, int F, int ldw, int col) {
    const HostTensor* t = get(key, {F, ldw});
    if (!t) return;
    std::vector<float> v(F);
    for (int f = 0; f < F; ++f) v[f] = t->data[(size_t)f * ldw + col];
    raw(slot, v);
  }
  void mlp(MlpW* w, const std::string& pre, int in, int hid, int out, int out_pad = 0) {
    packA(&w->W1, pre + ".net.0.weight", hid, in, 0, in);
    vec(&w->b1, pre + ".net.0.bias", hid);
    vec(&w->g, pre + ".net.1.weight", hid);
    vec(&w->be, pre + ".net.1.bias", hid);
    packA(&w->W2, pre + ".net.3.weight", out, hid, 0, hid);
    vec(&w->b2, pre + ".net.3.bias", out, out_pad);
  }
};

int pack_model(mdx_model_s* m) {
  PackCtx c{m};
  m->split_ok = true;
  m->split_wmax = 0.f;
  const mdx_config& cf = m->cfg;
  const int ND = MDX_ND, ED = MDX_ED, GIN = ED + ND + 1;  // 321
  const std::string net = cf.kind == MDX_KIND_MOLDIFF ? "denoiser." : cf.kind == MDX_KIND_BONDPRED ? "encoder." : "";
  m->blocks.assign(cf.num_blocks, BlockW{});
  c.vec(&m->soff, net + "distance_expansion.offset", MDX_NG);
  c.vec(&m->scoef, net + "distance_expansion.coeff", MDX_NG);
  std::vector<float> scalars_bi2(cf.num_blocks, 0.f), scalars_bg2(cf.num_blocks, 0.f);
  for (int i = 0; i < cf.num_blocks; ++i) {
    BlockW& b = m->blocks[i];
    const std::string si = std::to_string(i);
    const std::string nb = net + "node_blocks_with_edge." + si, eb = net + "edge_blocks." + si, pb = net + "pos_blocks." + si;
    // ---- edge kernel A
    c.packA(&b.ea.Wemb, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, 0, ED + MDX_NG);
    c.vec(&b.ea.bemb, net + "edge_embs." + si + ".bias", ED);
    c.packA(&b.ea.Wg1e, nb + ".gate.net.0.weight", ND, GIN, 0, ED);
    c.vec(&b.ea.bg1, nb + ".gate.net.0.bias", ND);
    c.column(&b.ea.wtg1, nb + ".gate.net.0.weight", ND, GIN, GIN - 1);
    c.vec(&b.ea.gg, nb + ".gate.net.1.weight", ND);
    c.vec(&b.ea.gb, nb + ".gate.net.1.bias", ND);
    c.packA(&b.ea.Wg2, nb + ".gate.net.3.weight", ND, ND, 0, ND);
    c.vec(&b.ea.bg2, nb + ".gate.net.3.bias", ND);
    c.mlp(&b.ea.en, nb + ".edge_net", ED, ND, ND);
    c.packA(&b.ea.Wm, nb + ".msg_net.weight", ND, ND, 0, ND);
    c.vec(&b.ea.bm, nb + ".msg_net.bias", ND);
    c.packS(&b.ea.s.Wemb, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, 0, ED + MDX_NG);
    c.packS(&b.ea.s.Wg1e, nb + ".gate.net.0.weight", ND, GIN, 0, ED);
    c.packS(&b.ea.s.Wg2, nb + ".gate.net.3.weight", ND, ND, 0, ND);
    c.packS(&b.ea.s.W1, nb + ".edge_net.net.0.weight", ND, ED, 0, ED);
    c.packS(&b.ea.s.W2, nb + ".edge_net.net.3.weight", ND, ND, 0, ND);
    c.packS(&b.ea.s.Wm, nb + ".msg_net.weight", ND, ND, 0, ND);
    c.packSS(&b.ea.ss.Wemb, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, 0, ED + MDX_NG);
    c.packSS(&b.ea.ss.Wg1e, nb + ".gate.net.0.weight", ND, GIN, 0, ED);
    c.packSS(&b.ea.ss.Wg2, nb + ".gate.net.3.weight", ND, ND, 0, ND);
    c.packSS(&b.ea.ss.W1, nb + ".edge_net.net.0.weight", ND, ED, 0, ED);
    c.packSS(&b.ea.ss.W2, nb + ".edge_net.net.3.weight", ND, ND, 0, ND);
    c.packSS(&b.ea.ss.Wm, nb + ".msg_net.weight", ND, ND, 0, ND);
    for (int s = 0; s < 2; ++s) {
      FfnW& f = b.ea.ffn[s];
      const std::string fp = eb + (s ? ".bond_ffn_right" : ".bond_ffn_left");
      c.packA(&f.Wbl, fp + ".bond_linear.weight", 2 * ED, ED, 0, ED);
      c.mlp(&f.inter, fp + ".inter_module", 2 * ED, 2 * ED, ED);
      c.packA(&f.Wg1e, fp + ".gate.net.0.weight", 32, GIN, 0, ED);
      c.vec(&f.bg1, fp + ".gate.net.0.bias", 32);
      c.column(&f.wtg1, fp + ".gate.net.0.weight", 32, GIN, GIN - 1);
      c.vec(&f.gg, fp + ".gate.net.1.weight", 32);
      c.vec(&f.gb, fp + ".gate.net.1.bias", 32);
      c.packA(&f.Wg2, fp + ".gate.net.3.weight", ED, 32, 0, 32);
      c.vec(&f.bg2, fp + ".gate.net.3.bias", ED);
      FfnS& fs = b.ea.s.ffn[s];
      c.packS(&fs.Wbl, fp + ".bond_linear.weight", 2 * ED, ED, 0, ED);
      c.packS(&fs.Wg1e, fp + ".gate.net.0.weight", 32, GIN, 0, ED);
      c.packS(&fs.W1, fp + ".inter_module.net.0.weight", 2 * ED, 2 * ED, 0, 2 * ED);
      c.packS(&fs.W2, fp + ".inter_module.net.3.weight", ED, 2 * ED, 0, 2 * ED);
      c.packS(&fs.Wg2, fp + ".gate.net.3.weight", ED, 32, 0, 32);
      FfnS& fss = b.ea.ss.ffn[s];
      c.packSS(&fss.Wbl, fp + ".bond_linear.weight", 2 * ED, ED, 0, ED);
      c.packSS(&fss.Wg1e, fp + ".gate.net.0.weight", 32, GIN, 0, ED);
      c.packSS(&fss.W1, fp + ".inter_module.net.0.weight", 2 * ED, 2 * ED, 0, 2 * ED);
      c.packSS(&fss.W2, fp + ".inter_module.net.3.weight", ED, 2 * ED, 0, 2 * ED);
      c.packSS(&fss.Wg2, fp + ".gate.net.3.weight", ED, 32, 0, 32);
    }
    // ---- edge kernel B
    c.packA(&b.eb.Wself, eb + ".self_ffn.weight", ED, ED, 0, ED);
    c.vec(&b.eb.bself, eb + ".self_ffn.bias", ED);
    c.vec(&b.eb.lng, eb + ".layer_norm.weight", ED);
    c.vec(&b.eb.lnb, eb + ".layer_norm.bias", ED);
    c.packA(&b.eb.Wout, eb + ".out_transform.weight", ED, ED, 0, ED);
    c.vec(&b.eb.bout, eb + ".out_transform.bias", ED);
    c.packS(&b.eb.s.Wself, eb + ".self_ffn.weight", ED, ED, 0, ED);
    c.packS(&b.eb.s.Wout, eb + ".out_transform.weight", ED, ED, 0, ED);
    c.packSS(&b.eb.ss.Wself, eb + ".self_ffn.weight", ED, ED, 0, ED);
    c.packSS(&b.eb.ss.Wout, eb + ".out_transform.weight", ED, ED, 0, ED);
    // ---- node kernel
    c.vec(&b.nd.lng, nb + ".layer_norm.weight", ND);
    c.vec(&b.nd.lnb, nb + ".layer_norm.bias", ND);
    c.packA(&b.nd.Wout, nb + ".out_transform.weight", ND, ND, 0, ND);
    c.vec(&b.nd.bout, nb + ".out_transform.bias", ND);
    c.mlp(&b.nd.nn, nb + ".node_net", ND, ND, ND);
    c.packDS(&b.nds.Wout, nb + ".out_transform.weight", ND, ND, 0, ND);
    c.packDS(&b.nds.nnW1, nb + ".node_net.net.0.weight", ND, ND, 0, ND);
    c.packDS(&b.nds.nnW2, nb + ".node_net.net.3.weight", ND, ND, 0, ND);
    {  // concatenated per-node table weights (960 x 256) + bias
      std::vector<float> W((size_t)MDX_NTW * ND, 0.f), bias(MDX_NTW, 0.f);
      auto put = [&](const std::string& key, int rows, int ldw, int col0, int dst_row, const std::string& bkey) {
        const HostTensor* t = c.get(key, {rows, ldw});
        if (!t) return;
        for (int r = 0; r < rows; ++r)
          for (int k = 0; k < ND; ++k) W[(size_t)(dst_row + r) * ND + k] = t->data[(size_t)r * ldw + col0 + k];
        if (!bkey.empty()) {
          const HostTensor* bt = c.get(bkey, {rows});
          if (bt) std::copy(bt->data.begin(), bt->data.end(), bias.begin() + dst_row);
        }
      };
      put(nb + ".centroid_lin.weight", ND, ND, 0, MDX_NT_C, nb + ".centroid_lin.bias");
      put(nb + ".gate.net.0.weight", ND, GIN, ED, MDX_NT_GX, "");
      put(eb + ".bond_ffn_left.node_linear.weight", 2 * ED, ND, 0, MDX_NT_NLL, "");
      put(eb + ".bond_ffn_right.node_linear.weight", 2 * ED, ND, 0, MDX_NT_NLR, "");
      put(eb + ".node_ffn_left.weight", ED, ND, 0, MDX_NT_NFL, eb + ".node_ffn_left.bias");
      put(eb + ".node_ffn_right.weight", ED, ND, 0, MDX_NT_NFR, eb + ".node_ffn_right.bias");
      put(eb + ".bond_ffn_left.gate.net.0.weight", 32, GIN, ED, MDX_NT_GXL, "");
      put(eb + ".bond_ffn_right.gate.net.0.weight", 32, GIN, ED, MDX_NT_GXR, "");
      c.pack_dense(&b.nd.Wcat, W, MDX_NTW, ND, 0, ND);
      c.pack_dense_split(&b.nds.Wcat, W, MDX_NTW, ND, 0, ND);
      c.wcat_dense[i] = W;
      c.raw(&b.nd.bcat, bias);
    }
    if (cf.update_pos) {
      c.mlp(&b.nd.left, pb + ".left_lin_edge", ND, ED, ED);
      c.mlp(&b.nd.right, pb + ".right_lin_edge", ND, ED, ED);
      c.packDS(&b.nds.leftW1, pb + ".left_lin_edge.net.0.weight", ED, ND, 0, ND);
      c.packDS(&b.nds.leftW2, pb + ".left_lin_edge.net.3.weight", ED, ED, 0, ED);
      c.packDS(&b.nds.rightW1, pb + ".right_lin_edge.net.0.weight", ED, ND, 0, ND);
      c.packDS(&b.nds.rightW2, pb + ".right_lin_edge.net.3.weight", ED, ED, 0, ED);
      const std::string el = pb + ".edge_lin";
      c.packA(&b.eb.Wbl, el + ".bond_linear.weight", ND, ED, 0, ED);
      c.packA(&b.eb.Wnl, el + ".node_linear.weight", ND, ED, 0, ED);
      c.packA(&b.eb.Wi1, el + ".inter_module.net.0.weight", ND, ND, 0, ND);
      c.vec(&b.eb.bi1, el + ".inter_module.net.0.bias", ND);
      c.vec(&b.eb.ig, el + ".inter_module.net.1.weight", ND);
      c.vec(&b.eb.ib, el + ".inter_module.net.1.bias", ND);
      if (const HostTensor* t = c.get(el + ".inter_module.net.3.weight", {1, ND})) c.raw(&b.eb.wi2, t->data);
      if (const HostTensor* t = c.get(el + ".inter_module.net.3.bias", {1})) b.eb.bi2 = t->data[0];
      c.packA(&b.eb.Wg1h, el + ".gate.net.0.weight", 32, 2 * ED + 1, 0, ED);
      c.packA(&b.eb.Wg1a, el + ".gate.net.0.weight", 32, 2 * ED + 1, ED, ED);
      c.packS(&b.eb.s.Wbl, el + ".bond_linear.weight", ND, ED, 0, ED);
      c.packS(&b.eb.s.Wnl, el + ".node_linear.weight", ND, ED, 0, ED);
      c.packS(&b.eb.s.Wi1, el + ".inter_module.net.0.weight", ND, ND, 0, ND);
      c.packS(&b.eb.s.Wg1h, el + ".gate.net.0.weight", 32, 2 * ED + 1, 0, ED);
      c.packS(&b.eb.s.Wg1a, el + ".gate.net.0.weight", 32, 2 * ED + 1, ED, ED);
      c.packSS(&b.eb.ss.Wbl, el + ".bond_linear.weight", ND, ED, 0, ED);
      c.packSS(&b.eb.ss.Wnl, el + ".node_linear.weight", ND, ED, 0, ED);
      c.packSS(&b.eb.ss.Wi1, el + ".inter_module.net.0.weight", ND, ND, 0, ND);
      c.packSS(&b.eb.ss.Wg1h, el + ".gate.net.0.weight", 32, 2 * ED + 1, 0, ED);
      c.packSS(&b.eb.ss.Wg1a, el + ".gate.net.0.weight", 32, 2 * ED + 1, ED, ED);
      c.vec(&b.eb.bg1, el + ".gate.net.0.bias", 32);
      c.column(&b.eb.wtg1, el + ".gate.net.0.weight", 32, 2 * ED + 1, 2 * ED);
      c.vec(&b.eb.gg, el + ".gate.net.1.weight", 32);
      c.vec(&b.eb.gb, el + ".gate.net.1.bias", 32);
      if (const HostTensor* t = c.get(el + ".gate.net.3.weight", {1, 32})) c.raw(&b.eb.wg2, t->data);
      if (const HostTensor* t = c.get(el + ".gate.net.3.bias", {1})) b.eb.bg2 = t->data[0];
    }
  }
  if (cf.kind == MDX_KIND_MOLDIFF || cf.kind == MDX_KIND_BONDPRED) {
    const int nd_emb = ND - cf.time_dim, ed_emb = ED - cf.time_dim;
    const int ein = cf.kind == MDX_KIND_MOLDIFF ? cf.num_edge_types : 2 * cf.num_node_types;
    if (const HostTensor* t = c.get("node_embedder.weight", {nd_emb, cf.num_node_types})) c.raw(&m->Wn, t->data);
    if (const HostTensor* t = c.get("edge_embedder.weight", {ed_emb, ein})) c.raw(&m->We, t->data);
    const std::string te = cf.kind == MDX_KIND_MOLDIFF ? "time_emb.0." : "time_emb.";
    if (cf.time_dim > 0) {  // 0: the time-free bond predictor (models/bond_predictor.py:27-31) has no time embedding
      c.vec(&m->toff, te + "offset", cf.time_dim);
      c.vec(&m->tcoef, te + "coeff", cf.time_dim);
    }
  }
  if (cf.kind == MDX_KIND_MOLDIFF) {
    c.mlp(&m->nodedec, "node_decoder", ND, ND, cf.num_node_types, 16);
    c.mlp(&m->edgedec, "edge_decoder", ED, ED, cf.num_edge_types, 16);
  }
  if (cf.kind == MDX_KIND_BONDPRED) {
    const std::string d = "edge_decoder.net.";
    BondDecW& w = m->dec;
    c.packA(&w.W1e, d + "0.weight", ED, ED + ND, 0, ED);
    c.packA(&w.W1n, d + "0.weight", ED, ED + ND, ED, ND);
    c.vec(&w.b1, d + "0.bias", ED);
    c.vec(&w.g1, d + "1.weight", ED);
    c.vec(&w.be1, d + "1.bias", ED);
    c.packA(&w.W2, d + "3.weight", ED, ED, 0, ED);
    c.vec(&w.b2, d + "3.bias", ED);
    c.vec(&w.g2, d + "4.weight", ED);
    c.vec(&w.be2, d + "4.bias", ED);
    c.packA(&w.W3, d + "6.weight", cf.num_edge_types, ED, 0, ED);
    c.vec(&w.b3, d + "6.bias", cf.num_edge_types, 16);
    c.packT(&w.W1eT, d + "0.weight", ED, ED + ND, 0, ED);
    c.packT(&w.W1nT, d + "0.weight", ED, ED + ND, ED, ND);
    c.packT(&w.W2T, d + "3.weight", ED, ED, 0, ED);
    c.packT(&w.W3T, d + "6.weight", cf.num_edge_types, ED, 0, ED);
    // transposed packs for the data-gradient backward (guidance)
    m->ebw.assign(cf.num_blocks, EdgeBwdW{});
    m->nbw.assign(cf.num_blocks, NodeBwdW{});
    m->nbws.assign(cf.num_blocks, NodeBwdWS{});
    for (int i = 0; i < cf.num_blocks; ++i) {
      const std::string si = std::to_string(i);
      const std::string nb = net + "node_blocks_with_edge." + si, eb = net + "edge_blocks." + si;
      EdgeBwdW& e = m->ebw[i];
      c.packT(&e.WembHT, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, 0, ED);
      c.packT(&e.WembDT, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, ED, MDX_NG);
      c.packT(&e.Wg1eT, nb + ".gate.net.0.weight", ND, GIN, 0, ED);
      c.packT(&e.Wg2T, nb + ".gate.net.3.weight", ND, ND, 0, ND);
      c.packT(&e.W1T, nb + ".edge_net.net.0.weight", ND, ED, 0, ED);
      c.packT(&e.W2T, nb + ".edge_net.net.3.weight", ND, ND, 0, ND);
      c.packT(&e.WmT, nb + ".msg_net.weight", ND, ND, 0, ND);
      for (int s = 0; s < 2; ++s) {
        const std::string fp = eb + (s ? ".bond_ffn_right" : ".bond_ffn_left");
        FfnWT& f = e.ffn[s];
        c.packT(&f.WblT, fp + ".bond_linear.weight", 2 * ED, ED, 0, ED);
        c.packT(&f.Wi1T, fp + ".inter_module.net.0.weight", 2 * ED, 2 * ED, 0, 2 * ED);
        c.packT(&f.Wi2T, fp + ".inter_module.net.3.weight", ED, 2 * ED, 0, 2 * ED);
        c.packT(&f.Wg1eT, fp + ".gate.net.0.weight", 32, GIN, 0, ED);
        c.packT(&f.Wg2T, fp + ".gate.net.3.weight", ED, 32, 0, 32);
      }
      c.packT(&e.WselfT, eb + ".self_ffn.weight", ED, ED, 0, ED);
      c.packT(&e.WoutT, eb + ".out_transform.weight", ED, ED, 0, ED);
      // the same transposes as stream packs for the row-owner backward kernel
      c.packTS(&e.s.WembHT, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, 0, ED);
      c.packTS(&e.s.WembDT, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, ED, MDX_NG);
      c.packTS(&e.s.Wg1eT, nb + ".gate.net.0.weight", ND, GIN, 0, ED);
      c.packTS(&e.s.Wg2T, nb + ".gate.net.3.weight", ND, ND, 0, ND);
      c.packTS(&e.s.W1T, nb + ".edge_net.net.0.weight", ND, ED, 0, ED);
      c.packTS(&e.s.W2T, nb + ".edge_net.net.3.weight", ND, ND, 0, ND);
      c.packTS(&e.s.WmT, nb + ".msg_net.weight", ND, ND, 0, ND);
      c.packTS(&e.s.WselfT, eb + ".self_ffn.weight", ED, ED, 0, ED);
      c.packTS(&e.s.WoutT, eb + ".out_transform.weight", ED, ED, 0, ED);
      for (int s = 0; s < 2; ++s) {
        const std::string fp = eb + (s ? ".bond_ffn_right" : ".bond_ffn_left");
        FfnTS& f = e.s.ffn[s];
        c.packTS(&f.WblT, fp + ".bond_linear.weight", 2 * ED, ED, 0, ED);
        c.packTS(&f.Wi1T, fp + ".inter_module.net.0.weight", 2 * ED, 2 * ED, 0, 2 * ED);
        c.packTS(&f.Wi2T, fp + ".inter_module.net.3.weight", ED, 2 * ED, 0, 2 * ED);
        c.packTS(&f.Wg1eT, fp + ".gate.net.0.weight", 32, GIN, 0, ED);
        c.packTS(&f.Wg2T, fp + ".gate.net.3.weight", ED, 32, 0, 32);
      }
      // ... and as split float16 stream packs for the split build of that kernel (mdx_bwd2s.hip)
      c.packTSS(&e.ss.WembHT, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, 0, ED);
      c.packTSS(&e.ss.WembDT, net + "edge_embs." + si + ".weight", ED, ED + MDX_NG, ED, MDX_NG);
      c.packTSS(&e.ss.Wg1eT, nb + ".gate.net.0.weight", ND, GIN, 0, ED);
      c.packTSS(&e.ss.Wg2T, nb + ".gate.net.3.weight", ND, ND, 0, ND);
      c.packTSS(&e.ss.W1T, nb + ".edge_net.net.0.weight", ND, ED, 0, ED);
      c.packTSS(&e.ss.W2T, nb + ".edge_net.net.3.weight", ND, ND, 0, ND);
      c.packTSS(&e.ss.WmT, nb + ".msg_net.weight", ND, ND, 0, ND);
      c.packTSS(&e.ss.WselfT, eb + ".self_ffn.weight", ED, ED, 0, ED);
      c.packTSS(&e.ss.WoutT, eb + ".out_transform.weight", ED, ED, 0, ED);
      for (int s = 0; s < 2; ++s) {
        const std::string fp = eb + (s ? ".bond_ffn_right" : ".bond_ffn_left");
        FfnTS& f = e.ss.ffn[s];
        c.packTSS(&f.WblT, fp + ".bond_linear.weight", 2 * ED, ED, 0, ED);
        c.packTSS(&f.Wi1T, fp + ".inter_module.net.0.weight", 2 * ED, 2 * ED, 0, 2 * ED);
        c.packTSS(&f.Wi2T, fp + ".inter_module.net.3.weight", ED, 2 * ED, 0, 2 * ED);
        c.packTSS(&f.Wg1eT, fp + ".gate.net.0.weight", 32, GIN, 0, ED);
        c.packTSS(&f.Wg2T, fp + ".gate.net.3.weight", ED, 32, 0, 32);
      }
      NodeBwdW& n = m->nbw[i];
      c.packT(&n.WoutT, nb + ".out_transform.weight", ND, ND, 0, ND);
      c.packT(&n.W1T, nb + ".node_net.net.0.weight", ND, ND, 0, ND);
      c.packT(&n.W2T, nb + ".node_net.net.3.weight", ND, ND, 0, ND);
      const std::vector<float>& Wcat = c.wcat_dense[i];
      if (!Wcat.empty())
        for (int ch = 0; ch < 4; ++ch) {
          const int kc = ch < 3 ? 256 : 192;
          std::vector<float> Mt((size_t)ND * kc);
          for (int f = 0; f < ND; ++f)
            for (int k = 0; k < kc; ++k) Mt[(size_t)f * kc + k] = Wcat[(size_t)(256 * ch + k) * ND + f];
          c.pack_dense(&n.WcatT[ch], Mt, ND, kc, 0, kc);
          c.pack_dense_split(&m->nbws[i].WcatT[ch], Mt, ND, kc, 0, kc);
        }
      c.packTDS(&m->nbws[i].WoutT, nb + ".out_transform.weight", ND, ND, 0, ND);
      c.packTDS(&m->nbws[i].W1T, nb + ".node_net.net.0.weight", ND, ND, 0, ND);
      c.packTDS(&m->nbws[i].W2T, nb + ".node_net.net.3.weight", ND, ND, 0, ND);
    }
  }
  if (!c.missing.empty()) return fail(MDX_ERR_STATE, "missing or mis-shaped parameter: %s", c.missing.c_str());
  if (m->arena) hipFree(m->arena);
  m->arena = nullptr;
  m->arena_floats = c.pk.host.size();
  HIPCHK(hipMalloc((void**)&m->arena, m->arena_floats * sizeof(float)));
  HIPCHK(hipMemcpy(m->arena, c.pk.host.data(), m->arena_floats * sizeof(float), hipMemcpyHostToDevice));
  for (auto& f : c.pk.fix) *f.first = m->arena + f.second;
  m->finalized = true;
  // a handle re-finalized with weights the split path cannot hold falls back to the exact path instead of silently running
  // float16 operands that overflowed at pack time (the caller's next mdx_model_set_matrix_path(split) is refused as usual)
  if (!m->split_ok) m->matrix_path = MDX_MATRIX_EXACT_F32;
  return MDX_OK;
}

}  // namespace

extern "C" int mdx_model_create(const mdx_config* cfg, mdx_model_t* out) {
  if (!cfg || !out) return fail(MDX_ERR_ARG, "null argument");
  if (cfg->node_dim != MDX_ND || cfg->edge_dim != MDX_ED || cfg->num_gaussians != MDX_NG)
    return fail(MDX_ERR_UNSUPPORTED, "kernels are built for node_dim=%d edge_dim=%d num_gaussians=%d (got %d/%d/%d)", MDX_ND,
                MDX_ED, MDX_NG, cfg->node_dim, cfg->edge_dim, cfg->num_gaussians);
  if (cfg->num_blocks < 1 || cfg->num_blocks > 64) return fail(MDX_ERR_ARG, "num_blocks out of range");
  if (cfg->kind != MDX_KIND_NET) {
    if (cfg->num_node_types < 1 || cfg->num_node_types > 8 || cfg->num_edge_types < 1 || cfg->num_edge_types > 8)
      return fail(MDX_ERR_UNSUPPORTED, "class counts must be in 1..8");
    if (cfg->time_dim < 0 || cfg->time_dim >= MDX_ED) return fail(MDX_ERR_ARG, "time_dim out of range");
  }
  mdx_model_s* m = new mdx_model_s();
  m->cfg = *cfg;
  *out = m;
  return MDX_OK;
}

extern "C" int mdx_model_destroy(mdx_model_t m) {
  if (!m) return MDX_OK;
  if (m->arena) hipFree(m->arena);
  delete m;
  return MDX_OK;
}

extern "C" int mdx_model_set_param(mdx_model_t m, const char* key, const float* h_data, const int64_t* shape,
                                   int32_t ndim) {
  if (!m || !key || !h_data || (ndim > 0 && !shape) || ndim < 0 || ndim > 4) return fail(MDX_ERR_ARG, "bad argument");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 0) return fail(MDX_ERR_ARG, "negative dim");
    t.shape.push_back(shape[i]);
    n *= (size_t)shape[i];
  }
  t.data.assign(h_data, h_data + n);
  m->params[key] = std::move(t);
  m->finalized = false;
  return MDX_OK;
}

extern "C" int mdx_model_finalize(mdx_model_t m) {
  if (!m) return fail(MDX_ERR_ARG, "null model");
  return pack_model(m);
}

// Matrix path of the row-owner edge kernels (the products of models/common.py:181-201 MLP / models/graph.py:29-55,133-141,268-295
// Linear layers evaluated per edge): MDX_MATRIX_EXACT_F32 = v_mfma_f32_16x16x4_f32, bit-for-bit an fmaf chain (default);
// MDX_MATRIX_SPLIT_F16 = operands split into float16 hi + lo halves, three v_mfma_f32_16x16x32_f16 per k-group, fp32 accumulation
// (mdx_split.h).  Per-node layers, reductions, LayerNorm, gates and transitions are fp32 in both.
extern "C" int mdx_model_set_matrix_path(mdx_model_t m, int32_t path) {
  if (!m) return fail(MDX_ERR_ARG, "null model");
  if (path != MDX_MATRIX_EXACT_F32 && path != MDX_MATRIX_SPLIT_F16) return fail(MDX_ERR_ARG, "unknown matrix path %d", (int)path);
  if (path == MDX_MATRIX_SPLIT_F16) {
    if (!m->finalized) return fail(MDX_ERR_STATE, "model not finalized (call mdx_model_finalize)");
    if (!m->split_ok)
      return fail(MDX_ERR_UNSUPPORTED, "split float16 path refused: a weight of magnitude %g is outside float16's range (|w| < 65504)",
                  (double)m->split_wmax);
  }
  m->matrix_path = path;
  return MDX_OK;
}
extern "C" int mdx_model_set_smear_start(mdx_model_t m, float start) {
  if (!m) return fail(MDX_ERR_ARG, "null model");
  if (!(start >= 0.f) || !(start < m->cfg.cutoff)) return fail(MDX_ERR_ARG, "smearing start %g outside [0, cutoff)", (double)start);
  m->smear_start = start;
  return MDX_OK;
}
extern "C" int mdx_model_get_matrix_path(mdx_model_t m, int32_t* path) {
  if (!m || !path) return fail(MDX_ERR_ARG, "null argument");
  *path = m->matrix_path;
  return MDX_OK;
}

// ------------------------------------------------------------------------------------------------
// graph
// ------------------------------------------------------------------------------------------------
constexpr int MDX_WQ_SETS = 4, MDX_WQ_SET_INTS = 512 * 32;  // mdx_row.h: MDX_WQ_PAIRS lines of MDX_WQ_STRIDE ints per set
struct mdx_graph_s {
  int64_t N = 0, E = 0, Eh = 0, B = 0;
  int32_t* dev = nullptr;   // one int32 slab
  int64_t* mol_ids = nullptr;
  const int32_t *left, *right, *int2ref, *ref2int, *row_ptr, *col_ptr, *col_eids, *node_graph, *node_local, *he_graph,
      *he_local, *half_of_int, *node_ptr, *he_ptr;
  // graph-aligned 16-row units of edge kernel A's in-kernel aggregation (EA_AGG, mdx_kernels.h): units (2 ints per unit: first
  // edge, rows), per-edge partial-row offset, per-node partial-row ranges
  const int32_t *units = nullptr, *epo = nullptr, *pbase = nullptr;
  int64_t nunits = 0, nparts = 0;
  // the same in BY-RIGHT order (positions of col_eids), for the guidance backward's in-kernel sums of its by-right payloads
  // (round 5): units_r (first col position, rows), epo_r per col position, pbase_r per node; col_left / col_right = the end
  // points of edge col_eids[j] (so a tile's indices are independent loads)
  const int32_t *units_r = nullptr, *epo_r = nullptr, *pbase_r = nullptr, *col_left = nullptr, *col_right = nullptr;
  int64_t nunits_r = 0, nparts_r = 0;
  hipEvent_t ev_in = nullptr, ev_done = nullptr;  // stream hand-offs of mdx_sample_step_full's concurrent guidance chain
  // work-queue counter sets of the persistent edge kernels, one per launching stream (wq_for)
  int* wq = nullptr;
  hipStream_t wq_stream[MDX_WQ_SETS] = {};
  int wq_n = 0;
  std::mutex wq_mu;
};

namespace {
struct HostPlan {
  std::vector<int32_t> left, right, int2ref, ref2int, row_ptr, col_ptr, col_eids;
};

int plan_graph(int64_t N, int64_t E, const int64_t* ei, HostPlan& p) {
  if (N < 0 || E < 0 || N > INT32_MAX / 4 || E > INT32_MAX / 4) return fail(MDX_ERR_ARG, "graph too large");
  const int64_t* L = ei;
  const int64_t* R = ei + E;
  for (int64_t e = 0; e < E; ++e)
    if (L[e] < 0 || L[e] >= N || R[e] < 0 || R[e] >= N) return fail(MDX_ERR_ARG, "edge_index out of range at %lld", (long long)e);
  p.int2ref.resize(E);
  std::iota(p.int2ref.begin(), p.int2ref.end(), 0);
  std::stable_sort(p.int2ref.begin(), p.int2ref.end(), [&](int32_t a, int32_t b) {
    if (L[a] != L[b]) return L[a] < L[b];
    return R[a] < R[b];
  });
  p.left.resize(E);
  p.right.resize(E);
  p.ref2int.resize(E);
  p.row_ptr.assign(N + 1, 0);
  p.col_ptr.assign(N + 1, 0);
  for (int64_t i = 0; i < E; ++i) {
    const int32_t e = p.int2ref[i];
    p.left[i] = (int32_t)L[e];
    p.right[i] = (int32_t)R[e];
    p.ref2int[e] = (int32_t)i;
    p.row_ptr[L[e] + 1]++;
    p.col_ptr[R[e] + 1]++;
  }
  for (int64_t v = 0; v < N; ++v) {
    p.row_ptr[v + 1] += p.row_ptr[v];
    p.col_ptr[v + 1] += p.col_ptr[v];
  }
  p.col_eids.resize(E);
  std::vector<int32_t> cur(p.col_ptr.begin(), p.col_ptr.end() - 1);
  for (int64_t i = 0; i < E; ++i) p.col_eids[cur[p.right[i]]++] = (int32_t)i;
  return MDX_OK;
}
}  // namespace

extern "C" int mdx_graph_plan_host(int64_t N, int64_t E, const int64_t* ei, int32_t* left, int32_t* right,
                                   int32_t* int2ref, int32_t* row_ptr, int32_t* col_ptr, int32_t* col_eids) {
  if ((E > 0 && !ei) || !left || !right || !int2ref || !row_ptr || !col_ptr || !col_eids) return fail(MDX_ERR_ARG, "null argument");
  HostPlan p;
  int rc = plan_graph(N, E, ei, p);
  if (rc) return rc;
  std::copy(p.left.begin(), p.left.end(), left);
  std::copy(p.right.begin(), p.right.end(), right);
  std::copy(p.int2ref.begin(), p.int2ref.end(), int2ref);
  std::copy(p.row_ptr.begin(), p.row_ptr.end(), row_ptr);
  std::copy(p.col_ptr.begin(), p.col_ptr.end(), col_ptr);
  std::copy(p.col_eids.begin(), p.col_eids.end(), col_eids);
  return MDX_OK;
}

extern "C" int mdx_graph_create(int64_t N, int64_t E, const int64_t* ei, const int64_t* bn, int64_t B,
                                const int64_t* mol_ids, mdx_graph_t* out) {
  if (!out || (E > 0 && !ei) || (N > 0 && !bn) || B < 0) return fail(MDX_ERR_ARG, "null argument");
  HostPlan p;
  int rc = plan_graph(N, E, ei, p);
  if (rc) return rc;
  std::vector<int32_t> node_graph(N), node_local(N);
  {
    int64_t prev = -1, cnt = 0;
    for (int64_t v = 0; v < N; ++v) {
      if (bn[v] < 0 || bn[v] >= B) return fail(MDX_ERR_ARG, "batch_node[%lld] out of range", (long long)v);
      if (bn[v] < prev) return fail(MDX_ERR_ARG, "batch_node must be non-decreasing");
      if (bn[v] != prev) cnt = 0;
      node_graph[v] = (int32_t)bn[v];
      node_local[v] = (int32_t)cnt++;
      prev = bn[v];
    }
  }
  // half-edge bookkeeping (reference order: first E/2 columns are the i<j pairs, model.py:269)
  const int64_t Eh = E / 2;
  std::vector<int32_t> he_graph(Eh), he_local(Eh);
  {
    int64_t prev = -1, cnt = 0;
    for (int64_t h = 0; h < Eh; ++h) {
      const int64_t gph = bn[ei[h]];
      if (gph != prev) cnt = 0;
      he_graph[h] = (int32_t)gph;
      he_local[h] = (int32_t)cnt++;
      prev = gph;
    }
  }
  mdx_graph_s* g = new mdx_graph_s();
  g->N = N; g->E = E; g->Eh = Eh; g->B = B;
  std::vector<int32_t> slab;
  auto add = [&](const std::vector<int32_t>& v) {
    size_t off = (slab.size() + 63) & ~size_t(63);
    slab.resize(off + v.size() + 1, 0);
    std::copy(v.begin(), v.end(), slab.begin() + off);
    return off;
  };
  const size_t o_l = add(p.left), o_r = add(p.right), o_i2r = add(p.int2ref), o_r2i = add(p.ref2int), o_rp = add(p.row_ptr),
               o_cp = add(p.col_ptr), o_ce = add(p.col_eids), o_ng = add(node_graph), o_nl = add(node_local),
               o_hg = add(he_graph), o_hl = add(he_local);
  std::vector<int32_t> node_ptr(B + 1, 0), he_ptr(B + 1, 0);
  for (int64_t v = 0; v < N; ++v) node_ptr[bn[v] + 1]++;
  for (int64_t h = 0; h < Eh; ++h) he_ptr[he_graph[h] + 1]++;
  for (int64_t b = 0; b < B; ++b) { node_ptr[b + 1] += node_ptr[b]; he_ptr[b + 1] += he_ptr[b]; }
  const size_t o_np = add(node_ptr), o_hp = add(he_ptr);
  // Units of 16 edges aligned to each graph's first edge: graph b owns the edges row_ptr[node_ptr[b]] .. row_ptr[node_ptr[b+1]]
  // (batch_node is non-decreasing and edges are sorted by left node).  A node's run is cut where a unit ends, so the cuts --
  // and the association of the in-kernel sums -- are a function of the graph alone, not of its offset in the batch.
  std::vector<int32_t> units, epo(E), pbase(N + 1, 0);
  for (int64_t b = 0; b < B; ++b) {
    const int32_t e_lo = p.row_ptr[node_ptr[b]], e_hi = p.row_ptr[node_ptr[b + 1]];
    for (int32_t e0 = e_lo; e0 < e_hi; e0 += 16) {
      units.push_back(e0);
      units.push_back(std::min<int32_t>(16, e_hi - e0));
    }
  }
  {
    const int64_t U = (int64_t)units.size() / 2;
    int64_t u = 0;  // unit of the current edge (units are in edge order)
    for (int64_t v = 0; v < N; ++v) {
      const int32_t r0 = p.row_ptr[v], r1 = p.row_ptr[v + 1];
      int32_t pieces = 0, ufirst = 0;
      if (r1 > r0) {
        while (u + 1 < U && units[2 * (u + 1)] <= r0) ++u;
        ufirst = (int32_t)u;
        int64_t ul = u;
        while (ul + 1 < U && units[2 * (ul + 1)] <= r1 - 1) ++ul;
        pieces = (int32_t)(ul - u + 1);
      }
      pbase[v + 1] = pbase[v] + pieces;
      for (int32_t e = r0; e < r1; ++e) epo[e] = pbase[v] - ufirst;
    }
  }
  g->nunits = (int64_t)units.size() / 2;
  g->nparts = pbase[N];
  const size_t o_un = add(units), o_epo = add(epo), o_pb = add(pbase);
  // by-right twin: graph b owns the col positions col_ptr[node_ptr[b]] .. col_ptr[node_ptr[b+1]] (col_eids is ordered by right node)
  std::vector<int32_t> units_r, epo_r(E), pbase_r(N + 1, 0), col_left(E), col_right(E);
  for (int64_t j = 0; j < E; ++j) { col_left[j] = p.left[p.col_eids[j]]; col_right[j] = p.right[p.col_eids[j]]; }
  for (int64_t b = 0; b < B; ++b) {
    const int32_t j_lo = p.col_ptr[node_ptr[b]], j_hi = p.col_ptr[node_ptr[b + 1]];
    for (int32_t j0 = j_lo; j0 < j_hi; j0 += 16) {
      units_r.push_back(j0);
      units_r.push_back(std::min<int32_t>(16, j_hi - j0));
    }
  }
  {
    const int64_t U = (int64_t)units_r.size() / 2;
    int64_t u = 0;
    for (int64_t v = 0; v < N; ++v) {
      const int32_t r0 = p.col_ptr[v], r1 = p.col_ptr[v + 1];
      int32_t pieces = 0, ufirst = 0;
      if (r1 > r0) {
        while (u + 1 < U && units_r[2 * (u + 1)] <= r0) ++u;
        ufirst = (int32_t)u;
        int64_t ul = u;
        while (ul + 1 < U && units_r[2 * (ul + 1)] <= r1 - 1) ++ul;
        pieces = (int32_t)(ul - u + 1);
      }
      pbase_r[v + 1] = pbase_r[v] + pieces;
      for (int32_t j = r0; j < r1; ++j) epo_r[j] = pbase_r[v] - ufirst;
    }
  }
  g->nunits_r = (int64_t)units_r.size() / 2;
  g->nparts_r = pbase_r[N];
  const size_t o_unr = add(units_r), o_epor = add(epo_r), o_pbr = add(pbase_r), o_cl = add(col_left), o_cr = add(col_right);
  std::vector<int32_t> half_of_int(E);
  for (int64_t i = 0; i < E; ++i) half_of_int[i] = Eh > 0 ? (int32_t)(p.int2ref[i] % Eh) : 0;
  const size_t o_hoi = add(half_of_int);
  const size_t o_wq = add(std::vector<int32_t>((size_t)MDX_WQ_SET_INTS * MDX_WQ_SETS, 0));  // work-queue counters: all zero between launches
  if (hipMalloc((void**)&g->dev, slab.size() * 4) != hipSuccess ||
      hipMemcpy(g->dev, slab.data(), slab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
    delete g;
    return fail(MDX_ERR_HIP, "graph upload failed: %s", hipGetErrorString(hipGetLastError()));
  }
  std::vector<int64_t> ids(B);
  for (int64_t i = 0; i < B; ++i) ids[i] = mol_ids ? mol_ids[i] : i;
  if (hipMalloc((void**)&g->mol_ids, std::max<int64_t>(B, 1) * 8) != hipSuccess ||
      (B > 0 && hipMemcpy(g->mol_ids, ids.data(), B * 8, hipMemcpyHostToDevice) != hipSuccess)) {
    hipFree(g->dev);
    delete g;
    return fail(MDX_ERR_HIP, "graph upload failed");
  }
  g->left = g->dev + o_l; g->right = g->dev + o_r; g->int2ref = g->dev + o_i2r; g->ref2int = g->dev + o_r2i;
  g->row_ptr = g->dev + o_rp; g->col_ptr = g->dev + o_cp; g->col_eids = g->dev + o_ce; g->node_graph = g->dev + o_ng;
  g->node_local = g->dev + o_nl; g->he_graph = g->dev + o_hg; g->he_local = g->dev + o_hl;
  g->half_of_int = g->dev + o_hoi;
  g->node_ptr = g->dev + o_np;
  g->he_ptr = g->dev + o_hp;
  g->units = g->dev + o_un; g->epo = g->dev + o_epo; g->pbase = g->dev + o_pb;
  g->units_r = g->dev + o_unr; g->epo_r = g->dev + o_epor; g->pbase_r = g->dev + o_pbr; g->col_left = g->dev + o_cl;
  g->col_right = g->dev + o_cr;
  g->wq = g->dev + o_wq;
  *out = g;
  return MDX_OK;
}

extern "C" int mdx_graph_destroy(mdx_graph_t g) {
  if (!g) return MDX_OK;
  if (g->dev) hipFree(g->dev);
  if (g->mol_ids) hipFree(g->mol_ids);
  if (g->ev_in) hipEventDestroy(g->ev_in);
  if (g->ev_done) hipEventDestroy(g->ev_done);
  delete g;
  return MDX_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
namespace {
struct Ws {
  float *Hn, *H, *NT, *NT2, *aggr, *SL, *SR, *Lf, *Rf, *tn, *posA, *posB;
  float *HeA, *HeB, *M, *FL, *FR, *Fe, *te, *tnr, *tmpE;  // tmpE: (E,64) scratch for boundary permutes
  float *P, *PR;  // partial rows of the in-kernel aggregation: (2N + E/16 + 1) x 256 / x 64 (bound on a graph's partial rows)
  bool tnr_set;  // tnr holds node_time[right] (only the bare NodeEdgeNet API can make it differ from te)
  size_t bytes;
};

size_t ws_layout(int64_t N, int64_t E, char* base, Ws* w) {
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((nfloat * 4 + 255) / 256) * 256;
    return p;
  };
  const size_t n = (size_t)std::max<int64_t>(N, 1), e = (size_t)std::max<int64_t>(E, 1);
  Ws t{};
  t.Hn = take(n * MDX_ND); t.H = take(n * MDX_ND); t.NT = take(n * MDX_NTW); t.NT2 = take(n * MDX_NTW); t.aggr = take(n * MDX_ND);
  t.SL = take(n * 64); t.SR = take(n * 64); t.Lf = take(n * 64); t.Rf = take(n * 64); t.tn = take(n);
  t.posA = take(n * 3); t.posB = take(n * 3);
  t.HeA = take(e * 64); t.HeB = take(e * 64); t.M = take(e * MDX_ND); t.FL = take(e * 64); t.FR = take(e * 64);
  t.Fe = take(e * 3); t.te = take(e); t.tnr = take(e); t.tmpE = take(e * 64);
  // every node with edges has one piece per unit its run touches: <= N + (#units) <= N + (E/16 + #graphs) <= 2N + E/16 rows
  const size_t np = 2 * n + e / 16 + 1;
  t.P = take(np * MDX_ND); t.PR = take(np * 64);
  t.bytes = off;
  if (w) *w = t;
  return off;
}
}  // namespace

extern "C" size_t mdx_workspace_bytes(int64_t N, int64_t E) { return ws_layout(N, E, nullptr, nullptr); }

// ------------------------------------------------------------------------------------------------
// small utility kernels (permutes at the ABI boundary)
// ------------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ dst,
                                   int n, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * C) return;
  const int r = i / C, c = i - (size_t)r * C;
  dst[i] = src[(size_t)idx[r] * C + c];
}
static void gather_rows(const float* src, const int* idx, float* dst, int64_t n, int C, hipStream_t s) {
  if (n <= 0) return;
  const size_t tot = (size_t)n * C;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, src, idx, dst, (int)n, C);
}



// ------------------------------------------------------------------------------------------------
// live kernel timing (hipEvents on the launch stream) -- used by bench.py for the roofline numbers
// ------------------------------------------------------------------------------------------------
namespace {
enum { PK_EDGE_A = 0, PK_EDGE_B = 1, PK_NODE = 2, PK_AGGR = 3, PK_EDGE_BWD = 4, PK_COUNT = 5 };
constexpr int PROF_RING = 1024;
struct ProfSlot {
  hipEvent_t a[PROF_RING], b[PROF_RING];
  bool made = false;
  int head = 0, pending = 0;
  double total_ms = 0.0;
  long long count = 0;
};
unsigned g_prof_mask = 0;  // bit k: kernel k is timed
ProfSlot g_prof[PK_COUNT];

void prof_drain(ProfSlot& p, int n) {
  for (; n > 0 && p.pending > 0; --n, --p.pending) {
    const int i = (p.head - p.pending + PROF_RING * 4) % PROF_RING;
    hipEventSynchronize(p.b[i]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a[i], p.b[i]) == hipSuccess) {
      p.total_ms += ms;
      p.count++;
    }
  }
}
struct ProfScope {
  ProfSlot* p = nullptr;
  hipStream_t s;
  int i = 0;
  ProfScope(int k, hipStream_t st) : s(st) {
    if (!((g_prof_mask >> k) & 1u)) return;
    p = &g_prof[k];
    if (!p->made) {
      for (int j = 0; j < PROF_RING; ++j) {
        hipEventCreate(&p->a[j]);
        hipEventCreate(&p->b[j]);
      }
      p->made = true;
    }
    if (p->pending == PROF_RING) prof_drain(*p, PROF_RING / 2);
    i = p->head;
    hipEventRecord(p->a[i], s);
  }
  ~ProfScope() {
    if (!p) return;
    hipEventRecord(p->b[i], s);
    p->head = (p->head + 1) % PROF_RING;
    p->pending++;
  }
};
}  // namespace

extern "C" int mdx_profile_enable(int32_t on) {
  g_prof_mask = on == 1 ? ~0u : (unsigned)on >> 1;  // 0 off, 1 all kernels, else bit (k+1) selects kernel k
  if (on)
    for (auto& p : g_prof) { prof_drain(p, PROF_RING); p.total_ms = 0.0; p.count = 0; }
  return MDX_OK;
}
// name of the kernel function slot `kernel` brackets in THIS build (bench.py checks a committed PMC summary against it before
// quoting its traffic figure)
extern "C" const char* mdx_profile_kernel_name(int32_t kernel) {
  switch (kernel) {
    case PK_EDGE_A: return "edge_a2_kernel<15>";
    case PK_EDGE_B: return "edge_b2_kernel";
    case PK_NODE: return "node_kernel";
    case PK_AGGR: return "";   // the reduction left after edge kernel A's in-kernel sums runs inside node_kernel
    case PK_EDGE_BWD: return "edge_bwd2_kernel";
    default: return "";
  }
}
extern "C" int mdx_profile_read(int32_t kernel, int64_t* count, double* total_ms) {
  if (kernel < 0 || kernel >= PK_COUNT || !count || !total_ms) return fail(MDX_ERR_ARG, "bad argument");
  prof_drain(g_prof[kernel], PROF_RING);
  *count = g_prof[kernel].count;
  *total_ms = g_prof[kernel].total_ms;
  return MDX_OK;
}

// ------------------------------------------------------------------------------------------------
// block driver
// ------------------------------------------------------------------------------------------------
namespace {

#define CHECK_READY(m, g, ws, ws_bytes)                                                                    \
  if (!(m) || !(g)) return fail(MDX_ERR_ARG, "null handle");                                               \
  if (!(m)->finalized) return fail(MDX_ERR_STATE, "model not finalized (call mdx_model_finalize)");        \
  if (!(ws) || (ws_bytes) < mdx_workspace_bytes((g)->N, (g)->E))                                           \
    return fail(MDX_ERR_STATE, "workspace too small: need %zu bytes", mdx_workspace_bytes((g)->N, (g)->E)); \
  if (reinterpret_cast<uintptr_t>(ws) & 255) return fail(MDX_ERR_ARG, "workspace must be 256-byte aligned");

EdgeAArgs make_ea(const mdx_model_s* m, const mdx_graph_s* g, const Ws& w, int i, const float* pos, const float* He_in,
                  float* He_out, int flags, const float* NT = nullptr) {
  EdgeAArgs a{};
  a.E = (int)g->E; a.flags = flags; a.l = g->left; a.r = g->right; a.te = w.te; a.tn_r = w.tnr_set ? w.tnr : nullptr;
  a.pos = pos; a.dist_in = nullptr;
  a.soff = m->soff; a.scoef = m->scoef; a.cutoff = m->cfg.cutoff; a.smear_start = m->smear_start; a.He_in = He_in; a.He_out = He_out; a.H = w.H; a.NT = NT ? NT : w.NT;
  a.M = w.M; a.F[0] = w.FL; a.F[1] = w.FR; a.w = m->blocks[i].ea;
  if (flags & EA_AGG) {  // segment sums inside the kernel: M and the BondFFN-right rows stay out of HBM
    a.M = nullptr; a.F[1] = nullptr;
    a.units = g->units; a.nunits = (int)g->nunits; a.epo = g->epo; a.P = w.P; a.PR = w.PR;
  }
  return a;
}

#define LCHK(x)                    \
  do {                             \
    const int rc_ = (x);           \
    if (rc_ != MDX_OK) return rc_; \
  } while (0)

EdgeBArgs make_eb(const mdx_model_s* m, const mdx_graph_s* g, const Ws& w, int i, const float* pos, const float* Hep,
                  float* He_out, int flags, const float* NT = nullptr) {
  EdgeBArgs a{};
  a.E = (int)g->E; a.flags = flags; a.l = g->left; a.r = g->right; a.te = w.te; a.pos = pos; a.rel_in = nullptr;
  a.dist_in = nullptr; a.Hep = Hep; a.SL = w.SL; a.SR = w.SR; a.NT = NT ? NT : w.NT; a.He_out = He_out; a.Lf = w.Lf; a.Rf = w.Rf;
  a.Fe = w.Fe; a.w = m->blocks[i].eb;
  return a;
}

// Work-queue counters of the persistent edge kernels (mdx_row.h, WorkQ): a set per stream that launches on this graph, because
// launches that share a set must be ordered (the guidance chain runs on a second stream beside the denoiser).  A fifth stream
// gets none: its launches use the static split.  MDX_STATIC_SPLIT=1 turns the queues off (A/B).
int* wq_for(const mdx_graph_s* gc, hipStream_t s) {
  static const bool off = [] {
    const char* e = getenv("MDX_STATIC_SPLIT");
    return e && e[0] == '1';
  }();
  mdx_graph_s* g = const_cast<mdx_graph_s*>(gc);
  if (off || !g->wq) return nullptr;
  std::lock_guard<std::mutex> lk(g->wq_mu);
  for (int i = 0; i < g->wq_n; ++i)
    if (g->wq_stream[i] == s) return g->wq + (size_t)MDX_WQ_SET_INTS * i;
  if (g->wq_n == MDX_WQ_SETS) return nullptr;
  g->wq_stream[g->wq_n] = s;
  return g->wq + (size_t)MDX_WQ_SET_INTS * g->wq_n++;
}
// The split float16 forward kernels take the STATIC split: their waves then run identical programs from a common start and stay close
// enough for the L1 to serve the weight fragments one wave pulled to the other three (tools/ubench_split.hip UB_DESYNC: 23 vs 29
// cycles per MFMA), which is worth more to them than the queues' balance -- kernel A 373.5 vs 385.0 us per launch, kernel B 134.5 vs
// 136.0 (profiles/r4_split_phase_trace.txt).  The exact kernels are bound by the matrix pipe and keep the queues (6.95 vs 6.96 ms per
// step either way).  Results do not depend on the distribution (tests/test_gpu_round3.py static-vs-queue parity).
int run_ea(const mdx_graph_s* g, EdgeAArgs a, hipStream_t s) {
  a.wq = (a.flags & EA_SPLIT) ? nullptr : wq_for(g, s);
  return launch_edge_a(a, s);
}
int run_eb(const mdx_graph_s* g, EdgeBArgs a, hipStream_t s) {
  a.wq = (a.flags & EB_SPLIT) ? nullptr : wq_for(g, s);
  return launch_edge_b(a, s);
}

NodeArgs make_nd(const mdx_model_s* m, const mdx_graph_s* g, const Ws& w, int imid, int ipre, int flags,
                 const float* NTin = nullptr, float* NTout = nullptr) {
  NodeArgs a{};
  a.N = (int)g->N; a.flags = flags; a.Hn = w.Hn; a.aggr = w.aggr; a.NTin = NTin ? NTin : w.NT; a.dHn = nullptr; a.Lf = w.Lf;
  a.Rf = w.Rf; a.H = w.H; a.NT = NTout ? NTout : w.NT;
  if (imid >= 0) { a.wmid = m->blocks[imid].nd; a.smid = m->blocks[imid].nds; }
  if (ipre >= 0) { a.wpre = m->blocks[ipre].nd; a.spre = m->blocks[ipre].nds; }
  if (m->matrix_path == MDX_MATRIX_SPLIT_F16) a.flags |= ND_SPLIT;
  return a;
}

// Runs all blocks.  In: w.Hn, w.HeA (internal order), pos_in, w.tn / w.te.  Out: w.Hn, He (returned pointer), pos (returned).
// (Round 3 tried the next block's PRE stage on a low-priority side stream behind edge kernel B, to back-fill the CUs its early
// finishers free: the dispatcher interleaves both kernels from the start and the step got slower, 7.41 vs 6.99 ms; removed.)
int run_blocks(const mdx_model_s* m, mdx_graph_s* g, const Ws& w, const float* pos_in, const float** He_final,
               const float** pos_final, hipStream_t s, float* pos_out = nullptr) {
  const int nb = m->cfg.num_blocks;
  const bool upos = m->cfg.update_pos != 0;
  const float* pos = pos_in;
  float* pos_next = w.posA;
  // The per-node table is double-buffered: edge kernel B of block i still reads block i's table (node_ffn columns)
  // after the node kernel has already produced block i+1's table.
  float* NTcur = w.NT;
  float* NTnxt = w.NT2;
  const int split_a = m->matrix_path == MDX_MATRIX_SPLIT_F16 ? EA_SPLIT : 0, split_b = split_a ? EB_SPLIT : 0;
  launch_node(make_nd(m, g, w, -1, 0, ND_PRE, nullptr, NTcur), s);
  for (int i = 0; i < nb; ++i) {
    {
      ProfScope ps(PK_EDGE_A, s);
      LCHK(run_ea(g, make_ea(m, g, w, i, pos, w.HeA, w.HeB, EA_EMB | EA_NODE | EA_FFN | EA_AGG | split_a, NTcur), s));
    }
    // The reduction that is left after edge kernel A's in-kernel sums (combine ~2.5 partial rows per node, the by-right BondFFN sum)
    // is done by the node kernel itself for its 16 nodes: MID of this block + PRE of the next in one launch.
    const bool pre = i + 1 < nb;
    {
      ProfScope ps(PK_NODE, s);
      NodeArgs na = make_nd(m, g, w, i, pre ? i + 1 : -1, ND_MID | (upos ? ND_POSMLP : 0) | (pre ? ND_PRE : 0), NTcur, NTnxt);
      na.P = w.P; na.PR = w.PR; na.FL = w.FL; na.pbase = g->pbase; na.col_ptr = g->col_ptr; na.col_eids = g->col_eids;
      na.SL = w.SL; na.SR = w.SR;
      launch_node(na, s);
    }
    {
      ProfScope ps(PK_EDGE_B, s);
      LCHK(run_eb(g, make_eb(m, g, w, i, pos, w.HeB, w.HeA, EB_EDGE | (upos ? EB_POS : 0) | split_b, NTcur), s));
    }
    if (upos) {
      if (pos_out && i + 1 == nb) pos_next = pos_out;  // the last update lands in the caller's buffer (no copy afterwards)
      launch_seg_reduce(w.Fe, g->row_ptr, nullptr, pos_next, pos, (int)g->N, 3, s);
      pos = pos_next;
      pos_next = (pos_next == w.posA) ? w.posB : w.posA;
    }
    std::swap(NTcur, NTnxt);
  }
  *He_final = w.HeA;
  *pos_final = pos;
  return MDX_OK;
}

}  // namespace

extern "C" int mdx_net_forward(mdx_model_t m, mdx_graph_t g, const float* h_node, const float* pos, const float* h_edge,
                               const float* node_time, const float* edge_time, float* h_node_out, float* pos_out,
                               float* h_edge_out, void* ws, size_t ws_bytes, void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (!h_node || !pos || !h_edge || !node_time || !edge_time) return fail(MDX_ERR_ARG, "null input");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  HIPCHK(hipMemcpyAsync(w.Hn, h_node, (size_t)g->N * MDX_ND * 4, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(w.tn, node_time, (size_t)g->N * 4, hipMemcpyDeviceToDevice, s));
  gather_rows(h_edge, g->int2ref, w.HeA, g->E, 64, s);
  gather_rows(edge_time, g->int2ref, w.te, g->E, 1, s);
  gather_rows(node_time, g->right, w.tnr, g->E, 1, s);
  w.tnr_set = true;
  const float *He, *pf;
  LCHK(run_blocks(m, g, w, pos, &He, &pf, s));
  if (h_node_out) HIPCHK(hipMemcpyAsync(h_node_out, w.Hn, (size_t)g->N * MDX_ND * 4, hipMemcpyDeviceToDevice, s));
  if (pos_out) HIPCHK(hipMemcpyAsync(pos_out, pf, (size_t)g->N * 12, hipMemcpyDeviceToDevice, s));
  if (h_edge_out) gather_rows(He, g->ref2int, h_edge_out, g->E, 64, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_node_block(mdx_model_t m, mdx_graph_t g, int32_t i, const float* x, const float* edge_attr,
                              const float* node_time, float* out, void* ws, size_t ws_bytes, void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (i < 0 || i >= m->cfg.num_blocks || !x || !edge_attr || !node_time || !out) return fail(MDX_ERR_ARG, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  HIPCHK(hipMemcpyAsync(w.Hn, x, (size_t)g->N * MDX_ND * 4, hipMemcpyDeviceToDevice, s));
  gather_rows(edge_attr, g->int2ref, w.HeA, g->E, 64, s);
  // NodeBlock's gate sees node_time[col]: per-edge time = node_time[right]
  gather_rows(node_time, g->right, w.te, g->E, 1, s);
  launch_node(make_nd(m, g, w, -1, i, ND_PRE), s);
  LCHK(run_ea(g, make_ea(m, g, w, i, nullptr, w.HeA, w.HeA, EA_NODE), s));
  launch_seg_reduce(w.M, g->row_ptr, nullptr, w.aggr, nullptr, (int)g->N, 256, s);
  NodeArgs na = make_nd(m, g, w, i, -1, ND_MID | ND_DELTA);
  na.dHn = out;
  launch_node(na, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_edge_block(mdx_model_t m, mdx_graph_t g, int32_t i, const float* h_bond, const float* h_node,
                              const float* bond_time, float* out, void* ws, size_t ws_bytes, void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (i < 0 || i >= m->cfg.num_blocks || !h_bond || !h_node || !bond_time || !out) return fail(MDX_ERR_ARG, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  HIPCHK(hipMemcpyAsync(w.Hn, h_node, (size_t)g->N * MDX_ND * 4, hipMemcpyDeviceToDevice, s));
  gather_rows(h_bond, g->int2ref, w.HeA, g->E, 64, s);
  gather_rows(bond_time, g->int2ref, w.te, g->E, 1, s);
  launch_node(make_nd(m, g, w, -1, i, ND_PRE), s);
  LCHK(run_ea(g, make_ea(m, g, w, i, nullptr, w.HeA, w.HeA, EA_FFN), s));
  launch_seg_reduce(w.FL, g->col_ptr, g->col_eids, w.SL, nullptr, (int)g->N, 64, s);
  launch_seg_reduce(w.FR, g->row_ptr, nullptr, w.SR, nullptr, (int)g->N, 64, s);
  LCHK(run_eb(g, make_eb(m, g, w, i, nullptr, w.HeA, w.HeB, EB_EDGE | EB_DELTA), s));
  gather_rows(w.HeB, g->ref2int, out, g->E, 64, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_bond_ffn(mdx_model_t m, mdx_graph_t g, int32_t i, int32_t side, const float* bond_feat, const float* node_feat,
                            const float* time, float* out, void* ws, size_t ws_bytes, void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (i < 0 || i >= m->cfg.num_blocks || side < 0 || side > 1 || !bond_feat || !node_feat || !time || !out)
    return fail(MDX_ERR_ARG, "bad argument");
  if (g->N != g->E) return fail(MDX_ERR_ARG, "mdx_bond_ffn expects the identity graph (edge e joins node e to itself)");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  // every edge is its own node: the hoisted node_linear / gate node parts become per-edge rows of the node table
  HIPCHK(hipMemcpyAsync(w.Hn, node_feat, (size_t)g->N * MDX_ND * 4, hipMemcpyDeviceToDevice, s));
  gather_rows(bond_feat, g->int2ref, w.HeA, g->E, 64, s);
  gather_rows(time, g->int2ref, w.te, g->E, 1, s);
  launch_node(make_nd(m, g, w, -1, i, ND_PRE), s);
  LCHK(run_ea(g, make_ea(m, g, w, i, nullptr, w.HeA, w.HeA, EA_FFN), s));
  gather_rows(side ? w.FR : w.FL, g->ref2int, out, g->E, 64, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_pos_update(mdx_model_t m, mdx_graph_t g, int32_t i, const float* h_node, const float* h_edge,
                              const float* rel, const float* dist, const float* edge_time, float* out, void* ws,
                              size_t ws_bytes, void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (!m->cfg.update_pos) return fail(MDX_ERR_STATE, "model has no pos blocks");
  if (i < 0 || i >= m->cfg.num_blocks || !h_node || !h_edge || !rel || !dist || !edge_time || !out)
    return fail(MDX_ERR_ARG, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  HIPCHK(hipMemcpyAsync(w.Hn, h_node, (size_t)g->N * MDX_ND * 4, hipMemcpyDeviceToDevice, s));
  gather_rows(h_edge, g->int2ref, w.HeA, g->E, 64, s);
  gather_rows(edge_time, g->int2ref, w.te, g->E, 1, s);
  gather_rows(rel, g->int2ref, w.M, g->E, 3, s);        // M reused as scratch for rel (E,3)
  gather_rows(dist, g->int2ref, w.FL, g->E, 1, s);      // FL reused as scratch for dist (E)
  launch_node(make_nd(m, g, w, i, -1, ND_POSMLP), s);
  EdgeBArgs eb = make_eb(m, g, w, i, nullptr, w.HeA, nullptr, EB_POS);
  eb.rel_in = w.M;
  eb.dist_in = w.FL;
  LCHK(run_eb(g, eb, s));
  launch_seg_reduce(w.Fe, g->row_ptr, nullptr, out, nullptr, (int)g->N, 3, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_segment_sum(mdx_graph_t g, const float* src, int32_t C, int32_t by_right, float* out, void* ws,
                               size_t ws_bytes, void* stream) {
  if (!g || !src || !out) return fail(MDX_ERR_ARG, "null argument");
  if (C != 3 && C != 64 && C != 256) return fail(MDX_ERR_UNSUPPORTED, "C must be 3, 64 or 256");
  if (!ws || ws_bytes < mdx_workspace_bytes(g->N, g->E)) return fail(MDX_ERR_STATE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  const float* rows = src;
  if (!(by_right & 2)) {  // reference edge order -> internal order (bit 1 set: src already is in the graph plan's internal order)
    gather_rows(src, g->int2ref, w.M, g->E, C, s);
    rows = w.M;
  }
  if (by_right & 1)
    launch_seg_reduce(rows, g->col_ptr, g->col_eids, out, nullptr, (int)g->N, C, s);
  else
    launch_seg_reduce(rows, g->row_ptr, nullptr, out, nullptr, (int)g->N, C, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_moldiff_forward(mdx_model_t m, mdx_graph_t g, const float* h_node_pert, const float* pos_pert,
                                   const float* h_edge_pert, const float* h_halfedge_pert, const int64_t* t,
                                   float* pred_node, float* pred_pos, float* pred_halfedge, void* ws, size_t ws_bytes,
                                   void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (m->cfg.kind != MDX_KIND_MOLDIFF) return fail(MDX_ERR_STATE, "not a MolDiff model handle");
  if (g->N == 0) return MDX_OK;  // empty batch: nothing to compute, outputs are empty
  if (!h_node_pert || !pos_pert || (g->E > 0 && !h_edge_pert && !h_halfedge_pert) || !t)
    return fail(MDX_ERR_ARG, "null input");
  if (g->E % 2) return fail(MDX_ERR_ARG, "MolDiff.forward needs E = 2*Eh directed edges");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  const mdx_config& cf = m->cfg;
  EmbedArgs ea{};
  ea.N = (int)g->N; ea.E = (int)g->E; ea.Kn = cf.num_node_types; ea.Ke = cf.num_edge_types; ea.time_dim = cf.time_dim;
  ea.T = std::max(cf.num_timesteps, 1); ea.nd_emb = MDX_ND - cf.time_dim; ea.ed_emb = MDX_ED - cf.time_dim; ea.xn = h_node_pert;
  ea.int2ref = g->int2ref; ea.l = g->left; ea.r = g->right; ea.node_graph = g->node_graph; ea.t = t; ea.Wn = m->Wn;
  ea.We = m->We; ea.toff = m->toff; ea.tcoef = m->tcoef; ea.Hn = w.Hn; ea.He = w.HeA; ea.tn = w.tn; ea.te = w.te;
  if (h_edge_pert) {
    ea.xe = h_edge_pert;
  } else {  // both directions share the half-edge one-hot (model.py:273): the embed kernel folds reference row r onto r mod Eh
    ea.xe = h_halfedge_pert;
    ea.half_rows = (int)g->Eh;
  }
  launch_embed(ea, s);
  const float *He, *pf;
  LCHK(run_blocks(m, g, w, pos_pert, &He, &pf, s, m->cfg.update_pos ? pred_pos : nullptr));
  DecodeArgs da{};
  da.N = (int)g->N; da.Eh = (int)g->Eh; da.Kn = cf.num_node_types; da.Ke = cf.num_edge_types; da.Hn = w.Hn; da.He = He;
  da.ref2int = g->ref2int; da.nodedec = m->nodedec; da.edgedec = m->edgedec; da.pred_node = pred_node;
  da.pred_halfedge = pred_halfedge;
  launch_decode(da, s);
  if (pred_pos && pf != pred_pos) HIPCHK(hipMemcpyAsync(pred_pos, pf, (size_t)g->N * 12, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

static int sample_step_core(mdx_model_t m, mdx_graph_t g, const mdx_tables* tb, const int64_t* t, const int64_t* batch_node,
                            const int64_t* batch_halfedge, const mdx_state* cur, const mdx_state* next, float* pred_node,
                            float* pred_pos, float* pred_halfedge, const float* eps_pos, const float* u_node,
                            const float* u_halfedge, uint8_t* node_cls, uint8_t* halfedge_cls, void* ws, size_t ws_bytes,
                            void* stream) {
  if (!tb || !cur || !next) return fail(MDX_ERR_ARG, "null tables / state");
  if (!m || !g) return fail(MDX_ERR_ARG, "null handle");
  const int N = (int)g->N, Eh = (int)g->Eh;
  if (N == 0) return MDX_OK;
  if (!pred_node || !pred_pos || (Eh > 0 && !pred_halfedge) || !eps_pos || !u_node || (Eh > 0 && !u_halfedge) || !batch_node ||
      (Eh > 0 && !batch_halfedge))
    return fail(MDX_ERR_ARG, "null buffer");
  int rc = mdx_moldiff_forward(m, g, cur->h_node, cur->pos, nullptr, cur->h_halfedge, t, pred_node, pred_pos, pred_halfedge, ws,
                               ws_bytes, stream);
  if (rc != MDX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  const mdx_config& cf = m->cfg;
  if (cf.num_node_types == 8 && cf.num_edge_types == 6) {  // MolDiff's class counts: all five transition launches in one
    StepTransArgs ta{};
    ta.N = N; ta.Eh = Eh; ta.T = cf.num_timesteps; ta.c0 = tb->pos_coef_x0; ta.ct = tb->pos_coef_xt; ta.sd = tb->pos_std;
    ta.node_q = tb->node_q_mats; ta.node_qT1 = tb->node_qT_onestep; ta.edge_q = tb->edge_q_mats; ta.edge_qT1 = tb->edge_qT_onestep;
    ta.t = t; ta.batch_node = batch_node; ta.batch_half = batch_halfedge; ta.pos = cur->pos; ta.pred_pos = pred_pos; ta.eps = eps_pos;
    ta.pred_node = pred_node; ta.log_node = cur->log_node; ta.u_node = u_node; ta.pred_half = pred_halfedge;
    ta.log_half = cur->log_halfedge; ta.u_half = u_halfedge; ta.pos_next = next->pos; ta.log_node_next = next->log_node;
    ta.h_node_next = next->h_node; ta.log_half_next = next->log_halfedge; ta.h_half_next = next->h_halfedge; ta.node_cls = node_cls;
    ta.half_cls = halfedge_cls;
    launch_step_transition(ta, s);
    HIPCHK(hipGetLastError());
    return MDX_OK;
  }
  launch_pos_posterior(tb->pos_coef_x0, tb->pos_coef_xt, tb->pos_std, cur->pos, pred_pos, eps_pos, t, batch_node, N, next->pos, s);
  launch_cat_posterior(tb->node_q_mats, tb->node_qT_onestep, cf.num_node_types, cf.num_timesteps, pred_node, 1, cur->log_node, t,
                       batch_node, N, next->log_node, s);
  launch_gumbel_argmax(next->log_node, u_node, cf.num_node_types, N, nullptr, next->h_node, s, node_cls);
  if (Eh > 0) {
    launch_cat_posterior(tb->edge_q_mats, tb->edge_qT_onestep, cf.num_edge_types, cf.num_timesteps, pred_halfedge, 1,
                         cur->log_halfedge, t, batch_halfedge, Eh, next->log_halfedge, s);
    launch_gumbel_argmax(next->log_halfedge, u_halfedge, cf.num_edge_types, Eh, nullptr, next->h_halfedge, s, halfedge_cls);
  }
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_sample_step(mdx_model_t m, mdx_graph_t g, const mdx_tables* tb, const int64_t* t, const int64_t* batch_node,
                               const int64_t* batch_halfedge, const mdx_state* cur, const mdx_state* next, float* pred_node,
                               float* pred_pos, float* pred_halfedge, const float* eps_pos, const float* u_node,
                               const float* u_halfedge, void* ws, size_t ws_bytes, void* stream) {
  return sample_step_core(m, g, tb, t, batch_node, batch_halfedge, cur, next, pred_node, pred_pos, pred_halfedge, eps_pos, u_node,
                          u_halfedge, nullptr, nullptr, ws, ws_bytes, stream);
}

extern "C" int mdx_sample_step_full(mdx_model_t m, mdx_graph_t g, const mdx_tables* tb, int32_t step, const int64_t* batch_node,
                                    const int64_t* batch_halfedge, const mdx_state* cur, const mdx_state* next, float* pred_node,
                                    float* pred_pos, float* pred_halfedge, const mdx_step_noise* noise, int64_t* t_buf,
                                    uint8_t* node_cls, uint8_t* halfedge_cls, const mdx_guidance* gd, void* ws, size_t ws_bytes,
                                    void* stream) {
  if (!m || !g || !tb || !cur || !next || !noise || !t_buf) return fail(MDX_ERR_ARG, "null argument");
  if (m->cfg.kind != MDX_KIND_MOLDIFF) return fail(MDX_ERR_STATE, "not a MolDiff model handle");
  if (step < 0 || step >= m->cfg.num_timesteps) return fail(MDX_ERR_ARG, "step %d outside [0, %d)", step, m->cfg.num_timesteps);
  if (g->N == 0) return MDX_OK;
  hipStream_t s = (hipStream_t)stream;
  const mdx_config& cf = m->cfg;
  if (noise->draw >= 0) {  // the Philox launch also fills the time tensor
    if (cf.num_node_types > 8 || cf.num_edge_types > 8) return fail(MDX_ERR_ARG, "class counts must be in 1..8");
    launch_philox_noise(noise->seed, noise->draw, g->node_graph, g->node_local, g->he_graph, g->he_local, g->mol_ids, (int)g->N,
                        (int)g->Eh, cf.num_node_types, cf.num_edge_types, noise->eps_pos, noise->u_node, noise->u_halfedge, s, t_buf,
                        step, (int)g->B);
  } else {
    launch_fill_i64(t_buf, step, (int)g->B, s);
  }
  hipStream_t gs = s;  // the stream the guidance chain runs on
  if (gd) {
    if (!gd->predictor || !gd->tape || !gd->logits || !gd->glogits || !gd->delta) return fail(MDX_ERR_ARG, "incomplete mdx_guidance");
    if (gd->side_stream) {
      if (!gd->ws2) return fail(MDX_ERR_ARG, "a concurrent guidance chain needs its own workspace (ws2)");
      if (!g->ev_in) {
        HIPCHK(hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming));
      }
      gs = (hipStream_t)gd->side_stream;
      // everything the chain reads (state, t_buf) and the previous step's use of `delta` is ordered before this point
      HIPCHK(hipEventRecord(g->ev_in, s));
      HIPCHK(hipStreamWaitEvent(gs, g->ev_in, 0));
    }
    void* gws = gd->side_stream ? gd->ws2 : ws;
    const size_t gwb = gd->side_stream ? gd->ws2_bytes : ws_bytes;
    auto chain = [&]() -> int {
      int rc = mdx_bondpred_forward(gd->predictor, g, cur->h_node, cur->pos, t_buf, gd->logits, gws, gwb, gd->tape, gd->tape_bytes, gs);
      if (rc != MDX_OK) return rc;
      const int Kb = gd->predictor->cfg.num_edge_types;
      launch_uncertainty_grad(gd->logits, Kb, (int)g->Eh, gd->glogits, gs);
      return mdx_bondpred_backward(gd->predictor, g, cur->pos, gd->glogits, -gd->scale, gd->delta, gws, gwb, gd->tape,
                                   gd->tape_bytes, gs);
    };
    if (gd->side_stream) {  // enqueue the chain first: it then runs under the denoiser launched below
      int rc = chain();
      if (rc != MDX_OK) return rc;
      HIPCHK(hipEventRecord(g->ev_done, gs));
    }
    int rc = sample_step_core(m, g, tb, t_buf, batch_node, batch_halfedge, cur, next, pred_node, pred_pos, pred_halfedge,
                              noise->eps_pos, noise->u_node, noise->u_halfedge, node_cls, halfedge_cls, ws, ws_bytes, stream);
    if (rc != MDX_OK) return rc;
    if (gd->side_stream) {
      HIPCHK(hipStreamWaitEvent(s, g->ev_done, 0));
    } else {  // in line: the predictor reuses the denoiser's workspace after it
      rc = chain();
      if (rc != MDX_OK) return rc;
    }
    launch_add_inplace(next->pos, gd->delta, 3 * (int)g->N, s);
    HIPCHK(hipGetLastError());
    return MDX_OK;
  }
  return sample_step_core(m, g, tb, t_buf, batch_node, batch_halfedge, cur, next, pred_node, pred_pos, pred_halfedge, noise->eps_pos,
                          noise->u_node, noise->u_halfedge, node_cls, halfedge_cls, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------
// bond predictor: forward with a per-block tape, and the data-gradient backward w.r.t. positions
// ------------------------------------------------------------------------------------------------
namespace {
struct TapeBlock {
  float *Hep, *Hn, *H, *NT, *aggr, *SL, *SR;
  float *SG, *HE, *M;  // (E,256): sigmoid(gate), edge_net output, gated message -- read back by edge_bwd instead of recomputed
  float *BL[2], *H1[2], *O[2];  // BondFFN intermediates (EdgeAArgs tBL / tH1 / tO), left and right: 2.5 KB per edge and block
};
struct Tape {
  std::vector<TapeBlock> b;
  float *HeF, *HnF, *te;                       // final states + per-edge time
  float *GGX, *GNL0, *GNL1, *GGXS0, *GGXS1, *GBN, *gdist, *tmpE3;  // backward scratch
  size_t bytes;
};
size_t tape_layout(int64_t N, int64_t E, int nb, char* base, Tape* t) {
  size_t off = 0;
  auto take = [&](size_t nfloat) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += ((nfloat * 4 + 255) / 256) * 256;
    return p;
  };
  const size_t n = (size_t)std::max<int64_t>(N, 1), e = (size_t)std::max<int64_t>(E, 1);
  Tape tp;
  tp.b.resize(nb);
  for (int i = 0; i < nb; ++i) {
    TapeBlock& k = tp.b[i];
    k.Hep = take(e * 64); k.Hn = take(n * MDX_ND); k.H = take(n * MDX_ND); k.NT = take(n * MDX_NTW);
    k.aggr = take(n * MDX_ND); k.SL = take(n * 64); k.SR = take(n * 64);
    k.SG = take(e * MDX_ND); k.HE = take(e * MDX_ND); k.M = take(e * MDX_ND);
    for (int sd = 0; sd < 2; ++sd) { k.BL[sd] = take(e * 128); k.H1[sd] = take(e * 128); k.O[sd] = take(e * 64); }
  }
  tp.HeF = take(e * 64); tp.HnF = take(n * MDX_ND); tp.te = take(e);
  tp.GGX = take(e * MDX_ND); tp.GNL0 = take(e * 128); tp.GNL1 = take(e * 128); tp.GGXS0 = take(e * 32);
  tp.GGXS1 = take(e * 32); tp.GBN = take(e * 128); tp.gdist = take(e); tp.tmpE3 = take(e * 3);
  tp.bytes = off;
  if (t) *t = std::move(tp);
  return off;
}
}  // namespace

extern "C" size_t mdx_bondpred_tape_bytes(int64_t N, int64_t E, int32_t num_blocks) {
  return tape_layout(N, E, num_blocks, nullptr, nullptr);
}

extern "C" int mdx_bondpred_forward(mdx_model_t m, mdx_graph_t g, const float* h_node, const float* pos, const int64_t* t,
                                    float* logits, void* ws, size_t ws_bytes, void* tape, size_t tape_bytes, void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (m->cfg.kind != MDX_KIND_BONDPRED) return fail(MDX_ERR_STATE, "not a BondPredictor model handle");
  if (!h_node || !pos || !t || !logits) return fail(MDX_ERR_ARG, "null input");
  if (g->E % 2) return fail(MDX_ERR_ARG, "BondPredictor.forward needs E = 2*Eh directed edges");
  const int nb = m->cfg.num_blocks;
  if (tape && (tape_bytes < mdx_bondpred_tape_bytes(g->N, g->E, nb) || (reinterpret_cast<uintptr_t>(tape) & 255)))
    return fail(MDX_ERR_STATE, "tape too small or misaligned: need %zu bytes", mdx_bondpred_tape_bytes(g->N, g->E, nb));
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  Tape tp;
  if (tape) tape_layout(g->N, g->E, nb, (char*)tape, &tp);
  const mdx_config& cf = m->cfg;
  EmbedArgs ea{};
  ea.N = (int)g->N; ea.E = (int)g->E; ea.Kn = cf.num_node_types; ea.Ke = cf.num_edge_types; ea.time_dim = cf.time_dim;
  ea.T = std::max(cf.num_timesteps, 1); ea.nd_emb = MDX_ND - cf.time_dim; ea.ed_emb = MDX_ED - cf.time_dim; ea.xn = h_node; ea.xe = nullptr;
  ea.zero_time = cf.num_timesteps == 0;  // time-free predictor: the reference replaces t by zeros (bond_predictor.py:141-144), whatever the caller passes
  ea.int2ref = g->int2ref; ea.l = g->left; ea.r = g->right; ea.node_graph = g->node_graph; ea.t = t; ea.Wn = m->Wn;
  ea.We = m->We; ea.toff = m->toff; ea.tcoef = m->tcoef; ea.Hn = w.Hn; ea.He = w.HeA; ea.tn = w.tn;
  ea.te = tape ? tp.te : w.te;
  launch_embed(ea, s);
  Ws wr = w;
  wr.te = ea.te;
  const size_t nHn = (size_t)g->N * MDX_ND * 4;
  const int split_a = m->matrix_path == MDX_MATRIX_SPLIT_F16 ? EA_SPLIT : 0, split_b = split_a ? EB_SPLIT : 0;
  for (int i = 0; i < nb; ++i) {
    // same sequence as run_blocks (update_pos = false), with the per-block outputs redirected into the tape
    Ws wi = wr;
    float* Hep = wr.HeB;
    if (tape) {
      const TapeBlock& k = tp.b[i];
      wi.H = k.H; wi.NT = k.NT; wi.aggr = k.aggr; wi.SL = k.SL; wi.SR = k.SR; wi.M = k.M;
      Hep = k.Hep;
      HIPCHK(hipMemcpyAsync(k.Hn, wr.Hn, nHn, hipMemcpyDeviceToDevice, s));
    }
    // tape + in-kernel sums: the reduction, MID of this block and PRE of the next are ONE node launch as in run_blocks (round 3; it
    // used to be three plus the reduction kernel), with every output pointed straight at its tape buffer
    const bool fused_tape = tape != nullptr;
    if (i == 0) launch_node(make_nd(m, g, wi, -1, i, ND_PRE, nullptr, wi.NT), s);
    {
      EdgeAArgs ea_args = make_ea(m, g, wi, i, pos, wr.HeA, Hep, EA_EMB | EA_NODE | EA_FFN | EA_AGG, wi.NT);
      if (tape) {  // the launch also writes the guidance tape (template flags of the kernel)
        ea_args.flags |= EA_TAPE | EA_TAPE_FFN;
        ea_args.tSG = tp.b[i].SG;
        ea_args.tHE = tp.b[i].HE;
        ea_args.M = tp.b[i].M;  // the backward reads the gated message back (with EA_AGG it is no longer the reduction's input)
        for (int sd = 0; sd < 2; ++sd) { ea_args.tBL[sd] = tp.b[i].BL[sd]; ea_args.tH1[sd] = tp.b[i].H1[sd]; ea_args.tO[sd] = tp.b[i].O[sd]; }
      }
      ea_args.flags |= split_a;  // split float16 matrix path: the same launch on mdx_edge2s.hip
      { ProfScope ps(PK_EDGE_A, s); LCHK(run_ea(g, ea_args, s)); }
    }
    if (fused_tape) {
      const bool pre = i + 1 < nb;
      NodeArgs na = make_nd(m, g, wi, i, pre ? i + 1 : -1, ND_MID | (pre ? ND_PRE : 0), wi.NT, pre ? tp.b[i + 1].NT : nullptr);
      if (pre) na.H = tp.b[i + 1].H;
      na.P = wi.P; na.PR = wi.PR; na.FL = wi.FL; na.pbase = g->pbase; na.col_ptr = g->col_ptr; na.col_eids = g->col_eids;
      na.SL = wi.SL; na.SR = wi.SR; na.aggr_out = wi.aggr;
      launch_node(na, s);
      LCHK(run_eb(g, make_eb(m, g, wi, i, pos, Hep, wr.HeA, EB_EDGE | split_b, wi.NT), s));
      continue;
    }
    // no tape: the reduction left after the in-kernel sums as a launch of its own, then MID of this block fused with the next
    // block's PRE (tables double-buffered like run_blocks)
    launch_seg_reduce_block2(wi.P, wi.PR, wi.FL, g->pbase, g->col_ptr, g->col_eids, wi.aggr, wi.SL, wi.SR, (int)g->N, s);
    float* NTn = (wr.NT == w.NT) ? w.NT2 : w.NT;
    launch_node(make_nd(m, g, wi, i, i + 1 < nb ? i + 1 : -1, ND_MID | (i + 1 < nb ? ND_PRE : 0), wi.NT, NTn), s);
    LCHK(run_eb(g, make_eb(m, g, wi, i, pos, Hep, wr.HeA, EB_EDGE | split_b, wi.NT), s));
    wr.NT = NTn;
  }
  if (tape) {
    HIPCHK(hipMemcpyAsync(tp.HnF, wr.Hn, nHn, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(tp.HeF, wr.HeA, (size_t)g->E * 64 * 4, hipMemcpyDeviceToDevice, s));
  }
  BondDecArgs da{};
  da.Eh = (int)g->Eh; da.Ke = cf.num_edge_types; da.He = wr.HeA; da.Hn = wr.Hn; da.ref2int = g->ref2int; da.left = g->left;
  da.right = g->right; da.logits = logits; da.w = m->dec;
  launch_bond_decode(da, false, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

// gpos (N,3) = scale * dL/dpos given glogits = dL/dlogits (Eh,Ke); needs the tape of the matching forward.
extern "C" int mdx_bondpred_backward(mdx_model_t m, mdx_graph_t g, const float* pos, const float* glogits, float scale,
                                     float* gpos, void* ws, size_t ws_bytes, void* tape, size_t tape_bytes, void* stream) {
  CHECK_READY(m, g, ws, ws_bytes);
  if (m->cfg.kind != MDX_KIND_BONDPRED) return fail(MDX_ERR_STATE, "not a BondPredictor model handle");
  if (!pos || !glogits || !gpos || !tape) return fail(MDX_ERR_ARG, "null argument");
  const int nb = m->cfg.num_blocks;
  if (tape_bytes < mdx_bondpred_tape_bytes(g->N, g->E, nb) || (reinterpret_cast<uintptr_t>(tape) & 255))
    return fail(MDX_ERR_STATE, "tape too small or misaligned");
  hipStream_t s = (hipStream_t)stream;
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  Tape tp;
  tape_layout(g->N, g->E, nb, (char*)tape, &tp);
  const mdx_config& cf = m->cfg;
  const int N = (int)g->N, E = (int)g->E;
  // scratch mapping onto the (now dead) forward workspace
  float *gHn = w.Hn, *GNT = w.NT, *gH = w.H, *gHe = w.HeA, *GU = w.FL, *GHEP = w.FR, *GH = w.M;
  HIPCHK(hipMemsetAsync(tp.gdist, 0, (size_t)std::max(E, 1) * 4, s));
  BondDecArgs da{};
  da.Eh = (int)g->Eh; da.Ke = cf.num_edge_types; da.He = tp.HeF; da.Hn = tp.HnF; da.ref2int = g->ref2int; da.left = g->left;
  da.right = g->right; da.glogits = glogits; da.gHe = gHe; da.GBN = tp.GBN; da.w = m->dec;
  launch_bond_decode(da, true, s);
  launch_seg_reduce_ld(tp.GBN, g->row_ptr, g->half_of_int, gHn, MDX_ND, N, 256, s);
  for (int i = nb - 1; i >= 0; --i) {
    const TapeBlock& k = tp.b[i];
    NodeBwdArgs nt{};
    nt.N = N; nt.flags = NB_TAIL; nt.gHn = gHn; nt.Hn = k.Hn; nt.NTin = k.NT; nt.aggr = k.aggr; nt.GNT = GNT; nt.gH = gH;
    nt.w = m->blocks[i].nd; nt.wt = m->nbw[i]; nt.ws = m->blocks[i].nds; nt.wts = m->nbws[i];
    const int nb_split = m->matrix_path == MDX_MATRIX_SPLIT_F16 ? NB_SPLIT : 0;
    nt.flags |= nb_split;
    launch_node_bwd(nt, s);
    // EdgeBlock-tail backward: for the last block a launch of its own (its dL/dHe'' comes from the decoder); for every other
    // block it ran inside block i + 1's edge kernel, fused behind that block's edge_embs backward (GU, GHEP are ready)
    const bool split_bwd = m->matrix_path == MDX_MATRIX_SPLIT_F16;
    int* wq = wq_for(g, s);
    if (i == nb - 1) {
      EdgeTailBwdArgs et{};
      et.E = E; et.l = g->left; et.r = g->right; et.te = tp.te; et.Hep = k.Hep; et.gHe = gHe; et.SL = k.SL; et.SR = k.SR;
      et.NT = k.NT; et.GU = GU; et.GHEP = GHEP; et.w = m->blocks[i].eb; et.WselfT = m->ebw[i].WselfT; et.WoutT = m->ebw[i].WoutT;
      et.sWselfT = m->ebw[i].s.WselfT; et.sWoutT = m->ebw[i].s.WoutT; et.wq = wq;
      et.ssWselfT = m->ebw[i].ss.WselfT; et.ssWoutT = m->ebw[i].ss.WoutT;
      et.split = split_bwd ? 1 : 0;
      if (split_bwd) launch_edge_tail_bwd2s(et, s);
      else launch_edge_tail_bwd2(et, s);
    }
    launch_seg_reduce_tail_block(GU, g->row_ptr, g->col_ptr, g->col_eids, GNT, N, s);
    EdgeBwdArgs eb{};
    eb.E = E; eb.l = g->left; eb.r = g->right; eb.te = tp.te; eb.pos = pos; eb.soff = m->soff; eb.scoef = m->scoef;
    eb.cutoff = cf.cutoff; eb.smear_start = m->smear_start; eb.Hep = k.Hep; eb.GHEP = GHEP; eb.H = k.H; eb.NT = k.NT; eb.GNT = GNT;
    eb.SG = k.SG; eb.HE = k.HE; eb.M = k.M;
    for (int sd = 0; sd < 2; ++sd) { eb.BL[sd] = k.BL[sd]; eb.H1[sd] = k.H1[sd]; eb.O[sd] = k.O[sd]; }
    eb.gdist = tp.gdist; eb.GH = GH; eb.GGX = tp.GGX; eb.GNL[0] = tp.GNL0; eb.GNL[1] = tp.GNL1; eb.GGXS[0] = tp.GGXS0;
    eb.GGXS[1] = tp.GGXS1; eb.w = m->blocks[i].ea; eb.wt = m->ebw[i]; eb.wq = wq;
    eb.split = split_bwd ? 1 : 0;
    eb.fuse_tail = i > 0;
    if (i > 0) {
      const TapeBlock& kp = tp.b[i - 1];
      const EdgeBW& wb = m->blocks[i - 1].eb;
      eb.tHep = kp.Hep; eb.tSL = kp.SL; eb.tSR = kp.SR; eb.tNT = kp.NT; eb.tGU = GU; eb.tGHEP = GHEP;
      eb.tWself = split_bwd ? wb.ss.Wself : wb.s.Wself;
      eb.tWoutT = split_bwd ? m->ebw[i - 1].ss.WoutT : m->ebw[i - 1].s.WoutT;
      eb.tWselfT = split_bwd ? m->ebw[i - 1].ss.WselfT : m->ebw[i - 1].s.WselfT;
      eb.tbself = wb.bself; eb.tlng = wb.lng; eb.tlnb = wb.lnb;
    }
    eb.units_r = g->units_r; eb.epo_r = g->epo_r; eb.col_eids = g->col_eids; eb.col_left = g->col_left; eb.col_right = g->col_right;
    eb.nunits_r = (int)g->nunits_r;
    {
      ProfScope ps(PK_EDGE_BWD, s);
      if (split_bwd) {
        if (launch_edge_bwd2s(eb, s) != 0) return fail(MDX_ERR_STATE, "split float16 backward: the forward tape holds no BondFFN intermediates");
      } else launch_edge_bwd2(eb, s);
    }
    {
      SegBwdArgs sr{};
      sr.N = N; sr.row_ptr = g->row_ptr; sr.col_ptr = g->col_ptr; sr.col_eids = g->col_eids; sr.pbase_r = g->pbase_r; sr.GH = GH; sr.GGX = tp.GGX;
      sr.GNL0 = tp.GNL0; sr.GNL1 = tp.GNL1; sr.GGXS0 = tp.GGXS0; sr.GGXS1 = tp.GGXS1; sr.gH = gH; sr.GNT = GNT;
      launch_seg_reduce_bwd_block(sr, s);
    }
    nt.flags = NB_PRE | nb_split;
    launch_node_bwd(nt, s);
  }
  launch_dist_to_pos(tp.gdist, pos, g->left, g->right, g->row_ptr, g->col_ptr, g->col_eids, tp.tmpE3, nullptr, gpos, scale, N,
                     E, cf.cutoff, s);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

// ------------------------------------------------------------------------------------------------
// transitions
// ------------------------------------------------------------------------------------------------
extern "C" int mdx_pos_posterior(const float* c0, const float* ct, const float* sd, const float* x_t, const float* x_recon,
                                 const float* eps, const int64_t* t, const int64_t* batch, int64_t n, float* out,
                                 void* stream) {
  if (n < 0 || (n > 0 && (!c0 || !ct || !sd || !x_t || !x_recon || !eps || !t || !batch || !out)))
    return fail(MDX_ERR_ARG, "bad argument");
  launch_pos_posterior(c0, ct, sd, x_t, x_recon, eps, t, batch, (int)n, out, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_gauss_posterior(const float* c0, const float* ct, const float* sd, const float* x_t, const float* x_recon,
                                   const float* eps, const int64_t* t, const int64_t* batch, int64_t n, int32_t C, float* out,
                                   void* stream) {
  if (n < 0 || C < 1 || (n > 0 && (!c0 || !ct || !sd || !x_t || !x_recon || !eps || !t || !batch || !out)))
    return fail(MDX_ERR_ARG, "bad argument");
  launch_gauss_posterior(c0, ct, sd, x_t, x_recon, eps, t, batch, (int)n, (int)C, out, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_cat_posterior(const float* q_mats, const float* qT, int32_t K, int32_t T, const float* in0,
                                 int32_t is_logits, const float* log_vt, const int64_t* t, const int64_t* batch, int64_t n,
                                 float* out, void* stream) {
  if (K < 2 || K > 8) return fail(MDX_ERR_UNSUPPORTED, "K must be in 2..8");
  if (n < 0 || (n > 0 && (!q_mats || !qT || !in0 || !log_vt || !t || !batch || !out))) return fail(MDX_ERR_ARG, "bad argument");
  launch_cat_posterior(q_mats, qT, K, T, in0, is_logits, log_vt, t, batch, (int)n, out, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_op_cat_add_noise(const float* q_mats, int32_t K, int32_t T, const int64_t* v, const int64_t* t, const int64_t* batch,
                                    const float* u, int64_t n, float log_off, float* onehot, float* log_vt, float* log_v0, void* stream) {
  if (K < 2 || K > 8) return fail(MDX_ERR_UNSUPPORTED, "K must be in 2..8");
  if (n < 0 || T < 1 || (n > 0 && (!q_mats || !v || !t || !batch || !u || !onehot || !log_vt || !log_v0))) return fail(MDX_ERR_ARG, "bad argument");
  launch_cat_add_noise(q_mats, K, v, t, batch, u, (int)n, log_off, onehot, log_vt, log_v0, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_op_cat_loss(const float* q_mats, const float* qT, int32_t K, int32_t T, const float* logits, const float* log_vt,
                               const float* log_v0, const int64_t* t, const int64_t* batch, int64_t n, float* row_loss, float* dlogits,
                               void* stream) {
  if (K < 2 || K > 8) return fail(MDX_ERR_UNSUPPORTED, "K must be in 2..8");
  if (n < 0 || T < 1 || (n > 0 && (!q_mats || !qT || !logits || !log_vt || !log_v0 || !t || !batch || !row_loss || !dlogits)))
    return fail(MDX_ERR_ARG, "bad argument");
  launch_cat_loss(q_mats, qT, K, logits, log_vt, log_v0, t, batch, (int)n, row_loss, dlogits, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_gumbel_argmax(const float* logits, const float* u, int32_t K, int64_t n, int64_t* cls, float* onehot,
                                 void* stream) {
  if (K < 1 || n < 0 || (n > 0 && (!logits || !u))) return fail(MDX_ERR_ARG, "bad argument");
  launch_gumbel_argmax(logits, u, K, (int)n, cls, onehot, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_prior_draw(const double* logits64, int32_t K, const void* u, int32_t u_is_f64, int64_t n, int64_t* cls,
                              float* onehot, float* log_onehot, float log_off, uint8_t* cls8, void* stream) {
  if (K < 1 || K > 8 || n < 0 || !logits64 || (n > 0 && !u)) return fail(MDX_ERR_ARG, "bad argument");
  launch_prior_draw(logits64, K, u, u_is_f64 != 0, (int)n, cls, onehot, log_onehot, log_off, cls8, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

void launch_decode_output(const float* pred_node, int Kn, const float* pred_pos, const float* pred_halfedge, int Ke, int N,
                          int Eh, int B, const int* node_ptr, const int* he_ptr, const int* ref2int, const int* left,
                          const int* right, int num_element, int num_bond_types, int* scratch_i, float* scratch_f,
                          int* atom_type, float* atom_prob, float* atom_pos, int* n_atoms, int* bond_type,
                          float* bond_prob, int* bond_index, int* n_bonds, hipStream_t s);

extern "C" int mdx_decode_output(mdx_graph_t g, const float* pred_node, int32_t Kn, const float* pred_pos,
                                 const float* pred_halfedge, int32_t Ke, int32_t num_element, int32_t num_bond_types,
                                 int32_t* atom_type, float* atom_prob, float* atom_pos, int32_t* n_atoms,
                                 int32_t* bond_type, float* bond_prob, int32_t* bond_index, int32_t* n_bonds, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (!g || !pred_node || !pred_pos || !pred_halfedge || !atom_type || !atom_prob || !atom_pos || !n_atoms || !bond_type ||
      !bond_prob || !bond_index || !n_bonds)
    return fail(MDX_ERR_ARG, "null argument");
  if (Kn < 1 || Ke < 1) return fail(MDX_ERR_ARG, "bad class count");
  if (!ws || ws_bytes < mdx_workspace_bytes(g->N, g->E)) return fail(MDX_ERR_STATE, "workspace too small");
  Ws w;
  ws_layout(g->N, g->E, (char*)ws, &w);
  if ((size_t)(2 * g->N + g->Eh) > (size_t)std::max<int64_t>(g->E, 1) * MDX_ND ||
      (size_t)(g->N + g->Eh) > (size_t)std::max<int64_t>(g->E, 1) * 64)
    return fail(MDX_ERR_UNSUPPORTED, "graph too sparse for the decode scratch layout");
  // scratch: ints in M (>= 2N + Eh words), floats in HeA (>= N + Eh words)
  launch_decode_output(pred_node, Kn, pred_pos, pred_halfedge, Ke, (int)g->N, (int)g->Eh, (int)g->B, g->node_ptr, g->he_ptr,
                       g->ref2int, g->left, g->right, num_element, num_bond_types, reinterpret_cast<int*>(w.M), w.HeA,
                       atom_type, atom_prob, atom_pos, n_atoms, bond_type, bond_prob, bond_index, n_bonds,
                       (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_guidance_uncertainty_grad(const float* logits, int32_t K, int64_t n, float* glogits, void* stream) {
  if (K < 1 || n < 0 || (n > 0 && (!logits || !glogits))) return fail(MDX_ERR_ARG, "bad argument");
  launch_uncertainty_grad(logits, K, (int)n, glogits, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_add_inplace(float* dst, const float* src, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!dst || !src))) return fail(MDX_ERR_ARG, "bad argument");
  launch_add_inplace(dst, src, (int)n, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}

extern "C" int mdx_noise(mdx_graph_t g, uint64_t seed, int32_t draw, int32_t Kn, int32_t Ke, float* eps_pos, float* u_node,
                         float* u_halfedge, void* stream) {
  if (!g) return fail(MDX_ERR_ARG, "null graph");
  if (Kn < 1 || Kn > 8 || Ke < 1 || Ke > 8) return fail(MDX_ERR_ARG, "class counts must be in 1..8");
  launch_philox_noise(seed, draw, g->node_graph, g->node_local, g->he_graph, g->he_local, g->mol_ids, (int)g->N,
                      (int)g->Eh, Kn, Ke, eps_pos, u_node, u_halfedge, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return MDX_OK;
}
