// On-device counterpart of the harness step that follows the hot path:
// `seperate_outputs` (utils/sample.py:4-30) + `FeaturizeMol.decode_output` (utils/transforms.py:65-122).
// Instead of copying the packed predictions to the host and looping over molecules in Python, two kernels
// produce, per molecule, the compacted atom list (mask-type atoms dropped, survivors re-indexed) and the
// compacted bond list (half-edges whose arg-max type is a real bond and whose two atoms survived), written
// in place at the molecule's original offsets with per-molecule counts.  Order inside a molecule is preserved,
// so the host only slices (and mirrors the bonds to both directions like the reference's wire format).
#include "mdx_kernels.h"

namespace {

// arg-max class (first maximum) and its soft-max probability for every row of an (n, K) logit matrix
__global__ void decode_rows_kernel(const float* __restrict__ logits, int K, int n, int* __restrict__ cls,
                                   float* __restrict__ prob) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int best = 0;
  float m = logits[(size_t)i * K];
  for (int k = 1; k < K; ++k) {
    const float v = logits[(size_t)i * K + k];
    if (v > m) { m = v; best = k; }
  }
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf(logits[(size_t)i * K + k] - m);
  cls[i] = best;
  prob[i] = 1.0f / s;
}

// one workgroup (256 threads) per molecule
__global__ __launch_bounds__(256) void decode_compact_kernel(
    const int* __restrict__ node_ptr, const int* __restrict__ he_ptr, const int* __restrict__ ref2int,
    const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ ncls,
    const float* __restrict__ nprob, const float* __restrict__ pos, const int* __restrict__ hcls,
    const float* __restrict__ hprob, int num_element, int num_bond_types, int* __restrict__ node_new,
    int* __restrict__ atom_type, float* __restrict__ atom_prob, float* __restrict__ atom_pos, int* __restrict__ n_atoms,
    int* __restrict__ bond_type, float* __restrict__ bond_prob, int* __restrict__ bond_i, int* __restrict__ bond_j,
    int* __restrict__ n_bonds) {
  __shared__ int wave_cnt[4];
  __shared__ int base;
  const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = node_ptr[m], n1 = node_ptr[m + 1], h0 = he_ptr[m], h1 = he_ptr[m + 1];
  // ---- atoms: ordered compaction in chunks of 256 -------------------------------------------------
  if (tid == 0) base = 0;
  __syncthreads();
  for (int c0 = n0; c0 < n1; c0 += 256) {
    const int v = c0 + tid;
    const bool keep = v < n1 && ncls[v] < num_element;
    const unsigned long long mask = __ballot(keep);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(mask);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (v < n1) node_new[v] = keep ? off + before : -1;
    if (keep) {
      const int o = n0 + off + before;
      atom_type[o] = ncls[v];
      atom_prob[o] = nprob[v];
      atom_pos[3 * (size_t)o + 0] = pos[3 * (size_t)v + 0];
      atom_pos[3 * (size_t)o + 1] = pos[3 * (size_t)v + 1];
      atom_pos[3 * (size_t)o + 2] = pos[3 * (size_t)v + 2];
    }
    __syncthreads();
    if (tid == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) {
    n_atoms[m] = base;
    base = 0;
  }
  __syncthreads();  // node_new of this molecule is complete (written by this workgroup) and visible
  // ---- bonds ---------------------------------------------------------------------------------------------
  for (int c0 = h0; c0 < h1; c0 += 256) {
    const int h = c0 + tid;
    bool keep = false;
    int ni = 0, nj = 0, t = 0;
    if (h < h1) {
      t = hcls[h];
      const int e = ref2int[h];
      ni = node_new[left[e]];
      nj = node_new[right[e]];
      keep = (t > 0) && (t <= num_bond_types) && ni >= 0 && nj >= 0;
    }
    const unsigned long long mask = __ballot(keep);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(mask);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (keep) {
      const int o = h0 + off + before;
      bond_type[o] = t;
      bond_prob[o] = hprob[h];
      bond_i[o] = ni;
      bond_j[o] = nj;
    }
    __syncthreads();
    if (tid == 0) base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) n_bonds[m] = base;
}

}  // namespace

void launch_decode_output(const float* pred_node, int Kn, const float* pred_pos, const float* pred_halfedge, int Ke, int N,
                          int Eh, int B, const int* node_ptr, const int* he_ptr, const int* ref2int, const int* left,
                          const int* right, int num_element, int num_bond_types, int* scratch_i, float* scratch_f,
                          int* atom_type, float* atom_prob, float* atom_pos, int* n_atoms, int* bond_type,
                          float* bond_prob, int* bond_index, int* n_bonds, hipStream_t s) {
  // scratch_i: ncls (N) | hcls (Eh) | node_new (N);  scratch_f: nprob (N) | hprob (Eh)
  int* ncls = scratch_i;
  int* hcls = scratch_i + N;
  int* node_new = scratch_i + N + Eh;
  float* nprob = scratch_f;
  float* hprob = scratch_f + N;
  if (N > 0) hipLaunchKernelGGL(decode_rows_kernel, dim3((N + 255) / 256), dim3(256), 0, s, pred_node, Kn, N, ncls, nprob);
  if (Eh > 0)
    hipLaunchKernelGGL(decode_rows_kernel, dim3((Eh + 255) / 256), dim3(256), 0, s, pred_halfedge, Ke, Eh, hcls, hprob);
  if (B > 0)
    hipLaunchKernelGGL(decode_compact_kernel, dim3(B), dim3(256), 0, s, node_ptr, he_ptr, ref2int, left, right, ncls, nprob,
                       pred_pos, hcls, hprob, num_element, num_bond_types, node_new, atom_type, atom_prob, atom_pos, n_atoms,
                       bond_type, bond_prob, bond_index, bond_index + Eh, n_bonds);
}
