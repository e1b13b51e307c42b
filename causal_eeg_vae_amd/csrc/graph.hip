// Graph structure kernels: edge list -> canonical CSR (by dst, then src, then edge id), its transpose,
// gcn_norm values, and the uniform negative sampler.
//
// Stands behind upstream PyG gcn_norm / add_remaining_self_loops / negative_sampling (un-vendored,
// unpinned: /root/reference/environment.yml:13); edge direction as built at scripts/dataset.py:147-153.
// The (dst, src, edge id) ordering is a stable LSD radix sort of 64-bit keys (rocPRIM device radix sort,
// header-only, compiled into this library; temporary storage comes from the caller's workspace).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "ceeg_internal.h"

namespace {

constexpr uint64_t KEY_INVALID = ~0ull;

// Keys are (dst << nb | src) with nb = key_bits(N): the sort then runs over 2 nb bits (40 at N = 1 M: five 8-bit passes) instead of
// the 32 + nb of a (dst << 32 | src) layout (seven passes).  The all-ones key of a dropped edge has a dst field >= N and sorts last.
__global__ void csr_edge_keys_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int64_t N, unsigned nb,
                                     int self_loops, uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                     int32_t* __restrict__ loop_eid) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = src[e], d = dst[e];
  uint64_t key = KEY_INVALID;
  if (s >= 0 && s < N && d >= 0 && d < N) {
    if (self_loops && s == d) {
      atomicMax(&loop_eid[s], (int32_t)e);  // integer max: order-independent -> deterministic
    } else {
      key = ((uint64_t)d << nb) | (uint64_t)s;
    }
  }
  keys[e] = key;
  vals[e] = (int32_t)e;
}

__global__ void csr_loop_keys_kernel(int64_t E, int64_t N, unsigned nb, uint64_t* __restrict__ keys, int32_t* __restrict__ vals,
                                     const int32_t* __restrict__ loop_eid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  keys[E + i] = ((uint64_t)i << nb) | (uint64_t)i;
  vals[E + i] = loop_eid[i];  // last existing self-loop's edge id (its weight survives) or -1
}

// sorted keys -> rowptr / col / perm.  Thread p closes every row between key[p-1] and key[p].
__global__ void csr_finish_kernel(const uint64_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t M, int64_t N, unsigned nb,
                                  int32_t* __restrict__ rowptr, int32_t* __restrict__ col, int32_t* __restrict__ perm,
                                  int64_t* __restrict__ nnz_out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > M) return;
  int64_t d_cur = N, d_prev = -1;
  if (p < M) {
    const uint64_t k = keys[p];
    const int64_t d = (int64_t)(k >> nb);
    d_cur = d < N ? d : N;
    if (d < N) {
      col[p] = (int32_t)(k & ((1ull << nb) - 1));
      perm[p] = vals[p];
    }
  }
  if (p > 0) {
    const int64_t d = (int64_t)(keys[p - 1] >> nb);
    d_prev = d < N ? d : N;
  }
  for (int64_t r = d_prev + 1; r <= d_cur; ++r) {
    rowptr[r] = (int32_t)p;
    if (r == N && nnz_out) *nnz_out = p;
  }
}

// Transpose by ONE stable sort on the column alone: the input CSR is ordered by (row, col), so a stable sort of its entries by
// col leaves them ordered by (col, row, position) - the transposed CSR's order - with 32-bit keys over nb bits (three 8-bit passes
// at N = 1 M) instead of 64-bit (col, row) keys over 2 nb.  rowid[p] = row of entry p; positions no row covers keep key 0xffffffff.
__global__ void csr_transpose_keys32_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N,
                                            uint32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t* __restrict__ rowid) {
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);  // 16 lanes per row (the mean row of an event graph)
  if (r >= N) return;
  const int lane = threadIdx.x & 15;
  const int lo = rowptr[r], hi = rowptr[r + 1];
  for (int p = lo + lane; p < hi; p += 16) {
    keys[p] = (uint32_t)col[p];
    vals[p] = p;
    rowid[p] = (int32_t)r;
  }
}
__global__ void csr_transpose_finish32_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                              const int32_t* __restrict__ rowid, int64_t M, int64_t N, int32_t* __restrict__ t_rowptr,
                                              int32_t* __restrict__ t_col, int32_t* __restrict__ t_perm) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > M) return;
  int64_t cur = N, prev = -1;
  if (p < M) {
    const int64_t c = (int64_t)keys[p];
    cur = c < N ? c : N;
    if (c < N) {
      const int32_t v = vals[p];
      t_perm[p] = v;
      t_col[p] = rowid[v];
    }
  }
  if (p > 0) {
    const int64_t c = (int64_t)keys[p - 1];
    prev = c < N ? c : N;
  }
  for (int64_t r = prev + 1; r <= cur; ++r) t_rowptr[r] = (int32_t)p;
}

__global__ void gcn_deg_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm, const float* __restrict__ ew,
                               int64_t N, float* __restrict__ dinv) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const int lo = rowptr[r], hi = rowptr[r + 1];
  float deg;
  if (ew == nullptr) {
    deg = (float)(hi - lo);
  } else {
    deg = 0.f;
    for (int p = lo; p < hi; ++p) {  // sequential, p ascending: reproducible
      const int e = perm[p];
      deg += e >= 0 ? ew[e] : 1.0f;
    }
  }
  dinv[r] = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;
}

__global__ void gcn_val_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
                               const float* __restrict__ ew, const float* __restrict__ dinv, int64_t N, float* __restrict__ val) {
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);  // 16 lanes per row
  if (r >= N) return;
  const int lane = threadIdx.x & 15;
  const int lo = rowptr[r], hi = rowptr[r + 1];
  const float dr = dinv[r];
  for (int p = lo + lane; p < hi; p += 16) {
    float w = 1.0f;
    if (ew) { const int e = perm[p]; w = e >= 0 ? ew[e] : 1.0f; }
    val[p] = dinv[col[p]] * w * dr;
  }
}

__global__ void neg_filter_build_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N, unsigned bits,
                                        uint32_t* __restrict__ filter) {
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // one wave per row
  if (r >= N) return;
  const int lane = threadIdx.x & 63;
  const int lo = rowptr[r], hi = rowptr[r + 1];
  for (int p = lo + lane; p < hi; p += 64) {
    const NegProbe pr = neg_filter_probe(col[p], (int32_t)r, bits);
    atomicOr(&filter[pr.word], pr.mask);  // integer OR: order-independent
  }
}

template <bool FILTER>
__global__ void neg_sample_uniform_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N, int64_t num,
                                          uint64_t seed, const uint32_t* __restrict__ filter, unsigned filter_bits,
                                          int32_t* __restrict__ neg_src, int32_t* __restrict__ neg_dst) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num) return;
  int32_t s, d;
  ceeg_neg_draw<FILTER>(rowptr, col, N, seed, k, filter, filter_bits, s, d);
  neg_src[k] = s;
  neg_dst[k] = d;
}

unsigned key_bits(int64_t N) {  // bits needed for the value N itself (so all-ones > any valid node id)
  unsigned b = 1;
  while (((int64_t)1 << b) <= N) ++b;
  return b;
}

size_t sort_temp_bytes(size_t M, unsigned end_bit) {
  size_t bytes = 0;
  uint64_t* k = nullptr;
  int32_t* v = nullptr;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, M, 0u, end_bit, (hipStream_t)0, false);
  if (e != hipSuccess) return 0;
  return bytes;
}

struct SortWs {
  uint64_t *keys_in, *keys_out;
  int32_t *vals_in, *vals_out, *loop_eid;
  void* temp;
  size_t temp_bytes;
};
size_t sort_layout(int64_t M, int64_t N, unsigned end_bit, void* ws, size_t ws_bytes, SortWs* out) {
  const size_t tb = sort_temp_bytes((size_t)M, end_bit);
  if (tb == 0 && M > 0) return 0;
  size_t need = 2 * ceeg_ws_part(M, 8) + 2 * ceeg_ws_part(M, 4) + ceeg_ws_part(N, 4) + ceeg_align_up(tb, 256);
  if (out) {
    CeegWs a(ws, ws_bytes);
    out->keys_in = a.take<uint64_t>(M);
    out->keys_out = a.take<uint64_t>(M);
    out->vals_in = a.take<int32_t>(M);
    out->vals_out = a.take<int32_t>(M);
    out->loop_eid = a.take<int32_t>(N);
    out->temp = a.take<char>(tb);
    out->temp_bytes = tb;
    if (!a.ok) return 0;
  }
  return need;
}

}  // namespace

extern "C" size_t ceeg_csr_build_ws_bytes(int64_t E, int64_t N, int self_loops) {
  if (E < 0 || N <= 0) return 0;
  const int64_t M = E + (self_loops ? N : 0);
  return sort_layout(M, N, 2 * key_bits(N), nullptr, 0, nullptr);
}

extern "C" int ceeg_csr_build(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, int self_loops,
                              int32_t* rowptr, int32_t* col, int32_t* perm, int64_t* nnz_out,
                              void* ws, size_t ws_bytes, void* stream) {
  if (E < 0 || N <= 0 || !rowptr || (E > 0 && (!src || !dst))) return CEEG_EINVAL;
  const int64_t M = E + (self_loops ? N : 0);
  if (M > 0 && (!col || !perm)) return CEEG_EINVAL;
  if (N >= ((int64_t)1 << 31) || M >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nb = key_bits(N), end_bit = 2 * nb;
  SortWs w;
  const size_t need = sort_layout(M, N, end_bit, nullptr, 0, nullptr);
  if (need == 0 || ws_bytes < need || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  if (sort_layout(M, N, end_bit, ws, ws_bytes, &w) == 0) return CEEG_EWORKSPACE;
  if (self_loops) {
    if (hipMemsetAsync(w.loop_eid, 0xff, (size_t)N * 4, st) != hipSuccess) return CEEG_ELAUNCH;
  }
  if (E > 0) {
    hipLaunchKernelGGL(csr_edge_keys_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, src, dst, E, N, nb, self_loops,
                       w.keys_in, w.vals_in, w.loop_eid);
    CEEG_CHECK_LAUNCH();
  }
  if (self_loops) {
    hipLaunchKernelGGL(csr_loop_keys_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, E, N, nb, w.keys_in, w.vals_in, w.loop_eid);
    CEEG_CHECK_LAUNCH();
  }
  size_t tb = w.temp_bytes;
  if (M > 0) {
    if (rocprim::radix_sort_pairs(w.temp, tb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)M, 0u, end_bit, st, false) != hipSuccess)
      return CEEG_ELAUNCH;
  }
  hipLaunchKernelGGL(csr_finish_kernel, dim3((unsigned)((M + 1 + 255) / 256)), dim3(256), 0, st, w.keys_out, w.vals_out, M, N, nb, rowptr, col,
                     perm, nnz_out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

namespace {
struct Sort32Ws {
  uint32_t *keys_in, *keys_out;
  int32_t *vals_in, *vals_out, *rowid;
  void* temp;
  size_t temp_bytes;
};
size_t sort32_temp_bytes(size_t M, unsigned end_bit) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  int32_t* v = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, M, 0u, end_bit, (hipStream_t)0, false) != hipSuccess) return 0;
  return bytes;
}
size_t sort32_layout(int64_t M, unsigned end_bit, void* ws, size_t ws_bytes, Sort32Ws* out) {
  const size_t tb = sort32_temp_bytes((size_t)M, end_bit);
  if (tb == 0 && M > 0) return 0;
  const size_t need = 5 * ceeg_ws_part(M, 4) + ceeg_align_up(tb, 256);
  if (out) {
    CeegWs a(ws, ws_bytes);
    out->keys_in = a.take<uint32_t>(M);
    out->keys_out = a.take<uint32_t>(M);
    out->vals_in = a.take<int32_t>(M);
    out->vals_out = a.take<int32_t>(M);
    out->rowid = a.take<int32_t>(M);
    out->temp = a.take<char>(tb);
    out->temp_bytes = tb;
    if (!a.ok) return 0;
  }
  return need;
}
}  // namespace

// ---------------------------------------------------------------------------------------------- CSR of a collated batch
// A collated (block-diagonal) edge list keeps every edge inside its graph's block of node ids, so the source fits in lb bits when it
// is stored RELATIVE to its graph's first node: key = dst << lb | (src - node_ptr[batch[dst]]) has bits(N) + lb bits (19 + 13 for 64
// graphs of 6250 nodes) - a 32-bit key-value sort of 8-byte pairs instead of a 64-bit one of 12-byte pairs, and one digit pass fewer.
// The order is the (dst, src) order of ceeg_csr_build: same rowptr / col / perm, bit for bit.  An edge whose endpoints are not in one
// block cannot be encoded: it is dropped and *outside is raised - the caller then builds with ceeg_csr_build.
namespace {
__global__ void csr_edge_keys_blocks_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E, int64_t N, unsigned lb,
                                            const int64_t* __restrict__ batch, const int64_t* __restrict__ node_ptr, int64_t G,
                                            uint32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t* __restrict__ outside) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int64_t s = src[e], d = dst[e];
  uint32_t key = 0xffffffffu;
  if (s >= 0 && s < N && d >= 0 && d < N) {
    const int64_t g = batch[d];
    bool ok = g >= 0 && g < G;
    int64_t ls = 0;
    if (ok) {
      ls = s - node_ptr[g];
      ok = ls >= 0 && ls < ((int64_t)1 << lb) && s < node_ptr[g + 1];
    }
    if (ok) key = ((uint32_t)d << lb) | (uint32_t)ls;
    else atomicOr(outside, 1);   // integer OR: order-independent
  }
  keys[e] = key;
  vals[e] = (int32_t)e;
}

__global__ void csr_finish_blocks_kernel(const uint32_t* __restrict__ keys, const int32_t* __restrict__ vals, int64_t M, int64_t N, unsigned lb,
                                         const int64_t* __restrict__ batch, const int64_t* __restrict__ node_ptr,
                                         int32_t* __restrict__ rowptr, int32_t* __restrict__ col, int32_t* __restrict__ perm,
                                         int64_t* __restrict__ nnz_out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > M) return;
  int64_t d_cur = N, d_prev = -1;
  if (p < M) {
    const uint32_t k = keys[p];
    const int64_t d = (int64_t)(k >> lb);
    d_cur = d < N ? d : N;
    if (d < N) {
      col[p] = (int32_t)((int64_t)(k & ((1u << lb) - 1u)) + node_ptr[batch[d]]);
      perm[p] = vals[p];
    }
  }
  if (p > 0) {
    const int64_t d = (int64_t)(keys[p - 1] >> lb);
    d_prev = d < N ? d : N;
  }
  for (int64_t r = d_prev + 1; r <= d_cur; ++r) {
    rowptr[r] = (int32_t)p;
    if (r == N && nnz_out) *nnz_out = p;
  }
}

unsigned blocks_lb(int64_t max_block_nodes) {  // bits for a local id < max_block_nodes
  unsigned b = 1;
  while (((int64_t)1 << b) < max_block_nodes) ++b;
  return b;
}
}  // namespace

extern "C" int ceeg_csr_build_blocks_supported(int64_t N, int64_t max_block_nodes) {
  return N > 0 && N < ((int64_t)1 << 31) && max_block_nodes > 0 && key_bits(N) + blocks_lb(max_block_nodes) <= 32 ? 1 : 0;
}

extern "C" size_t ceeg_csr_build_blocks_ws_bytes(int64_t E, int64_t N, int64_t max_block_nodes) {
  if (E < 0 || N <= 0 || max_block_nodes <= 0 || key_bits(N) + blocks_lb(max_block_nodes) > 32) return 0;   // 0: the key does not fit
  return sort32_layout(E, key_bits(N) + blocks_lb(max_block_nodes), nullptr, 0, nullptr);
}

extern "C" int ceeg_csr_build_blocks(const int64_t* src, const int64_t* dst, int64_t E, int64_t N, const int64_t* batch,
                                     const int64_t* node_ptr, int64_t G, int64_t max_block_nodes, int32_t* rowptr, int32_t* col,
                                     int32_t* perm, int64_t* nnz_out, int32_t* outside, void* ws, size_t ws_bytes, void* stream) {
  if (E < 0 || N <= 0 || G <= 0 || max_block_nodes <= 0 || !rowptr || !batch || !node_ptr || !outside || (E > 0 && (!src || !dst || !col || !perm)))
    return CEEG_EINVAL;
  if (N >= ((int64_t)1 << 31) || E >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  const unsigned nb = key_bits(N), lb = blocks_lb(max_block_nodes), end_bit = nb + lb;
  if (end_bit > 32) return CEEG_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Sort32Ws w;
  const size_t need = sort32_layout(E, end_bit, nullptr, 0, nullptr);
  if ((need == 0 && E > 0) || ws_bytes < need || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  if (sort32_layout(E, end_bit, ws, ws_bytes, &w) == 0) return CEEG_EWORKSPACE;
  if (hipMemsetAsync(outside, 0, sizeof(int32_t), st) != hipSuccess) return CEEG_ELAUNCH;
  if (E > 0) {
    hipLaunchKernelGGL(csr_edge_keys_blocks_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, src, dst, E, N, lb, batch, node_ptr, G,
                       w.keys_in, w.vals_in, outside);
    CEEG_CHECK_LAUNCH();
    size_t tb = w.temp_bytes;
    if (rocprim::radix_sort_pairs(w.temp, tb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)E, 0u, end_bit, st, false) != hipSuccess)
      return CEEG_ELAUNCH;
  }
  hipLaunchKernelGGL(csr_finish_blocks_kernel, dim3((unsigned)((E + 1 + 255) / 256)), dim3(256), 0, st, w.keys_out, w.vals_out, E, N, lb, batch,
                     node_ptr, rowptr, col, perm, nnz_out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" size_t ceeg_csr_transpose_ws_bytes(int64_t nnz, int64_t N) {
  if (nnz < 0 || N <= 0) return 0;
  return sort32_layout(nnz, key_bits(N), nullptr, 0, nullptr);
}

extern "C" int ceeg_csr_transpose(const int32_t* rowptr, const int32_t* col, int64_t nnz, int64_t N,
                                  int32_t* t_rowptr, int32_t* t_col, int32_t* t_perm,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (nnz < 0 || N <= 0 || !rowptr || !t_rowptr || (nnz > 0 && (!col || !t_col || !t_perm))) return CEEG_EINVAL;
  if (N >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nb = key_bits(N);  // the padding key 0xffffffff has all nb low bits set: a column >= N, sorted last
  Sort32Ws w;
  const size_t need = sort32_layout(nnz, nb, nullptr, 0, nullptr);
  if ((need == 0 && nnz > 0) || ws_bytes < need || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  if (sort32_layout(nnz, nb, ws, ws_bytes, &w) == 0 && nnz > 0) return CEEG_EWORKSPACE;
  // `nnz` may be a CAPACITY (>= rowptr[N], the true count, which then never has to cross to the host): positions the rows do
  // not cover keep the all-ones key, sort behind every entry and are ignored by the finish kernel
  if (nnz > 0 && hipMemsetAsync(w.keys_in, 0xff, (size_t)nnz * 4, st) != hipSuccess) return CEEG_ELAUNCH;
  hipLaunchKernelGGL(csr_transpose_keys32_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, st, rowptr, col, N, w.keys_in, w.vals_in, w.rowid);
  CEEG_CHECK_LAUNCH();
  size_t tb = w.temp_bytes;
  if (nnz > 0) {
    if (rocprim::radix_sort_pairs(w.temp, tb, w.keys_in, w.keys_out, w.vals_in, w.vals_out, (size_t)nnz, 0u, nb, st, false) != hipSuccess)
      return CEEG_ELAUNCH;
  }
  hipLaunchKernelGGL(csr_transpose_finish32_kernel, dim3((unsigned)((nnz + 1 + 255) / 256)), dim3(256), 0, st, w.keys_out, w.vals_out, w.rowid,
                     nnz, N, t_rowptr, t_col, t_perm);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

// ---------------------------------------------------------------------------------------------- row statistics + piece lists
// What the host needs to know about a freshly built CSR before it can launch on it - the entry count, the longest row and the
// piece list of the rows longer than `chunk` entries (ceeg_spmm_csr_items / ceeg_decoder_codes_* layout) - produced ON THE DEVICE
// into one small record, so that a graph built inside a training loop (a new minibatch every step, scripts/run.py:310-326) costs
// ONE host read instead of one per quantity.
namespace {
__global__ __launch_bounds__(256) void csr_piece_count_kernel(const int32_t* __restrict__ rowptr, int64_t N, int chunk,
                                                              int32_t* __restrict__ cnt, unsigned long long* __restrict__ rec) {
  __shared__ int red[4];
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int len = 0;
  if (r < N) {
    const int lo = rowptr[r], hi = rowptr[r + 1];
    len = hi - lo;
    cnt[r] = len > chunk ? (hi - 1) / chunk - lo / chunk + 1 : 0;
  } else if (r == N) {
    cnt[N] = 0;
    rec[0] = (unsigned long long)rowptr[N];
  }
  int m = len;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    // integer max: order-independent; the plain read only skips atomics that could not raise the value (it never decreases)
    if (m > 0 && (unsigned long long)m > *reinterpret_cast<volatile unsigned long long*>(&rec[1])) atomicMax(&rec[1], (unsigned long long)m);
  }
}
__global__ __launch_bounds__(256) void csr_piece_fill_kernel(const int32_t* __restrict__ rowptr, int64_t N, int chunk,
                                                             const int32_t* __restrict__ off, int4* __restrict__ items, int64_t cap,
                                                             unsigned long long* __restrict__ rec) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r == N) rec[2] = (unsigned long long)off[N];
  if (r >= N) return;
  const int lo = rowptr[r], hi = rowptr[r + 1];
  if (hi - lo <= chunk) return;
  const int c_lo = lo / chunk, c_hi = (hi - 1) / chunk;
  int64_t o = off[r];
  for (int c = c_lo; c <= c_hi; ++c, ++o) {
    if (o >= cap) return;  // the host sees rec[2] > cap and raises
    const int plo = max(lo, c * chunk), phi = (int)min((int64_t)hi, ((int64_t)c + 1) * chunk);
    const int slot = (c == c_lo && lo != c * chunk) ? 1 : 0;
    items[o] = make_int4(plo, phi, c * 2 + slot, (int)r);
  }
}
size_t piece_scan_temp(int64_t n) {
  size_t bytes = 0;
  int32_t* p = nullptr;
  if (rocprim::exclusive_scan(nullptr, bytes, p, p, 0, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t)0, false) != hipSuccess) return 0;
  return bytes < 256 ? 256 : bytes;
}
}  // namespace

extern "C" size_t ceeg_csr_long_items_ws_bytes(int64_t N) {
  if (N <= 0) return 0;
  return 2 * ceeg_ws_part((size_t)N + 1, 4) + ceeg_align_up(piece_scan_temp(N + 1), 256);
}

extern "C" int ceeg_csr_long_items(const int32_t* rowptr, int64_t N, int chunk, int32_t* items, int64_t items_cap, int64_t* record,
                                   void* ws, size_t ws_bytes, void* stream) {
  if (!rowptr || !record || N <= 0 || chunk <= 0 || items_cap < 0 || (items_cap > 0 && (!items || !ceeg_aligned(items, 16)))) return CEEG_EINVAL;
  if (N >= ((int64_t)1 << 31) - 1) return CEEG_EUNSUPPORTED;
  const size_t need = ceeg_csr_long_items_ws_bytes(N);
  if (need == 0 || ws_bytes < need || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CeegWs a(ws, ws_bytes);
  int32_t* cnt = a.take<int32_t>((size_t)N + 1);
  int32_t* off = a.take<int32_t>((size_t)N + 1);
  size_t tb = piece_scan_temp(N + 1);
  void* temp = a.take<char>(tb);
  if (!a.ok) return CEEG_EWORKSPACE;
  if (hipMemsetAsync(record, 0, 3 * sizeof(int64_t), st) != hipSuccess) return CEEG_ELAUNCH;
  unsigned long long* rec = reinterpret_cast<unsigned long long*>(record);
  const dim3 grid((unsigned)((N + 1 + 255) / 256));
  hipLaunchKernelGGL(csr_piece_count_kernel, grid, dim3(256), 0, st, rowptr, N, chunk, cnt, rec);
  CEEG_CHECK_LAUNCH();
  if (rocprim::exclusive_scan(temp, tb, cnt, off, 0, (size_t)N + 1, rocprim::plus<int32_t>(), st, false) != hipSuccess) return CEEG_ELAUNCH;
  hipLaunchKernelGGL(csr_piece_fill_kernel, grid, dim3(256), 0, st, rowptr, N, chunk, off, reinterpret_cast<int4*>(items), items_cap, rec);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

// ---------------------------------------------------------------------------------------------- A + I from the raw-edge pair
// A minibatch loop builds BOTH structures of a graph every step - the raw edges for the decoder, A + I for the encoder (PyG
// add_remaining_self_loops: input self-loops dropped, one loop per node that keeps the last input loop's edge id) - and the
// second one does not need a second sort: row r of A + I is row r of the raw CSR with its run of `col == r` entries (contiguous:
// rows are sorted by column, equal pairs by edge id) replaced by ONE loop entry at the same place.  The same holds for the
// transposed rows, and both orientations drop the same number of entries per row, so one prefix sum serves both.
namespace {
// thread per row: lt[r] = entries with col < r, loops[r] = entries with col == r
__global__ __launch_bounds__(256) void csr_loops_count_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N,
                                                              int32_t* __restrict__ lt, int32_t* __restrict__ loops) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r > N) return;
  if (r == N) {
    if (loops) loops[N] = 0;
    return;
  }
  const int lo = rowptr[r], hi = rowptr[r + 1];
  int a = lo, b = hi;
  while (a < b) {  // first position with col >= r
    const int m = (a + b) >> 1;
    if (col[m] < (int32_t)r) a = m + 1; else b = m;
  }
  int e = a;
  while (e < hi && col[e] == (int32_t)r) ++e;
  lt[r] = a - lo;
  if (loops) loops[r] = e - a;
}

// 16 lanes per row.  TRANSPOSED = false: (col, perm) are the raw CSR's columns and edge ids, the loop entry gets the last dropped
// loop's edge id or -1.  TRANSPOSED = true: (col, perm) are the raw transpose's (t_col, t_perm); an entry's t_perm becomes its
// position in the NEW forward CSR (info[c] = shift of forward row c's start and of its entries behind the loop, written by the
// forward pass; g_rowptr = the new forward row pointers, f_lt = loop position inside a forward row), the loop entry of row r
// points at the forward loop entry of row r.
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void csr_loops_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ perm, const int32_t* __restrict__ lt,
                                                             const int32_t* __restrict__ loops_scan /* exclusive, N + 1 */, int64_t N,
                                                             int2* __restrict__ info /* forward: written; transposed: read */,
                                                             const int32_t* __restrict__ f_lt, const int32_t* __restrict__ g_rowptr,
                                                             int32_t* __restrict__ o_rowptr, int32_t* __restrict__ o_col,
                                                             int32_t* __restrict__ o_perm) {
  const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool valid = r < N;
  const int l = threadIdx.x & 15;
  int lo = 0, hi = 0, s0 = 0, nl = 0, a = 0;
  if (valid) {
    lo = rowptr[r];
    hi = rowptr[r + 1];
    s0 = loops_scan[r];
    nl = loops_scan[r + 1] - s0;
    a = lt[r];
  }
  const int base = lo - s0 + (int)r;
  if (valid && l == 0) {
    o_rowptr[r] = base;
    if (r == N - 1) o_rowptr[N] = hi - loops_scan[N] + (int)N;
    o_col[base + a] = (int32_t)r;
    if constexpr (TRANSPOSED) {
      o_perm[base + a] = g_rowptr[r] + f_lt[r];
    } else {
      o_perm[base + a] = nl > 0 ? perm[lo + a + nl - 1] : -1;
      info[r] = make_int2(base - lo, 1 - nl);  // what the transposed side needs of this row, in one 8-byte read per entry
    }
  }
  auto entry = [&](int64_t rr, int p, int lo_, int a_, int nl_, int base_) {
    const int k = p - lo_;
    if (k >= a_ && k < a_ + nl_) return;  // an input self-loop
    const int q = base_ + (k < a_ ? k : k - nl_ + 1);
    const int32_t c = col[p];
    o_col[q] = c;
    if constexpr (TRANSPOSED) {
      const int fp = perm[p];                       // position in the raw forward CSR, in forward row c, whose column is rr
      const int2 fi = info[c];                      // (new - old start of forward row c, 1 - its dropped loops)
      o_perm[q] = fp + fi.x + ((int32_t)rr > c ? fi.y : 0);
    } else {
      o_perm[q] = perm[p];
    }
  };
  constexpr int LONG = 256;   // rows above this go to the whole wave below (hub rows of the transposed event graph run to thousands)
  const bool is_long = hi - lo > LONG;
  if (!is_long)
    for (int p = lo + l; p < hi; p += 16) entry(r, p, lo, a, nl, base);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (!__shfl((int)is_long, g * 16, 64)) continue;   // wave-uniform
    const int glo = __shfl(lo, g * 16, 64), ghi = __shfl(hi, g * 16, 64), ga = __shfl(a, g * 16, 64), gnl = __shfl(nl, g * 16, 64);
    const int gbase = __shfl(base, g * 16, 64);
    const int64_t gr = r - (lane >> 4) + g;
    for (int p = glo + lane; p < ghi; p += 64) entry(gr, p, glo, ga, gnl, gbase);
  }
}
}  // namespace

extern "C" size_t ceeg_csr_add_self_loops_ws_bytes(int64_t N) {
  if (N <= 0) return 0;
  return 4 * ceeg_ws_part((size_t)N + 1, 4) + ceeg_ws_part((size_t)N + 1, 8) + ceeg_align_up(piece_scan_temp(N + 1), 256);
}

extern "C" int ceeg_csr_add_self_loops(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int32_t* t_rowptr,
                                       const int32_t* t_col, const int32_t* t_perm, int64_t N, int32_t* g_rowptr, int32_t* g_col,
                                       int32_t* g_perm, int32_t* gt_rowptr, int32_t* gt_col, int32_t* gt_perm, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (N <= 0 || !rowptr || !col || !perm || !t_rowptr || !t_col || !t_perm || !g_rowptr || !g_col || !g_perm || !gt_rowptr || !gt_col ||
      !gt_perm)
    return CEEG_EINVAL;
  if (N >= ((int64_t)1 << 31) - 1) return CEEG_EUNSUPPORTED;
  const size_t need = ceeg_csr_add_self_loops_ws_bytes(N);
  if (need == 0 || ws_bytes < need || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CeegWs a(ws, ws_bytes);
  int32_t* lt = a.take<int32_t>((size_t)N + 1);
  int32_t* t_lt = a.take<int32_t>((size_t)N + 1);
  int32_t* loops = a.take<int32_t>((size_t)N + 1);
  int32_t* scan = a.take<int32_t>((size_t)N + 1);
  int2* info = a.take<int2>((size_t)N + 1);
  size_t tb = piece_scan_temp(N + 1);
  void* temp = a.take<char>(tb);
  if (!a.ok) return CEEG_EWORKSPACE;
  const dim3 rows((unsigned)((N + 1 + 255) / 256)), groups((unsigned)((N + 15) / 16));
  hipLaunchKernelGGL(csr_loops_count_kernel, rows, dim3(256), 0, st, rowptr, col, N, lt, loops);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(csr_loops_count_kernel, rows, dim3(256), 0, st, t_rowptr, t_col, N, t_lt, static_cast<int32_t*>(nullptr));
  CEEG_CHECK_LAUNCH();
  if (rocprim::exclusive_scan(temp, tb, loops, scan, 0, (size_t)N + 1, rocprim::plus<int32_t>(), st, false) != hipSuccess) return CEEG_ELAUNCH;
  hipLaunchKernelGGL(csr_loops_fill_kernel<false>, groups, dim3(256), 0, st, rowptr, col, perm, lt, scan, N, info,
                     static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), g_rowptr, g_col, g_perm);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(csr_loops_fill_kernel<true>, groups, dim3(256), 0, st, t_rowptr, t_col, t_perm, t_lt, scan, N, info, lt, g_rowptr, gt_rowptr,
                     gt_col, gt_perm);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

namespace {
__global__ void gcn_tval_kernel(const int32_t* __restrict__ t_rowptr, const int32_t* __restrict__ t_perm, const float* __restrict__ val,
                                int64_t N, int64_t cap, float* __restrict__ tval) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < cap && q < t_rowptr[N]) tval[q] = val[t_perm[q]];
}
}  // namespace

extern "C" int ceeg_gcn_norm_t(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const float* ew, int64_t nnz, int64_t N,
                               float* deg_inv_sqrt, float* val, const int32_t* t_rowptr, const int32_t* t_perm, float* tval, void* stream) {
  if (!t_rowptr || (nnz > 0 && (!t_perm || !tval))) return CEEG_EINVAL;
  const int rc = ceeg_gcn_norm(rowptr, col, perm, ew, nnz, N, deg_inv_sqrt, val, stream);
  if (rc != CEEG_OK || nnz == 0) return rc;
  hipLaunchKernelGGL(gcn_tval_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), t_rowptr, t_perm, val, N,
                     nnz, tval);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_gcn_norm(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const float* ew,
                             int64_t nnz, int64_t N, float* deg_inv_sqrt, float* val, void* stream) {
  if (N <= 0 || nnz < 0 || !rowptr || !deg_inv_sqrt || (nnz > 0 && (!col || !val)) || (ew && !perm)) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(gcn_deg_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, rowptr, perm, ew, N, deg_inv_sqrt);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(gcn_val_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, st, rowptr, col, perm, ew, deg_inv_sqrt, N, val);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_neg_sample_uniform(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t num, uint64_t seed,
                                       int32_t* neg_src, int32_t* neg_dst, void* stream) {
  return ceeg_neg_sample_uniform_filtered(rowptr, col, N, num, seed, nullptr, 0, neg_src, neg_dst, stream);
}

extern "C" int ceeg_neg_filter_bits(int64_t nnz) {  // log2 of the bitmap size: >= 16 bits per edge (<= 1/16 of the bits set), 2^13 .. 2^32
  int b = 13;
  while (b < 32 && ((int64_t)1 << b) < 16 * nnz) ++b;
  return b;
}

extern "C" int ceeg_neg_filter_build(const int32_t* rowptr, const int32_t* col, int64_t N, int bits, uint32_t* filter, void* stream) {
  if (N <= 0 || !rowptr || !filter || bits < 13 || bits > 32 || !ceeg_aligned(filter, 4)) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(filter, 0, ((size_t)1 << bits) / 8, st) != hipSuccess) return CEEG_ELAUNCH;
  hipLaunchKernelGGL(neg_filter_build_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, rowptr, col, N, (unsigned)bits, filter);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_neg_sample_uniform_filtered(const int32_t* rowptr, const int32_t* col, int64_t N, int64_t num, uint64_t seed,
                                                const uint32_t* filter, int filter_bits, int32_t* neg_src, int32_t* neg_dst,
                                                void* stream) {
  if (N <= 1 || num < 0 || !rowptr || !col || (num > 0 && (!neg_src || !neg_dst))) return CEEG_EINVAL;
  if (filter && (filter_bits < 13 || filter_bits > 32)) return CEEG_EINVAL;
  if (num == 0) return CEEG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)((num + 255) / 256));
  if (filter)
    hipLaunchKernelGGL(neg_sample_uniform_kernel<true>, grid, dim3(256), 0, st, rowptr, col, N, num, seed, filter, (unsigned)filter_bits,
                       neg_src, neg_dst);
  else
    hipLaunchKernelGGL(neg_sample_uniform_kernel<false>, grid, dim3(256), 0, st, rowptr, col, N, num, seed, filter, 0u, neg_src, neg_dst);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

// ---------------------------------------------------------------------------------------------- collation (N2)
// PyG-style batching of disjoint graphs (torch_geometric DataLoader / Batch.from_data_list as the reference calls it at
// scripts/run.py:310-326 with follow_batch=[...], and its inverse `separate`, scripts/dataset.py:66-92), on the device:
//   ceeg_segment_ids   : out[i] = g for ptr[g] <= i < ptr[g+1]  - the `batch` vector and every followed key's `<key>_batch`
//   ceeg_collate_edges : dst_edges[:, e] = src_edges[:, e] + sign * node_ptr[g(e)], g(e) from edge_ptr - block-diagonal
//                        offsetting (sign = +1) and its inverse (sign = -1: what `separate` subtracts again)
// Streaming integer work; the graph of an element is a binary search in the (L2-resident) ptr array.
namespace {
__device__ __forceinline__ int seg_of(const int64_t* __restrict__ ptr, int G, int64_t i) {
  int lo = 0, hi = G;  // last g with ptr[g] <= i (empty graphs have ptr[g] == ptr[g+1] and are skipped)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}
__global__ void segment_ids_kernel(const int64_t* __restrict__ ptr, int G, int64_t total, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = seg_of(ptr, G, i);
}
__global__ void collate_edges_kernel(const int64_t* __restrict__ in, int64_t ld_in, const int64_t* __restrict__ edge_ptr,
                                     const int64_t* __restrict__ node_ptr, int G, int64_t E, int sign, int64_t* __restrict__ out,
                                     int64_t ld_out, int64_t* __restrict__ edge_batch, int32_t* __restrict__ outside) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int g = seg_of(edge_ptr, G, e);
  const int64_t off = sign * node_ptr[g];
  const int64_t s = in[e], d = in[ld_in + e];
  out[e] = s + off;
  out[ld_out + e] = d + off;
  if (edge_batch) edge_batch[e] = g;
  if (outside) {  // an endpoint that is not a local id of graph g: the collated edge leaves its graph's block (integer OR: order-independent)
    const int64_t n = node_ptr[g + 1] - node_ptr[g];
    if (s < 0 || s >= n || d < 0 || d >= n) atomicOr(outside, 1);
  }
}
}  // namespace

extern "C" int ceeg_segment_ids(const int64_t* ptr, int64_t G, int64_t total, int64_t* out, void* stream) {
  if (!ptr || G <= 0 || total < 0 || G >= ((int64_t)1 << 31) || (total > 0 && !out)) return CEEG_EINVAL;
  if (total == 0) return CEEG_OK;
  hipLaunchKernelGGL(segment_ids_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), ptr, (int)G, total, out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_collate_edges(const int64_t* edges_in, int64_t ld_in, const int64_t* edge_ptr, const int64_t* node_ptr, int64_t G,
                                  int64_t E, int sign, int64_t* edges_out, int64_t ld_out, int64_t* edge_batch, void* stream) {
  if (!edge_ptr || !node_ptr || G <= 0 || E < 0 || G >= ((int64_t)1 << 31) || (sign != 1 && sign != -1)) return CEEG_EINVAL;
  if (E == 0) return CEEG_OK;
  if (!edges_in || !edges_out || ld_in < E || ld_out < E) return CEEG_EINVAL;
  hipLaunchKernelGGL(collate_edges_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), edges_in, ld_in,
                     edge_ptr, node_ptr, (int)G, E, sign, edges_out, ld_out, edge_batch, static_cast<int32_t*>(nullptr));
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

// ceeg_collate_edges (sign = +1) that also reports whether every edge stays inside its graph: *outside (int32 on the device) is set
// to 0 and then to 1 if some endpoint is not a local id of its graph.
extern "C" int ceeg_collate_edges_checked(const int64_t* edges_in, int64_t ld_in, const int64_t* edge_ptr, const int64_t* node_ptr, int64_t G,
                                          int64_t E, int64_t* edges_out, int64_t ld_out, int64_t* edge_batch, int32_t* outside, void* stream) {
  if (!edge_ptr || !node_ptr || !outside || G <= 0 || E < 0 || G >= ((int64_t)1 << 31)) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(outside, 0, sizeof(int32_t), st) != hipSuccess) return CEEG_ELAUNCH;
  if (E == 0) return CEEG_OK;
  if (!edges_in || !edges_out || ld_in < E || ld_out < E) return CEEG_EINVAL;
  hipLaunchKernelGGL(collate_edges_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, edges_in, ld_in, edge_ptr, node_ptr, (int)G, E, 1,
                     edges_out, ld_out, edge_batch, outside);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}
