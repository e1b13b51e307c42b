// Ancestor-subgraph extraction of the reference's dataset builder (scripts/dataset.py:22-46,
// 201-225) for ALL target nodes at once, as a frontier expansion on the device.  Integer work, no atomics.
//   ceeg_top_predecessors   per node: its in-edges with entities >= min_entities, heaviest first, ties in the order the edges
//                           first appeared (edge id ascending = what sorted(..., reverse=True) over networkx's predecessor
//                           iteration gives), at most min(n, in-degree)                                   dataset.py:22-33
//   ceeg_ancestors          per target: the union of the top predecessors over `gens` generations          dataset.py:35-46
//                           (generation-by-generation gathers from the table above; sorted, duplicates removed)
//   ceeg_ancestor_subgraph_count / _fill  per target: the edges of the graph induced on its ancestors      dataset.py:201-225
//                           with LOCAL node ids (position in the sorted ancestor list), in (target, source) order
#include "ceeg_internal.h"

namespace {

constexpr int ANC_MAX_N = 8;      // predecessors kept per node
constexpr int ANC_MAX_CAP = 128;  // ancestors per target (3 + 9 + 27 = 39 for the reference's constants)

__global__ void top_pred_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ perm,
                                const int64_t* __restrict__ entities, int64_t N, int n, int64_t min_entities, int32_t* __restrict__ top) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= N) return;
  int32_t bs[ANC_MAX_N], be[ANC_MAX_N];
  int64_t bw[ANC_MAX_N];
  int m = 0;
  const int lo = rowptr[v], hi = rowptr[v + 1];
  const int keep = n < hi - lo ? n : hi - lo;  // [:min(n, in_degree)]
  for (int p = lo; p < hi; ++p) {
    const int32_t e = perm[p];
    const int64_t w = entities[e];
    if (w < min_entities) continue;
    // insertion into the list ordered by (weight descending, edge id ascending)
    int pos = m;
    while (pos > 0 && (bw[pos - 1] < w || (bw[pos - 1] == w && be[pos - 1] > e))) --pos;
    if (pos >= keep) continue;
    const int last = m < keep ? m : keep - 1;
    for (int i = last; i > pos; --i) { bs[i] = bs[i - 1]; be[i] = be[i - 1]; bw[i] = bw[i - 1]; }
    bs[pos] = col[p]; be[pos] = e; bw[pos] = w;
    if (m < keep) ++m;
  }
  for (int i = 0; i < n; ++i) top[v * n + i] = i < m ? bs[i] : -1;
}

__global__ __launch_bounds__(64) void ancestors_kernel(const int32_t* __restrict__ top, int64_t N, int n, const int32_t* __restrict__ targets,
                                                       int64_t T, int gens, int cap, int32_t* __restrict__ anc, int32_t* __restrict__ cnt) {
  __shared__ int32_t s_list[64 * ANC_MAX_CAP];
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= T) return;
  int32_t* lst = s_list + threadIdx.x * ANC_MAX_CAP;
  const int32_t start = targets ? targets[t] : (int32_t)t;
  int m = 0;
  // generation g occupies lst[g_lo, g_hi); generation 0 (the target itself) is kept aside
  int g_lo = 0, g_hi = 0;
  for (int g = 0; g < gens; ++g) {
    const int new_lo = m;
    if (g == 0) {
      for (int i = 0; i < n; ++i) { const int32_t p = top[(int64_t)start * n + i]; if (p >= 0 && m < cap) lst[m++] = p; }
    } else {
      for (int k = g_lo; k < g_hi; ++k)
        for (int i = 0; i < n; ++i) { const int32_t p = top[(int64_t)lst[k] * n + i]; if (p >= 0 && m < cap) lst[m++] = p; }
    }
    g_lo = new_lo;
    g_hi = m;
  }
  // sort ascending (insertion sort on <= cap values), drop duplicates
  for (int i = 1; i < m; ++i) {
    const int32_t x = lst[i];
    int j = i - 1;
    while (j >= 0 && lst[j] > x) { lst[j + 1] = lst[j]; --j; }
    lst[j + 1] = x;
  }
  int u = 0;
  for (int i = 0; i < m; ++i)
    if (i == 0 || lst[i] != lst[i - 1]) lst[u++] = lst[i];
  for (int i = 0; i < cap; ++i) anc[t * cap + i] = i < u ? lst[i] : -1;
  cnt[t] = u;
}

__device__ __forceinline__ int anc_find(const int32_t* __restrict__ a, int m, int32_t x) {
  int lo = 0, hi = m;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return (lo < m && a[lo] == x) ? lo : -1;
}

// one thread per (target, ancestor slot): the in-edges of that ancestor whose source is an ancestor too
template <bool FILL>
__global__ void anc_subgraph_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const int32_t* __restrict__ anc,
                                    const int32_t* __restrict__ cnt, int64_t T, int cap, int64_t* __restrict__ slot_count /* [T*cap] */,
                                    const int64_t* __restrict__ slot_off /* [T*cap] exclusive scan */, int32_t* __restrict__ sub_src,
                                    int32_t* __restrict__ sub_dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * cap) return;
  const int64_t t = i / cap;
  const int k = (int)(i % cap);
  const int m = cnt[t];
  int64_t c = 0;
  if (k < m) {
    const int32_t* a = anc + t * cap;
    const int32_t v = a[k];
    int64_t o = FILL ? slot_off[i] : 0;
    for (int p = rowptr[v]; p < rowptr[v + 1]; ++p) {
      const int loc = anc_find(a, m, col[p]);
      if (loc < 0) continue;
      if (FILL) { sub_src[o] = loc; sub_dst[o] = k; ++o; }
      ++c;
    }
  }
  if (!FILL) slot_count[i] = c;
}

}  // namespace

extern "C" int ceeg_top_predecessors(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int64_t* entities, int64_t N,
                                     int n, int64_t min_entities, int32_t* top, void* stream) {
  if (!rowptr || !top || N <= 0 || n <= 0) return CEEG_EINVAL;
  if (n > ANC_MAX_N) return CEEG_EUNSUPPORTED;
  if (!col || !perm || !entities) return CEEG_EINVAL;  // a graph without edges still passes 1-element dummies
  hipLaunchKernelGGL(top_pred_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, static_cast<hipStream_t>(stream), rowptr, col, perm, entities,
                     N, n, min_entities, top);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_ancestors_cap(int n, int gens) {  // sum_{g=1..gens} n^g, or -1 when it exceeds what the kernel holds
  if (n <= 0 || gens < 0) return -1;
  int64_t tot = 0, p = 1;
  for (int g = 0; g < gens; ++g) {
    p *= n;
    tot += p;
    if (tot > ANC_MAX_CAP) return -1;
  }
  return (int)tot;
}

extern "C" int ceeg_ancestors(const int32_t* top, int64_t N, int n, const int32_t* targets, int64_t T, int gens, int cap, int32_t* anc,
                              int32_t* cnt, void* stream) {
  if (!top || !anc || !cnt || N <= 0 || T <= 0 || n <= 0 || gens < 0) return CEEG_EINVAL;
  if (n > ANC_MAX_N || ceeg_ancestors_cap(n, gens) < 0 || cap < ceeg_ancestors_cap(n, gens) || cap > ANC_MAX_CAP || cap < 1) return CEEG_EUNSUPPORTED;
  hipLaunchKernelGGL(ancestors_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), top, N, n, targets, T, gens,
                     cap, anc, cnt);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_ancestor_subgraph_count(const int32_t* rowptr, const int32_t* col, const int32_t* anc, const int32_t* cnt, int64_t T, int cap,
                                            int64_t* slot_count, void* stream) {
  if (!rowptr || !col || !anc || !cnt || !slot_count || T <= 0 || cap <= 0 || cap > ANC_MAX_CAP) return CEEG_EINVAL;
  hipLaunchKernelGGL(anc_subgraph_kernel<false>, dim3((unsigned)((T * cap + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr,
                     col, anc, cnt, T, cap, slot_count, nullptr, nullptr, nullptr);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_ancestor_subgraph_fill(const int32_t* rowptr, const int32_t* col, const int32_t* anc, const int32_t* cnt, int64_t T, int cap,
                                           const int64_t* slot_off, int32_t* sub_src, int32_t* sub_dst, void* stream) {
  if (!rowptr || !col || !anc || !cnt || !slot_off || !sub_src || !sub_dst || T <= 0 || cap <= 0 || cap > ANC_MAX_CAP) return CEEG_EINVAL;
  hipLaunchKernelGGL(anc_subgraph_kernel<true>, dim3((unsigned)((T * cap + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), rowptr,
                     col, anc, cnt, T, cap, nullptr, slot_off, sub_src, sub_dst);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}
