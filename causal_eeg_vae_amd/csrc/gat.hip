// GATConv attention for gfx950 (upstream PyG GATConv as the reference instantiates it at
// scripts/models.py:106): per-destination softmax of leaky_relu(a_src[j] + a_dst[i]) over the CSR row of i
// (self-loops included), its backward, and the scalar segment sum over the transposed CSR.  The weighted
// aggregation itself is ceeg_spmm_csr with the softmax weights as values.  HBM/L2-bound integer/gather work:
// one wave per row, every reduction in a fixed order (lane partials -> xor butterfly), no atomics.
#include <math.h>
#include "ceeg_internal.h"

namespace {

constexpr int GAT_WAVES = 4;
constexpr int GAT_MAX_BLOCKS = 8192;

__device__ __forceinline__ float gat_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float gat_leaky(float v, float slope) { return v > 0.f ? v : slope * v; }

// w[p] = exp(e[p] - max_row) / (sum_row + 1e-16)   (PyG utils.softmax)
__global__ __launch_bounds__(GAT_WAVES * 64) void gat_softmax_fwd_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ a_src,
    const float* __restrict__ a_dst, int64_t N, float slope, float* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * GAT_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * GAT_WAVES;
  for (int64_t r = wave; r < N; r += nwaves) {
    const int p_lo = rowptr[r], p_hi = rowptr[r + 1];
    if (p_lo == p_hi) continue;
    const float ad = a_dst[r];
    float m = -INFINITY;
    float e0 = -INFINITY;  // the first batch stays in registers (rows of <= 64 entries: one load)
    for (int base = p_lo; base < p_hi; base += 64) {
      float e = -INFINITY;
      if (base + lane < p_hi) e = gat_leaky(a_src[col[base + lane]] + ad, slope);
      if (base == p_lo) e0 = e;
      m = fmaxf(m, e);
    }
    m = gat_wave_max(m);
    float s = 0.f;
    for (int base = p_lo; base < p_hi; base += 64) {
      float e = e0;
      if (base != p_lo) e = base + lane < p_hi ? gat_leaky(a_src[col[base + lane]] + ad, slope) : -INFINITY;
      if (base + lane < p_hi) s += expf(e - m);
    }
    s = ceeg_wave_sum(s) + 1e-16f;
    for (int base = p_lo; base < p_hi; base += 64) {
      float e = e0;
      if (base != p_lo) e = base + lane < p_hi ? gat_leaky(a_src[col[base + lane]] + ad, slope) : -INFINITY;
      if (base + lane < p_hi) w[base + lane] = expf(e - m) / s;
    }
  }
}

__device__ __forceinline__ float gat_dot4(const float4& a, const float4& b) {
  return __fmaf_rn(a.w, b.w, __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)));
}

// dw of one 64-entry batch: lane L ends up with <g[row], xp[col of entry L]> (0 past the batch end)
template <int LPR>
__device__ __forceinline__ float gat_batch_dots(const float* __restrict__ xp, int64_t ldx, int my_c, int n, int lane, bool fl_ok,
                                                const float4& gr) {
  constexpr int G = 64 / LPR;
  const int fl = lane % LPR, sub = lane / LPR;
  float mine = 0.f;
  const int ns = (n + G - 1) / G;
  for (int i0 = 0; i0 < ns; i0 += 4) {
    float4 x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int j = (i0 + k) * G + sub;
      const int c = __shfl(my_c, j & 63, 64);
      x[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i0 + k < ns && j < n && fl_ok) x[k] = *reinterpret_cast<const float4*>(xp + (int64_t)c * ldx + 4 * fl);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float d = gat_dot4(x[k], gr);
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      // entry L was computed at step L / G by lane group L % G
      const float t = __shfl(d, (lane % G) * LPR, 64);
      if (lane / G == i0 + k) mine = t;
    }
  }
  return lane < n ? mine : 0.f;
}

template <int LPR>
__global__ __launch_bounds__(GAT_WAVES * 64) void gat_attn_bwd_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
    const float* __restrict__ a_src, const float* __restrict__ a_dst, const float* __restrict__ g, int64_t ldg,
    const float* __restrict__ xp, int64_t ldx, int64_t N, int C, float slope, float* __restrict__ dl, float* __restrict__ d_adst) {
  const int lane = threadIdx.x & 63;
  const int fl = lane % LPR;
  const bool fl_ok = 4 * fl < C;
  const int64_t wave = (int64_t)blockIdx.x * GAT_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * GAT_WAVES;
  for (int64_t r = wave; r < N; r += nwaves) {
    const int p_lo = rowptr[r], p_hi = rowptr[r + 1];
    if (p_lo == p_hi) {
      if (lane == 0) d_adst[r] = 0.f;
      continue;
    }
    float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fl_ok) gr = *reinterpret_cast<const float4*>(g + r * ldg + 4 * fl);
    const float ad = a_dst[r];
    const bool one_batch = p_hi - p_lo <= 64;
    // pass 1: s = sum_k w[k] dw[k]  (lane partials over the batches in order, then the butterfly)
    float s = 0.f, dw0 = 0.f;
    for (int base = p_lo; base < p_hi; base += 64) {
      const int n = p_hi - base < 64 ? p_hi - base : 64;
      const int my_c = lane < n ? col[base + lane] : 0;
      const float dw = gat_batch_dots<LPR>(xp, ldx, my_c, n, lane, fl_ok, gr);
      if (base == p_lo) dw0 = dw;
      if (lane < n) s = __fmaf_rn(w[base + lane], dw, s);
    }
    s = ceeg_wave_sum(s);
    // pass 2: de = w (dw - s); through the leaky relu; row sum -> d a_dst
    float da = 0.f;
    for (int base = p_lo; base < p_hi; base += 64) {
      const int n = p_hi - base < 64 ? p_hi - base : 64;
      const int my_c = lane < n ? col[base + lane] : 0;
      const float dw = (one_batch || base == p_lo) ? dw0 : gat_batch_dots<LPR>(xp, ldx, my_c, n, lane, fl_ok, gr);
      if (lane < n) {
        const float de = w[base + lane] * (dw - s);
        const float raw = a_src[my_c] + ad;
        const float v = raw > 0.f ? de : slope * de;
        dl[base + lane] = v;
        da += v;
      }
    }
    da = ceeg_wave_sum(da);
    if (lane == 0) d_adst[r] = da;
  }
}

__global__ __launch_bounds__(GAT_WAVES * 64) void segment_sum_perm_kernel(const int32_t* __restrict__ t_rowptr,
                                                                          const int32_t* __restrict__ t_perm,
                                                                          const float* __restrict__ v, int64_t N,
                                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * GAT_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * GAT_WAVES;
  for (int64_t r = wave; r < N; r += nwaves) {
    const int p_lo = t_rowptr[r], p_hi = t_rowptr[r + 1];
    float s = 0.f;
    for (int p = p_lo + lane; p < p_hi; p += 64) s += v[t_perm ? t_perm[p] : p];
    s = ceeg_wave_sum(s);
    if (lane == 0) out[r] = s;
  }
}

int gat_blocks(int64_t N) {
  int64_t b = (N + GAT_WAVES - 1) / GAT_WAVES;
  if (b < 1) b = 1;
  return (int)(b < GAT_MAX_BLOCKS ? b : GAT_MAX_BLOCKS);
}
int gat_lpr(int C) {
  const int need = (C + 3) / 4;
  int l = 1;
  while (l < need) l <<= 1;
  return l;
}

}  // namespace

extern "C" int ceeg_gat_softmax_fwd(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                                    int64_t N, int64_t nnz, float slope, float* w, void* stream) {
  if (!rowptr || !a_src || !a_dst || N <= 0 || nnz < 0 || (nnz > 0 && (!col || !w))) return CEEG_EINVAL;
  if (N >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  hipLaunchKernelGGL(gat_softmax_fwd_kernel, dim3(gat_blocks(N)), dim3(GAT_WAVES * 64), 0, static_cast<hipStream_t>(stream), rowptr,
                     col, a_src, a_dst, N, slope, w);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_gat_attn_bwd(const int32_t* rowptr, const int32_t* col, const float* w, const float* a_src,
                                 const float* a_dst, const float* g, int64_t ldg, const float* xp, int64_t ldx, int64_t N,
                                 int64_t nnz, int C, float slope, float* dl, float* d_adst, void* stream) {
  if (!rowptr || !a_src || !a_dst || !g || !xp || !d_adst || N <= 0 || nnz < 0 || (nnz > 0 && (!col || !w || !dl))) return CEEG_EINVAL;
  if (C < 4 || (C % 4) != 0 || C > 256 || ldg < C || ldx < C || (ldg % 4) != 0 || (ldx % 4) != 0) return CEEG_EUNSUPPORTED;
  if (N >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(g, 16) || !ceeg_aligned(xp, 16)) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define CEEG_GAT_CASE(L)                                                                                                       \
  case L:                                                                                                                      \
    hipLaunchKernelGGL(gat_attn_bwd_kernel<L>, dim3(gat_blocks(N)), dim3(GAT_WAVES * 64), 0, st, rowptr, col, w, a_src, a_dst, g, \
                       ldg, xp, ldx, N, C, slope, dl, d_adst);                                                                 \
    break;
  switch (gat_lpr(C)) {
    CEEG_GAT_CASE(1)
    CEEG_GAT_CASE(2)
    CEEG_GAT_CASE(4)
    CEEG_GAT_CASE(8)
    CEEG_GAT_CASE(16)
    CEEG_GAT_CASE(32)
    CEEG_GAT_CASE(64)
    default: return CEEG_EUNSUPPORTED;
  }
#undef CEEG_GAT_CASE
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_segment_sum_perm(const int32_t* t_rowptr, const int32_t* t_perm, const float* v, int64_t N,
                                     float* out, void* stream) {
  if (!t_rowptr || !v || !out || N <= 0) return CEEG_EINVAL;
  hipLaunchKernelGGL(segment_sum_perm_kernel, dim3(gat_blocks(N)), dim3(GAT_WAVES * 64), 0, static_cast<hipStream_t>(stream), t_rowptr,
                     t_perm, v, N, out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}
