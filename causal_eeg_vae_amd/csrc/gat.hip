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

// LPR == 64 (128 < C <= 256: one entry's row spans the whole wave): 8 entries per round, 8 gathers in flight, and ONE
// transposing reduction for the 8 dots instead of 8 butterflies - v_permlane32_swap / v_permlane16_swap halve the number of
// live values while doubling what each covers (2 VALU per value, no LDS traffic), one select + row rotation does the same
// inside the 16-lane rows, three DPP adds finish inside the 8-lane groups.
__device__ __forceinline__ float gat_swap_add32(float a, float b) {  // lanes < 32: sum of a over the lane pairs (l, l+32); lanes >= 32: of b
  const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ float gat_swap_add16(float a, float b) {  // even 16-lane rows: a summed over the row pairs; odd rows: b
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
template <int CTRL>
__device__ __forceinline__ float gat_dpp(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false)); }

__device__ __forceinline__ float gat_batch_dots64(const float* __restrict__ xp, int64_t ldx, int my_c, int n, int lane, bool fl_ok,
                                                  const float4& gr, const float* __restrict__ grow, int C) {
  float mine = 0.f;
  const int rounds = n < 8 ? n : 8;  // round t: entries 8k + t, k = 0..7 (those below n); measured faster than entries 8t + k
  for (int t = 0; t < rounds; ++t) {   // (2.14 vs 2.33 ms on the 1M/16M graph) although a 17-entry row then takes 8 rounds of <= 3 gathers
    float d[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int j = 8 * k + t;
      const int c = __builtin_amdgcn_readlane(my_c, j);
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < n && fl_ok) x = *reinterpret_cast<const float4*>(xp + (int64_t)c * ldx + 4 * lane);
      d[k] = gat_dot4(x, gr);
    }
    // heads wider than 256 columns (the reference's 2048-wide layers): further 256-column chunks, the chunk of g[row]
    // fetched once per round and shared by the 8 entries
    for (int c0 = 256; c0 < C; c0 += 256) {
      const bool ok = c0 + 4 * lane < C;
      float4 gc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) gc = *reinterpret_cast<const float4*>(grow + c0 + 4 * lane);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int j = 8 * k + t;
        const int c = __builtin_amdgcn_readlane(my_c, j);
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < n && ok) x = *reinterpret_cast<const float4*>(xp + (int64_t)c * ldx + c0 + 4 * lane);
        d[k] += gat_dot4(x, gc);
      }
    }
    // the lanes with (b5, b4, b3) = k end up holding dot k - the group lane 8k + t lives in: no final shuffle
    const float a0 = gat_swap_add32(d[0], d[4]), a1 = gat_swap_add32(d[1], d[5]);
    const float a2 = gat_swap_add32(d[2], d[6]), a3 = gat_swap_add32(d[3], d[7]);
    const float b0 = gat_swap_add16(a0, a2), b1 = gat_swap_add16(a1, a3);
    const bool hi8 = (lane & 8) != 0;
    const float keep = hi8 ? b1 : b0, send = hi8 ? b0 : b1;
    float v = keep + gat_dpp<0x128>(send);  // row_ror:8: the partner half-row's share of the value this half keeps
    v += gat_dpp<0x141>(v);                 // row_half_mirror
    v += gat_dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
    v += gat_dpp<0xB1>(v);                  // quad_perm [1,0,3,2]
    if ((lane & 7) == t) mine = v;
  }
  return lane < n ? mine : 0.f;
}

template <int LPR>
__device__ __forceinline__ float gat_dots(const float* __restrict__ xp, int64_t ldx, int my_c, int n, int lane, bool fl_ok, const float4& gr,
                                          const float* __restrict__ grow, int C) {
  if constexpr (LPR == 64) return gat_batch_dots64(xp, ldx, my_c, n, lane, fl_ok, gr, grow, C);
  else return gat_batch_dots<LPR>(xp, ldx, my_c, n, lane, fl_ok, gr);
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
      const float dw = gat_dots<LPR>(xp, ldx, my_c, n, lane, fl_ok, gr, g + r * ldg, C);
      if (base == p_lo) dw0 = dw;
      else if (lane < n) dl[base + lane] = dw;  // parked in the output slot: pass 2 reads it back instead of gathering the rows again
      if (lane < n) s = __fmaf_rn(w[base + lane], dw, s);
    }
    s = ceeg_wave_sum(s);
    // pass 2: de = w (dw - s); through the leaky relu; row sum -> d a_dst
    float da = 0.f;
    for (int base = p_lo; base < p_hi; base += 64) {
      const int n = p_hi - base < 64 ? p_hi - base : 64;
      if (lane < n) {
        const float dw = (one_batch || base == p_lo) ? dw0 : dl[base + lane];
        const float de = w[base + lane] * (dw - s);
        const float raw = a_src[col[base + lane]] + ad;
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
                                                                          float* __restrict__ out, int skip_longer_than) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * GAT_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * GAT_WAVES;
  for (int64_t r = wave; r < N; r += nwaves) {
    const int p_lo = t_rowptr[r], p_hi = t_rowptr[r + 1];
    if (p_hi - p_lo > skip_longer_than) continue;  // summed by pieces (segment_sum_piece_kernel)
    float s = 0.f;
    for (int p = p_lo + lane; p < p_hi; p += 64) s += v[t_perm ? t_perm[p] : p];
    s = ceeg_wave_sum(s);
    if (lane == 0) out[r] = s;
  }
}

// rows longer than one chunk: a wave per piece (lo, hi, -, row) of the graph's static piece list, then the pieces of a row
// are added in list order by the thread of the row's first piece - a source with 90 k outgoing edges is 360 waves' work
// instead of one wave's 1450 dependent gather rounds
__global__ __launch_bounds__(GAT_WAVES * 64) void segment_sum_piece_kernel(const int4* __restrict__ items, int n_items,
                                                                           const int32_t* __restrict__ t_perm, const float* __restrict__ v,
                                                                           float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * GAT_WAVES + (threadIdx.x >> 6);
  if (i >= n_items) return;
  const int4 it = items[i];
  float s = 0.f;
  for (int p = it.x + lane; p < it.y; p += 64) s += v[t_perm ? t_perm[p] : p];
  s = ceeg_wave_sum(s);
  if (lane == 0) part[i] = s;
}
__global__ __launch_bounds__(256) void segment_sum_piece_finalize_kernel(const int4* __restrict__ items, int n_items,
                                                                         const float* __restrict__ part, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_items) return;
  const int row = items[i].w;
  if (i > 0 && items[i - 1].w == row) return;
  float s = 0.f;
  for (int j = i; j < n_items && items[j].w == row; ++j) s += part[j];
  out[row] = s;
}

int gat_blocks(int64_t N) {
  int64_t b = (N + GAT_WAVES - 1) / GAT_WAVES;
  if (b < 1) b = 1;
  return (int)(b < GAT_MAX_BLOCKS ? b : GAT_MAX_BLOCKS);
}
// ---- attention scores a_src[h][n] = <xp[n, h, :], att_src[h, :]> (and a_dst) in ONE pass over xp: LPR lanes per (n, h),
// one float4 each, xor butterfly inside the group (fixed order).  Stands behind (x_proj * att).sum(-1) of PyG GATConv.
template <int LPR>
__global__ __launch_bounds__(256) void gat_scores_fwd_kernel(const float* __restrict__ xp, int64_t ldx, const float* __restrict__ att_src,
                                                             const float* __restrict__ att_dst, int64_t N, int H, int C,
                                                             float* __restrict__ a_src, float* __restrict__ a_dst) {
  const int l = threadIdx.x % LPR;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
  const int64_t ngrp = (int64_t)gridDim.x * 256 / LPR;
  const int64_t items = N * H;
  for (int64_t it0 = grp; it0 < items + ngrp; it0 += 2 * ngrp) {  // two items in flight per group
    const int64_t itA = it0, itB = it0 + ngrp;
    const bool okA = itA < items, okB = itB < items;
    if (!okA && !okB) break;
    const int64_t nA = okA ? itA / H : 0, nB = okB ? itB / H : 0;
    const int hA = okA ? (int)(itA - nA * H) : 0, hB = okB ? (int)(itB - nB * H) : 0;
    float vs0 = 0.f, vd0 = 0.f, vs1 = 0.f, vd1 = 0.f;
    for (int c = 4 * l; c < C; c += 4 * LPR) {  // one trip for C <= 256; wider heads in chunks of 4 LPR = 256 columns
      const float4 xA = *reinterpret_cast<const float4*>(xp + nA * ldx + hA * C + c);
      const float4 xB = *reinterpret_cast<const float4*>(xp + nB * ldx + hB * C + c);
      const float4 sA = *reinterpret_cast<const float4*>(att_src + hA * C + c);
      const float4 dA = *reinterpret_cast<const float4*>(att_dst + hA * C + c);
      const float4 sB = *reinterpret_cast<const float4*>(att_src + hB * C + c);
      const float4 dB = *reinterpret_cast<const float4*>(att_dst + hB * C + c);
      vs0 += gat_dot4(xA, sA); vd0 += gat_dot4(xA, dA); vs1 += gat_dot4(xB, sB); vd1 += gat_dot4(xB, dB);
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) {
      vs0 += __shfl_xor(vs0, o, 64);
      vd0 += __shfl_xor(vd0, o, 64);
      vs1 += __shfl_xor(vs1, o, 64);
      vd1 += __shfl_xor(vd1, o, 64);
    }
    if (l == 0) {
      if (okA) { a_src[(int64_t)hA * N + nA] = vs0; a_dst[(int64_t)hA * N + nA] = vd0; }
      if (okB) { a_src[(int64_t)hB * N + nB] = vs1; a_dst[(int64_t)hB * N + nB] = vd1; }
    }
  }
}

// ---- backward of the scores, fused with the accumulation into dxp:
//   dxp[n, h, :] += d_asrc[h][n] att_src[h, :] + d_adst[h][n] att_dst[h, :]
//   d_att_src[h, c] = sum_n d_asrc[h][n] xp[n, h, c]      (d_att_dst likewise)
// block b owns rows [b * rows_per, ..); thread t owns column quad t % fq and walks rows t / fq, + rpb, ..; block partials
// in a fixed order, then one fp64 pass over the blocks (the layout of ceeg_colsum).
constexpr int GAT_SB_THREADS = 256;
constexpr int GAT_SB_MAX_BLOCKS = 1024;
__global__ __launch_bounds__(GAT_SB_THREADS) void gat_scores_bwd_kernel(const float* __restrict__ xp, int64_t ldx,
                                                                        const float* __restrict__ att_src, const float* __restrict__ att_dst,
                                                                        const float* __restrict__ d_asrc, const float* __restrict__ d_adst,
                                                                        int64_t N, int F, int col_off, int C, float* dxp, int64_t ldd,
                                                                        int64_t rows_per, float* __restrict__ part /* [grid][2][F] */) {
  // F columns starting at column col_off of the H * C wide matrices (xp, dxp, att_* already point at col_off)
  __shared__ float4 red[2][GAT_SB_THREADS];
  const int fq = F / 4;
  const int rpb = GAT_SB_THREADS / fq;
  const int cq = threadIdx.x % fq, rr = threadIdx.x / fq;
  const int h = (col_off + 4 * cq) / C;
  const int64_t lo = (int64_t)blockIdx.x * rows_per;
  const int64_t hi = lo + rows_per < N ? lo + rows_per : N;
  float4 as = make_float4(0.f, 0.f, 0.f, 0.f), ad = as;
  if (rr < rpb) {
    const float4 ws = *reinterpret_cast<const float4*>(att_src + 4 * cq);
    const float4 wd = *reinterpret_cast<const float4*>(att_dst + 4 * cq);
    const float* ps = d_asrc + (int64_t)h * N;
    const float* pd = d_adst + (int64_t)h * N;
    int64_t n = lo + rr;
    for (; n + rpb < hi; n += 2 * rpb) {  // two rows in flight
      const int64_t m = n + rpb;
      const float s0 = ps[n], d0 = pd[n], s1 = ps[m], d1 = pd[m];
      const float4 x0 = *reinterpret_cast<const float4*>(xp + n * ldx + 4 * cq);
      const float4 x1 = *reinterpret_cast<const float4*>(xp + m * ldx + 4 * cq);
      float4 g0 = *reinterpret_cast<float4*>(dxp + n * ldd + 4 * cq);
      float4 g1 = *reinterpret_cast<float4*>(dxp + m * ldd + 4 * cq);
      g0.x += __fmaf_rn(s0, ws.x, d0 * wd.x); g0.y += __fmaf_rn(s0, ws.y, d0 * wd.y); g0.z += __fmaf_rn(s0, ws.z, d0 * wd.z); g0.w += __fmaf_rn(s0, ws.w, d0 * wd.w);
      g1.x += __fmaf_rn(s1, ws.x, d1 * wd.x); g1.y += __fmaf_rn(s1, ws.y, d1 * wd.y); g1.z += __fmaf_rn(s1, ws.z, d1 * wd.z); g1.w += __fmaf_rn(s1, ws.w, d1 * wd.w);
      *reinterpret_cast<float4*>(dxp + n * ldd + 4 * cq) = g0;
      *reinterpret_cast<float4*>(dxp + m * ldd + 4 * cq) = g1;
      as.x = __fmaf_rn(s0, x0.x, as.x); as.y = __fmaf_rn(s0, x0.y, as.y); as.z = __fmaf_rn(s0, x0.z, as.z); as.w = __fmaf_rn(s0, x0.w, as.w);
      ad.x = __fmaf_rn(d0, x0.x, ad.x); ad.y = __fmaf_rn(d0, x0.y, ad.y); ad.z = __fmaf_rn(d0, x0.z, ad.z); ad.w = __fmaf_rn(d0, x0.w, ad.w);
      as.x = __fmaf_rn(s1, x1.x, as.x); as.y = __fmaf_rn(s1, x1.y, as.y); as.z = __fmaf_rn(s1, x1.z, as.z); as.w = __fmaf_rn(s1, x1.w, as.w);
      ad.x = __fmaf_rn(d1, x1.x, ad.x); ad.y = __fmaf_rn(d1, x1.y, ad.y); ad.z = __fmaf_rn(d1, x1.z, ad.z); ad.w = __fmaf_rn(d1, x1.w, ad.w);
    }
    if (n < hi) {
      const float s0 = ps[n], d0 = pd[n];
      const float4 x0 = *reinterpret_cast<const float4*>(xp + n * ldx + 4 * cq);
      float4 g0 = *reinterpret_cast<float4*>(dxp + n * ldd + 4 * cq);
      g0.x += __fmaf_rn(s0, ws.x, d0 * wd.x); g0.y += __fmaf_rn(s0, ws.y, d0 * wd.y); g0.z += __fmaf_rn(s0, ws.z, d0 * wd.z); g0.w += __fmaf_rn(s0, ws.w, d0 * wd.w);
      *reinterpret_cast<float4*>(dxp + n * ldd + 4 * cq) = g0;
      as.x = __fmaf_rn(s0, x0.x, as.x); as.y = __fmaf_rn(s0, x0.y, as.y); as.z = __fmaf_rn(s0, x0.z, as.z); as.w = __fmaf_rn(s0, x0.w, as.w);
      ad.x = __fmaf_rn(d0, x0.x, ad.x); ad.y = __fmaf_rn(d0, x0.y, ad.y); ad.z = __fmaf_rn(d0, x0.z, ad.z); ad.w = __fmaf_rn(d0, x0.w, ad.w);
    }
  }
  red[0][threadIdx.x] = as;
  red[1][threadIdx.x] = ad;
  __syncthreads();
  if (rr == 0) {  // fixed order over the block's row groups
    for (int k = 1; k < rpb; ++k) {
      const float4 t = red[0][k * fq + cq], u = red[1][k * fq + cq];
      as.x += t.x; as.y += t.y; as.z += t.z; as.w += t.w;
      ad.x += u.x; ad.y += u.y; ad.z += u.z; ad.w += u.w;
    }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * F + 4 * cq) = as;
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * F + F + 4 * cq) = ad;
  }
}

// 16 columns (of the 2F) per block; 16 thread groups walk the block partials in a fixed interleave, fp64, then a fixed-order LDS sum
__global__ __launch_bounds__(256) void gat_scores_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int F2, int F,
                                                                      float* __restrict__ d_att_src, float* __restrict__ d_att_dst) {
  __shared__ double red[256];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int f = blockIdx.x * 16 + c;
  double s = 0.0;
  if (f < F2)
    for (int b = grp; b < nblk; b += 16) s += (double)part[(size_t)b * F2 + f];
  red[threadIdx.x] = s;
  __syncthreads();
  if (grp == 0 && f < F2) {
    for (int k = 1; k < 16; ++k) s += red[k * 16 + c];
    if (f < F) d_att_src[f] = (float)s;
    else d_att_dst[f - F] = (float)s;
  }
}

int gat_lpr(int C) {  // lanes per row: next power of two >= C / 4, at most the wave (wider rows go in 256-column chunks)
  const int need = (C + 3) / 4;
  int l = 1;
  while (l < need && l < 64) l <<= 1;
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
  if (C < 4 || (C % 4) != 0 || C > 2048 || ldg < C || ldx < C || (ldg % 4) != 0 || (ldx % 4) != 0) return CEEG_EUNSUPPORTED;
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
                     t_perm, v, N, out, 0x7fffffff);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" size_t ceeg_segment_sum_items_ws_bytes(int64_t n_items) { return n_items > 0 ? ceeg_ws_part((size_t)n_items, 4) : 0; }

extern "C" int ceeg_segment_sum_perm_items(const int32_t* t_rowptr, const int32_t* t_perm, const float* v, int64_t N, const int32_t* items,
                                           int64_t n_items, int chunk, float* out, void* ws, size_t ws_bytes, void* stream) {
  if (!t_rowptr || !v || !out || N <= 0 || n_items < 0 || chunk <= 0 || (n_items > 0 && !items)) return CEEG_EINVAL;
  if (n_items >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (n_items > 0 && (ws_bytes < ceeg_segment_sum_items_ws_bytes(n_items) || !ws || !ceeg_aligned(ws, 256))) return CEEG_EWORKSPACE;
  if (n_items > 0 && !ceeg_aligned(items, 16)) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(segment_sum_perm_kernel, dim3(gat_blocks(N)), dim3(GAT_WAVES * 64), 0, st, t_rowptr, t_perm, v, N, out,
                     n_items > 0 ? chunk : 0x7fffffff);
  CEEG_CHECK_LAUNCH();
  if (n_items > 0) {
    float* part = static_cast<float*>(ws);
    const int4* its = reinterpret_cast<const int4*>(items);
    hipLaunchKernelGGL(segment_sum_piece_kernel, dim3((unsigned)((n_items + GAT_WAVES - 1) / GAT_WAVES)), dim3(GAT_WAVES * 64), 0, st, its,
                       (int)n_items, t_perm, v, part);
    CEEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(segment_sum_piece_finalize_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st, its, (int)n_items, part, out);
    CEEG_CHECK_LAUNCH();
  }
  return CEEG_OK;
}

extern "C" int ceeg_gat_scores_fwd(const float* xp, int64_t ldx, const float* att_src, const float* att_dst, int64_t N, int H, int C,
                                   float* a_src, float* a_dst, void* stream) {
  if (!xp || !att_src || !att_dst || !a_src || !a_dst || N <= 0 || H <= 0) return CEEG_EINVAL;
  if (C < 4 || (C % 4) != 0 || C > 2048 || ldx < (int64_t)H * C || (ldx % 4) != 0) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(xp, 16) || !ceeg_aligned(att_src, 16) || !ceeg_aligned(att_dst, 16)) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int lpr = gat_lpr(C);
  int64_t blocks = (N * H * lpr + 2 * 256 - 1) / (2 * 256);
  if (blocks > 4 * GAT_MAX_BLOCKS) blocks = 4 * GAT_MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
#define CEEG_GAT_CASE(L)                                                                                                                \
  case L:                                                                                                                               \
    hipLaunchKernelGGL(gat_scores_fwd_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, st, xp, ldx, att_src, att_dst, N, H, C, a_src, a_dst); \
    break;
  switch (lpr) {
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

extern "C" size_t ceeg_gat_scores_bwd_ws_bytes(int64_t N, int H, int C) {
  if (N <= 0 || H <= 0 || C <= 0) return 0;
  const size_t F = (size_t)H * C;
  return ceeg_ws_part((size_t)GAT_SB_MAX_BLOCKS * 2 * (F < 4 * GAT_SB_THREADS ? F : 4 * GAT_SB_THREADS), 4);
}

extern "C" int ceeg_gat_scores_bwd(const float* xp, int64_t ldx, const float* att_src, const float* att_dst, const float* d_asrc,
                                   const float* d_adst, int64_t N, int H, int C, float* dxp, int64_t ldd, float* d_att_src,
                                   float* d_att_dst, void* ws, size_t ws_bytes, void* stream) {
  if (!xp || !att_src || !att_dst || !d_asrc || !d_adst || !dxp || !d_att_src || !d_att_dst || N <= 0 || H <= 0) return CEEG_EINVAL;
  const int64_t F = (int64_t)H * C;
  if (C < 4 || (C % 4) != 0 || C > 2048 || F > 16384 || ldx < F || ldd < F || (ldx % 4) != 0 || (ldd % 4) != 0) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(xp, 16) || !ceeg_aligned(dxp, 16) || !ceeg_aligned(att_src, 16) || !ceeg_aligned(att_dst, 16)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_gat_scores_bwd_ws_bytes(N, H, C) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t nblk = (N + 255) / 256;
  if (nblk > GAT_SB_MAX_BLOCKS) nblk = GAT_SB_MAX_BLOCKS;
  const int64_t rows_per = (N + nblk - 1) / nblk;
  nblk = (N + rows_per - 1) / rows_per;
  float* part = static_cast<float*>(ws);
  for (int64_t off = 0; off < F; off += 4 * GAT_SB_THREADS) {  // column blocks of <= 1024 (one column quad per thread); stream-ordered reuse of part
    const int fb = (int)(F - off < 4 * GAT_SB_THREADS ? F - off : 4 * GAT_SB_THREADS);
    hipLaunchKernelGGL(gat_scores_bwd_kernel, dim3((unsigned)nblk), dim3(GAT_SB_THREADS), 0, st, xp + off, ldx, att_src + off, att_dst + off,
                       d_asrc, d_adst, N, fb, (int)off, C, dxp + off, ldd, rows_per, part);
    CEEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(gat_scores_bwd_finalize_kernel, dim3((unsigned)((2 * fb + 15) / 16)), dim3(256), 0, st, part, (int)nblk, 2 * fb, fb,
                       d_att_src + off, d_att_dst + off);
    CEEG_CHECK_LAUNCH();
  }
  return CEEG_OK;
}
