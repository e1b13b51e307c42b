// Deterministic CSR segmented SpMM for gfx950 — the GCNConv aggregation (upstream PyG GCNConv.propagate,
// SURVEY.md §8 A5) and every transposed / decoder-backward pass.  HBM/L2-bound gather: no MFMA here.
//
//   out[i,:] = act( sum_p v(p) * X[col[p],:] + bias )
//
// Mapping: one wave64 per output row.  A feature row of F fp32 is covered by LPR = F/4 lanes with one
// 16-byte load each (F=256: the whole wave = one 1 KiB coalesced gather per nonzero; F=128: two
// nonzeros per wave-instruction).  col/val are fetched 64 at a time with one coalesced load and
// broadcast from registers (v_readlane / ds_bpermute), so the only per-nonzero memory traffic is the
// gathered row.  8 gathers are kept in flight per wave.
// Order of summation (=> bitwise reproducible, no atomics):
//   * lane group g (of G = 64/LPR) adds the row's entries j = g, g+G, ... in ascending order, the G group
//     sums are then combined by a fixed xor-butterfly;
//   * rows longer than SPMM_CHUNK are cut at the GLOBAL positions p = k*SPMM_CHUNK: spmm_long_kernel sums
//     each piece with one wave into `partial`, spmm_rows_kernel adds the pieces in ascending order.
//     (hub rows of the transposed event graph reach 1e4-1e5 entries; one wave per hub would serialise.)
// XCD-aware: workgroups with equal blockIdx%8 (= same XCD, same L2) get one contiguous range of rows, so
// time-local rows of the event DAG share their gathered neighbours in one L2.
#include "ceeg_internal.h"

namespace {

constexpr int SPMM_CHUNK = 256;  // nonzeros per long-row piece
constexpr int SPMM_WAVES = 4;    // rows (waves) per workgroup

template <int LPR>
__device__ __forceinline__ float4 seg_sum(const int32_t* __restrict__ col, const float* __restrict__ val,
                                          const int32_t* __restrict__ vperm, const float* __restrict__ X, int64_t ldx,
                                          int p_lo, int p_hi, int lane, int foff, bool fl_ok, const CeegSpmmEx& ex) {
  constexpr int G = 64 / LPR;
  const int sub = lane / LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = p_lo; base < p_hi; base += 64) {
    const int n = p_hi - base < 64 ? p_hi - base : 64;
    int my_c = 0;
    float my_v = 0.f;
    if (lane < n) {
      my_c = col[base + lane];
      if (ex.col_a != 1 || ex.col_b != 0) my_c = (int)(((uint64_t)ex.col_a * (uint64_t)my_c + (uint64_t)ex.col_b) % (uint64_t)ex.mod);
      my_v = vperm ? val[vperm[base + lane]] : val[base + lane];
    }
#pragma unroll 8
    for (int i = 0; i < 64 / G; ++i) {
      if (i * G >= n) break;  // wave-uniform
      int c;
      float v;
      if constexpr (G == 1) {
        c = __builtin_amdgcn_readlane(my_c, i);
        v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_v), i));
      } else {
        const int j = i * G + sub;
        c = __shfl(my_c, j, 64);
        v = __shfl(my_v, j, 64);  // 0 for j >= n: contributes exactly nothing below
      }
      if (fl_ok && (G == 1 || i * G + sub < n)) {
        const float4 x = *reinterpret_cast<const float4*>(X + (int64_t)c * ldx + foff);
        acc.x = __fmaf_rn(v, x.x, acc.x);
        acc.y = __fmaf_rn(v, x.y, acc.y);
        acc.z = __fmaf_rn(v, x.z, acc.z);
        acc.w = __fmaf_rn(v, x.w, acc.w);
      }
    }
  }
  if constexpr (G > 1) {
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) {
      acc.x += __shfl_xor(acc.x, o, 64);
      acc.y += __shfl_xor(acc.y, o, 64);
      acc.z += __shfl_xor(acc.z, o, 64);
      acc.w += __shfl_xor(acc.w, o, 64);
    }
  }
  return acc;
}

__device__ __forceinline__ int row_of(const int32_t* __restrict__ rowptr, int64_t N, int p) {
  int64_t lo = 0, hi = N;  // last r in [0,N) with rowptr[r] <= p
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= p) lo = mid; else hi = mid;
  }
  return (int)lo;
}

// pieces of long rows: wave w handles global positions [w*CHUNK, (w+1)*CHUNK)
template <int LPR>
__global__ __launch_bounds__(SPMM_WAVES * 64) void spmm_long_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ vperm, const float* __restrict__ X, int64_t ldx, int64_t N, int F, int fbase,
    float* __restrict__ partial, CeegSpmmEx ex) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * SPMM_WAVES + (threadIdx.x >> 6);
  const int nnz = rowptr[N];
  const int64_t p0l = w * SPMM_CHUNK;
  if (p0l >= nnz) return;
  const int p0 = (int)p0l;
  const int p1 = p0 + SPMM_CHUNK < nnz ? p0 + SPMM_CHUNK : nnz;
  const int fl = lane % LPR;
  const int foff = fbase + 4 * fl;
  const bool fl_ok = foff < F;
  const int r0 = row_of(rowptr, N, p0);
  const int r1 = row_of(rowptr, N, p1 - 1);
  if (rowptr[r0 + 1] - rowptr[r0] > SPMM_CHUNK) {  // slot 0: the long row that covers p0
    const int e = rowptr[r0 + 1] < p1 ? rowptr[r0 + 1] : p1;
    const float4 a = seg_sum<LPR>(col, val, vperm, X, ldx, p0, e, lane, foff, fl_ok, ex);
    if (lane < LPR && fl_ok) *reinterpret_cast<float4*>(partial + ((size_t)w * 2 + 0) * (4 * LPR) + 4 * fl) = a;
  }
  if (r1 != r0 && rowptr[r1 + 1] - rowptr[r1] > SPMM_CHUNK) {  // slot 1: a long row that starts inside
    const float4 a = seg_sum<LPR>(col, val, vperm, X, ldx, rowptr[r1], p1, lane, foff, fl_ok, ex);
    if (lane < LPR && fl_ok) *reinterpret_cast<float4*>(partial + ((size_t)w * 2 + 1) * (4 * LPR) + 4 * fl) = a;
  }
}

template <int LPR>
__global__ __launch_bounds__(SPMM_WAVES * 64) void spmm_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ vperm, const float* __restrict__ X, int64_t ldx, int64_t N, int F, int fbase,
    const float* __restrict__ bias, int relu, int accumulate, float* __restrict__ out, int64_t ldo,
    const float* __restrict__ partial, int64_t nblk, CeegSpmmEx ex) {
  // XCD-aware remap: the blocks sharing blockIdx%8 walk one contiguous slab of rows
  const int64_t per = (nblk + 7) / 8;
  const int64_t vb = (int64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (vb >= nblk) return;
  const int lane = threadIdx.x & 63;
  const int64_t r = vb * SPMM_WAVES + (threadIdx.x >> 6);
  if (r >= N) return;
  const int fl = lane % LPR;
  const int foff = fbase + 4 * fl;
  const bool fl_ok = foff < F;
  const int p_lo = rowptr[r], p_hi = rowptr[r + 1];
  float4 acc;
  if (p_hi - p_lo <= SPMM_CHUNK) {
    acc = seg_sum<LPR>(col, val, vperm, X, ldx, p_lo, p_hi, lane, foff, fl_ok, ex);
  } else {
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c_lo = p_lo / SPMM_CHUNK, c_hi = (p_hi - 1) / SPMM_CHUNK;
    if (lane < LPR && fl_ok) {
      for (int c = c_lo; c <= c_hi; ++c) {  // pieces in ascending position order
        const int slot = (c == c_lo && p_lo != c * SPMM_CHUNK) ? 1 : 0;
        const float4 a = *reinterpret_cast<const float4*>(partial + ((size_t)c * 2 + slot) * (4 * LPR) + 4 * fl);
        acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
      }
    }
  }
  if (lane < LPR && fl_ok) {
    if (ex.scale) {
      const float sc = ex.scale[0];
      acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + foff);
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    }
    int64_t ro = r;
    if (ex.row_a != 1 || ex.row_b != 0) ro = (int64_t)(((uint64_t)ex.row_a * (uint64_t)r + (uint64_t)ex.row_b) % (uint64_t)ex.mod);
    float* o = out + ro * ldo + foff;
    if (accumulate) {
      const float4 t = *reinterpret_cast<const float4*>(o);
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    if (relu) {
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
    }
    *reinterpret_cast<float4*>(o) = acc;
  }
}

int lpr_for(int F) {
  const int need = (F + 3) / 4;
  int l = 1;
  while (l < need && l < 64) l <<= 1;
  return l;
}

template <int LPR>
int launch_pass(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm, const float* X,
                int64_t ldx, int64_t N, int64_t nnz, int F, int fbase, const float* bias, int relu, int accumulate,
                float* out, int64_t ldo, float* partial, hipStream_t st, const CeegSpmmEx& ex) {
  const int64_t nchunk = (nnz + SPMM_CHUNK - 1) / SPMM_CHUNK;
  if (nchunk > 0) {
    hipLaunchKernelGGL(spmm_long_kernel<LPR>, dim3((unsigned)((nchunk + SPMM_WAVES - 1) / SPMM_WAVES)), dim3(SPMM_WAVES * 64), 0, st,
                       rowptr, col, val, vperm, X, ldx, N, F, fbase, partial, ex);
    CEEG_CHECK_LAUNCH();
  }
  const int64_t nblk = (N + SPMM_WAVES - 1) / SPMM_WAVES;
  const int64_t grid = (nblk + 7) / 8 * 8;
  hipLaunchKernelGGL(spmm_rows_kernel<LPR>, dim3((unsigned)grid), dim3(SPMM_WAVES * 64), 0, st, rowptr, col, val, vperm, X, ldx, N, F,
                     fbase, bias, relu, accumulate, out, ldo, partial, nblk, ex);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

}  // namespace

extern "C" size_t ceeg_spmm_ws_bytes(int64_t N, int64_t nnz, int F) {
  (void)N;
  if (nnz < 0 || F <= 0) return 0;
  const int lpr = lpr_for(F);
  const int64_t nchunk = (nnz + SPMM_CHUNK - 1) / SPMM_CHUNK;
  return ceeg_ws_part((size_t)(nchunk + 1) * 2 * 4 * lpr, 4);
}

int ceeg_spmm_launch(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                     const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                     int accumulate, float* out, int64_t ldo, void* ws, size_t ws_bytes, hipStream_t st,
                     const CeegSpmmEx& ex) {
  if (N <= 0 || nnz < 0 || !rowptr || !X || !out || (nnz > 0 && (!col || !val))) return CEEG_EINVAL;
  if (F < 4 || F > 1024 || (F % 4) != 0 || (ldx % 4) != 0 || (ldo % 4) != 0 || ldx < F || ldo < F) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(X, 16) || !ceeg_aligned(out, 16) || (bias && !ceeg_aligned(bias, 16))) return CEEG_EINVAL;
  if (nnz >= ((int64_t)1 << 31) || N >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (ws_bytes < ceeg_spmm_ws_bytes(N, nnz, F) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  float* partial = static_cast<float*>(ws);
  const int lpr = lpr_for(F);
  int rc = CEEG_OK;
  for (int fbase = 0; fbase < F && rc == CEEG_OK; fbase += 4 * lpr) {
#define CEEG_SPMM_CASE(L)                                                                                                       \
  case L:                                                                                                                       \
    rc = launch_pass<L>(rowptr, col, val, vperm, X, ldx, N, nnz, F, fbase, bias, relu, accumulate, out, ldo, partial, st, ex); \
    break;
    switch (lpr) {
      CEEG_SPMM_CASE(1)
      CEEG_SPMM_CASE(2)
      CEEG_SPMM_CASE(4)
      CEEG_SPMM_CASE(8)
      CEEG_SPMM_CASE(16)
      CEEG_SPMM_CASE(32)
      CEEG_SPMM_CASE(64)
      default: rc = CEEG_EUNSUPPORTED;
    }
#undef CEEG_SPMM_CASE
  }
  return rc;
}

extern "C" int ceeg_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                             const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                             int accumulate, float* out, int64_t ldo, void* ws, size_t ws_bytes, void* stream) {
  CeegSpmmEx ex;
  ex.mod = N;
  return ceeg_spmm_launch(rowptr, col, val, vperm, X, ldx, N, nnz, F, bias, relu, accumulate, out, ldo, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), ex);
}
