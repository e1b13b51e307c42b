// InnerProductDecoder + GAE.recon_loss for gfx950 (upstream PyG semantics), literal
// "+EPS" form:  v = <z_i, z_j>;  pos: -log(sigmoid(v)+EPS);  neg: -log((1-sigmoid(v))+EPS), EPS = 1e-15.
//
// HBM/L2-bound gathers: each 16-byte piece of a latent row is one lane's load, the dot product is a
// cross-lane butterfly, the per-edge derivative d(term)/dv is stored so the backward is a pair of
// deterministic SpMMs (spmm.hip) — no float atomics anywhere.  Loss sums: thread -> wave butterfly ->
// block -> one fp64 pass over the block partials (fixed order).
#include "ceeg_internal.h"

namespace {

constexpr float DEC_EPS = 1e-15f;
constexpr int DEC_WAVES = 4;
constexpr int DEC_MAX_BLOCKS = 4096;

__device__ __forceinline__ float dec_sigmoid(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ void dec_pos(float v, float& term, float& coef) {
  const float s = dec_sigmoid(v);
  term = -logf(s + DEC_EPS);
  coef = -(s * (1.0f - s)) / (s + DEC_EPS);
}
__device__ __forceinline__ void dec_neg(float v, float& term, float& coef) {
  const float s = dec_sigmoid(v);
  const float om = 1.0f - s;
  term = -logf(om + DEC_EPS);
  coef = (s * om) / (om + DEC_EPS);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return __fmaf_rn(a.w, b.w, __fmaf_rn(a.z, b.z, __fmaf_rn(a.y, b.y, a.x * b.x)));
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {  // sum over the LPR lanes of a group (same result in all)
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block partial of up to 3 running sums -> part[blockIdx][3]
__device__ __forceinline__ void dec_block_out(float s0, float s1, float s2, float* red /* [3][DEC_WAVES] */, float* part) {
  s0 = ceeg_wave_sum(s0); s1 = ceeg_wave_sum(s1); s2 = ceeg_wave_sum(s2);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave] = s0; red[DEC_WAVES + wave] = s1; red[2 * DEC_WAVES + wave] = s2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < DEC_WAVES; ++w) t += red[threadIdx.x * DEC_WAVES + w];
    part[(size_t)blockIdx.x * 3 + threadIdx.x] = t;
  }
}

__global__ void dec_finalize_kernel(const float* __restrict__ part, int nblk, int nout, float* __restrict__ out) {
  __shared__ double red[256];
  for (int k = 0; k < nout; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)part[(size_t)i * 3 + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = (float)red[0];
    __syncthreads();
  }
}

// ------------------------------------------------------------------ edge-list form (any order)
template <int LPR>
__global__ __launch_bounds__(DEC_WAVES * 64) void edge_bce_fwd_kernel(
    const float* __restrict__ z, int64_t ldz, const int32_t* __restrict__ ei, const int32_t* __restrict__ ej, int64_t E, int D,
    int negative, float* __restrict__ coef, float* __restrict__ prob, float* __restrict__ part) {
  __shared__ float red[3 * DEC_WAVES];
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int fl = lane % LPR, sub = lane / LPR;
  const bool fl_ok = 4 * fl < D;
  const int64_t wave = (int64_t)blockIdx.x * DEC_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * DEC_WAVES;
  float acc = 0.f;
  for (int64_t e0 = wave * G; e0 < E; e0 += nwaves * G) {
    const int64_t e = e0 + sub;
    const bool ok = e < E;
    float d = 0.f;
    if (ok && fl_ok) {
      const float4 a = *reinterpret_cast<const float4*>(z + (int64_t)ei[e] * ldz + 4 * fl);
      const float4 b = *reinterpret_cast<const float4*>(z + (int64_t)ej[e] * ldz + 4 * fl);
      d = dot4(a, b);
    }
    d = group_sum<LPR>(d);
    if (ok && fl == 0) {
      float t, c;
      if (negative) dec_neg(d, t, c); else dec_pos(d, t, c);
      if (coef) coef[e] = c;
      if (prob) prob[e] = dec_sigmoid(d);
      acc += t;
    }
  }
  dec_block_out(acc, 0.f, 0.f, red, part);
}

// ------------------------------------------------------------------ fused CSR form, permutation-structured negatives
__device__ __forceinline__ bool sorted_has(const int32_t* __restrict__ col, int lo, int end, int32_t s) {
  int hi = end;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < s) lo = mid + 1; else hi = mid;
  }
  return lo < end && col[lo] == s;
}

template <int LPR>
__global__ __launch_bounds__(DEC_WAVES * 64) void decoder_fused_fwd_kernel(
    const float* __restrict__ z, int64_t ldz, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N, int D,
    int64_t sig_a, int64_t sig_b, float* __restrict__ coef_pos, float* __restrict__ coef_neg, float* __restrict__ part) {
  __shared__ float red[3 * DEC_WAVES];
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int fl = lane % LPR, sub = lane / LPR;
  const bool fl_ok = 4 * fl < D;
  const int64_t wave = (int64_t)blockIdx.x * DEC_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * DEC_WAVES;
  float s_pos = 0.f, s_neg = 0.f, s_cnt = 0.f;
  // contiguous slab of rows per wave: neighbouring rows (time-local events) stay on one CU / XCD
  const int64_t per = (N + nwaves - 1) / nwaves;
  const int64_t r_lo = wave * per, r_hi = r_lo + per < N ? r_lo + per : N;
  for (int64_t r = r_lo; r < r_hi; ++r) {
    const int p_lo = rowptr[r], p_hi = rowptr[r + 1];
    if (p_lo == p_hi) continue;
    const int64_t sr = (int64_t)(((uint64_t)sig_a * (uint64_t)r + (uint64_t)sig_b) % (uint64_t)N);
    float4 zr = make_float4(0.f, 0.f, 0.f, 0.f), zs = zr;
    if (fl_ok) {
      zr = *reinterpret_cast<const float4*>(z + r * ldz + 4 * fl);
      zs = *reinterpret_cast<const float4*>(z + sr * ldz + 4 * fl);
    }
    const int s_lo = rowptr[sr], s_hi = rowptr[sr + 1];
    for (int base = p_lo; base < p_hi; base += 64) {
      const int n = p_hi - base < 64 ? p_hi - base : 64;
      int my_c = 0, my_valid = 0;
      if (lane < n) {
        my_c = col[base + lane];
        my_valid = (my_c != (int)sr && !sorted_has(col, s_lo, s_hi, my_c)) ? 1 : 0;
      }
      const int ns = (n + G - 1) / G;  // steps in this batch (wave-uniform)
      for (int i0 = 0; i0 < ns; i0 += 4) {
        // up to 4 gathers in flight, then the 4 pairs of dot products
        float4 x[4];
        int cv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = (i0 + k) * G + sub;
          const int c = __shfl(my_c, j & 63, 64);
          cv[k] = __shfl(my_valid, j & 63, 64);
          x[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (i0 + k < ns && j < n && fl_ok) x[k] = *reinterpret_cast<const float4*>(z + (int64_t)c * ldz + 4 * fl);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (i0 + k >= ns) break;  // wave-uniform
          const int j = (i0 + k) * G + sub;
          const bool ok = j < n;
          const float dp = group_sum<LPR>(dot4(x[k], zr));
          const float dn = group_sum<LPR>(dot4(x[k], zs));
          if (ok && fl == 0) {
            float tp, cp, tn, cn;
            dec_pos(dp, tp, cp);
            dec_neg(dn, tn, cn);
            coef_pos[base + j] = cp;
            coef_neg[base + j] = cv[k] ? cn : 0.f;
            s_pos += tp;
            if (cv[k]) { s_neg += tn; s_cnt += 1.0f; }
          }
        }
      }
    }
  }
  dec_block_out(s_pos, s_neg, s_cnt, red, part);
}

int dec_lpr(int D) {
  const int need = (D + 3) / 4;
  int l = 1;
  while (l < need) l <<= 1;
  return l;
}

}  // namespace

extern "C" size_t ceeg_edge_bce_ws_bytes(int64_t E, int D) {
  if (E < 0 || D <= 0) return 0;
  return ceeg_ws_part((size_t)DEC_MAX_BLOCKS * 3, 4);
}

extern "C" int ceeg_edge_bce_fwd(const float* z, int64_t ldz, const int32_t* i, const int32_t* j, int64_t E, int D,
                                 int negative, float* coef, float* prob, float* loss_sum,
                                 void* ws, size_t ws_bytes, void* stream) {
  if (!z || !loss_sum || E < 0 || D <= 0 || (E > 0 && (!i || !j))) return CEEG_EINVAL;
  if ((D % 4) != 0 || D > 256 || (ldz % 4) != 0 || ldz < D) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(z, 16)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_edge_bce_ws_bytes(E, D) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(ws);
  const int lpr = dec_lpr(D);
  const int G = 64 / lpr;
  int64_t nb = (E + (int64_t)G * DEC_WAVES - 1) / ((int64_t)G * DEC_WAVES);
  if (nb < 1) nb = 1;
  const int nblk = (int)(nb < DEC_MAX_BLOCKS ? nb : DEC_MAX_BLOCKS);
#define CEEG_EDGE_CASE(L)                                                                                                              \
  case L:                                                                                                                              \
    hipLaunchKernelGGL(edge_bce_fwd_kernel<L>, dim3(nblk), dim3(DEC_WAVES * 64), 0, st, z, ldz, i, j, E, D, negative, coef, prob, part); \
    break;
  switch (lpr) {
    CEEG_EDGE_CASE(1)
    CEEG_EDGE_CASE(2)
    CEEG_EDGE_CASE(4)
    CEEG_EDGE_CASE(8)
    CEEG_EDGE_CASE(16)
    CEEG_EDGE_CASE(32)
    CEEG_EDGE_CASE(64)
    default: return CEEG_EUNSUPPORTED;
  }
#undef CEEG_EDGE_CASE
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(dec_finalize_kernel, dim3(1), dim3(256), 0, st, part, nblk, 1, loss_sum);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" size_t ceeg_decoder_fused_ws_bytes(int64_t N, int64_t nnz, int D) {
  if (N <= 0 || nnz < 0 || D <= 0) return 0;
  const size_t a = ceeg_ws_part((size_t)DEC_MAX_BLOCKS * 3, 4);
  const size_t b = ceeg_spmm_ws_bytes(N, nnz, D);
  return a > b ? a : b;
}

extern "C" int ceeg_decoder_fused_fwd(const float* z, int64_t ldz, const int32_t* rowptr, const int32_t* col,
                                      int64_t N, int64_t nnz, int D, int64_t sig_a, int64_t sig_b,
                                      float* coef_pos, float* coef_neg, float* sums,
                                      void* ws, size_t ws_bytes, void* stream) {
  if (!z || !rowptr || !sums || N <= 0 || nnz < 0 || D <= 0 || (nnz > 0 && (!col || !coef_pos || !coef_neg))) return CEEG_EINVAL;
  if ((D % 4) != 0 || D > 256 || (ldz % 4) != 0 || ldz < D || N >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (sig_a <= 0 || sig_a >= N + (N == 1) || sig_b < 0 || sig_b >= N) return CEEG_EINVAL;
  if (!ceeg_aligned(z, 16)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_decoder_fused_ws_bytes(N, nnz, D) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(ws);
  const int lpr = dec_lpr(D);
  int64_t nb = (N + DEC_WAVES * 4 - 1) / (DEC_WAVES * 4);  // ~4 rows per wave at least
  if (nb < 1) nb = 1;
  const int nblk = (int)(nb < DEC_MAX_BLOCKS ? nb : DEC_MAX_BLOCKS);
#define CEEG_DEC_CASE(L)                                                                                                      \
  case L:                                                                                                                     \
    hipLaunchKernelGGL(decoder_fused_fwd_kernel<L>, dim3(nblk), dim3(DEC_WAVES * 64), 0, st, z, ldz, rowptr, col, N, D, sig_a, \
                       sig_b, coef_pos, coef_neg, part);                                                                      \
    break;
  switch (lpr) {
    CEEG_DEC_CASE(1)
    CEEG_DEC_CASE(2)
    CEEG_DEC_CASE(4)
    CEEG_DEC_CASE(8)
    CEEG_DEC_CASE(16)
    CEEG_DEC_CASE(32)
    CEEG_DEC_CASE(64)
    default: return CEEG_EUNSUPPORTED;
  }
#undef CEEG_DEC_CASE
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(dec_finalize_kernel, dim3(1), dim3(256), 0, st, part, nblk, 3, sums);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_decoder_fused_bwd(const float* z, int64_t ldz, const int32_t* rowptr, const int32_t* col,
                                      const int32_t* t_rowptr, const int32_t* t_col, const int32_t* t_perm,
                                      int64_t N, int64_t nnz, int D, int64_t sig_a, int64_t sig_b, int64_t sig_ainv,
                                      const float* coef_pos, const float* coef_neg, const float* gscale,
                                      float* dz, int64_t lddz, int64_t max_row_nnz, int64_t max_trow_nnz,
                                      void* ws, size_t ws_bytes, void* stream) {
  if (!z || !rowptr || !t_rowptr || !gscale || !dz || N <= 0 || nnz < 0) return CEEG_EINVAL;
  if (nnz > 0 && (!col || !t_col || !t_perm || !coef_pos || !coef_neg)) return CEEG_EINVAL;
  if (sig_a <= 0 || sig_b < 0 || sig_b >= N || sig_ainv <= 0) return CEEG_EINVAL;
  if ((uint64_t)((unsigned __int128)(uint64_t)sig_a * (uint64_t)sig_ainv % (uint64_t)N) != (uint64_t)(1 % N)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_decoder_fused_ws_bytes(N, nnz, D)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  // (1) n as dst of positives:            dz[n]  = gp * sum_{p in row n} cp[p] z[col p]
  CeegSpmmEx e1; e1.mod = N; e1.scale = gscale + 0; e1.max_row_nnz = max_row_nnz;
  CeegSpmmEx e2 = e1; e2.max_row_nnz = max_trow_nnz;
  rc = ceeg_spmm_launch(rowptr, col, coef_pos, nullptr, z, ldz, N, nnz, D, nullptr, 0, 0, dz, lddz, ws, ws_bytes, st, e1);
  if (rc != CEEG_OK) return rc;
  // (2) n as src of positives:            dz[n] += gp * sum_{q in trow n} cp[tperm q] z[tcol q]
  rc = ceeg_spmm_launch(t_rowptr, t_col, coef_pos, t_perm, z, ldz, N, nnz, D, nullptr, 0, 1, dz, lddz, ws, ws_bytes, st, e2);
  if (rc != CEEG_OK) return rc;
  // (3) n = sigma(r) as negative endpoint: dz[sigma(r)] += gn * sum_{p in row r} cn[p] z[col p]
  CeegSpmmEx e3; e3.mod = N; e3.scale = gscale + 1; e3.row_a = sig_a; e3.row_b = sig_b; e3.max_row_nnz = max_row_nnz;
  rc = ceeg_spmm_launch(rowptr, col, coef_neg, nullptr, z, ldz, N, nnz, D, nullptr, 0, 1, dz, lddz, ws, ws_bytes, st, e3);
  if (rc != CEEG_OK) return rc;
  // (4) n as src of negatives:            dz[n] += gn * sum_{q in trow n} cn[tperm q] z[sigma(tcol q)]
  CeegSpmmEx e4; e4.mod = N; e4.scale = gscale + 1; e4.col_a = sig_a; e4.col_b = sig_b; e4.max_row_nnz = max_trow_nnz;
  rc = ceeg_spmm_launch(t_rowptr, t_col, coef_neg, t_perm, z, ldz, N, nnz, D, nullptr, 0, 1, dz, lddz, ws, ws_bytes, st, e4);
  return rc;
}
