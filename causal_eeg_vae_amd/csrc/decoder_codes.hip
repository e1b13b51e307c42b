// Fused decoder for a QUANTISED latent (z[n] = W[code[n]], W the VQ codebook) on gfx950 (MI355X).
//
// Same arithmetic contract as decoder.hip (upstream PyG InnerProductDecoder + GAE.recon_loss, literal "+EPS" form,
// permutation-structured negatives), but built on what quantisation gives for free:
//   * every dot product <z_i, z_j> is an entry of the Gram matrix  G = W W^T  (K x K floats, 1 MB at K = 512, L2-resident):
//     the forward reads TWO FLOATS per (entry, kind) instead of gathering a D*4-byte row through L2;
//   * every gradient term is (coefficient) x (codebook row): the backward keeps a 64-column slab of the WHOLE codebook in
//     LDS (K x 64 floats = 128 KB of the CU's 160 KB) and reads its rows with ds_read_b128 - no global gathers of
//     latent rows at all.  D/64 slabs are swept by different workgroups.
// The L2 -> CU gather plateau (~12 TB/s) that bounded decoder.hip's 512-byte row gathers is gone: what
// is left are the index/coefficient streams (16 bytes per entry) and LDS reads.
//
// Forward  : dc_gram_kernel -> dc_coef_rows_kernel (+ dc_coef_long_kernel for rows > DC_CHUNK) -> dc_finalize_kernel
//            leaves rec[p] = (cp, cn, code[r] | code[sigma r] << 16, code[col p]) for CSR entry p of row r.
// Backward : dc_grad_rows_kernel (+ dc_grad_long_kernel per direction when it has lists > DC_CHUNK):
//   dz[n] = gp * ( sum_{p in row n} cp[p] W[kc p]  +  sum_{q in trow n} cp[tperm q] W[kr(tperm q)] )
//         + gn * ( sum_{p in row sigma^-1 n} cn[p] W[kc p]  +  sum_{q in trow n} cn[tperm q] W[ksr(tperm q)] )
// One 16-lane group (a DPP row) owns a row / node: entries are held one per lane and broadcast inside the group with
// row_newbcast (no LDS traffic for the shuffles), four rows per wave in flight.  No float atomics; every sum has a
// fixed order (entry order inside a list, lists in the order above, long lists in ascending 256-entry pieces).
#include <rocprim/device/device_radix_sort.hpp>
#include "ceeg_internal.h"

namespace {

constexpr float DC_EPS = 1e-15f;
constexpr int DC_CHUNK = 256;          // lists longer than this are cut at the GLOBAL positions k*DC_CHUNK
constexpr int DC_CW = 64;              // codebook columns per LDS slab: 16 lanes x float4
constexpr int DC_MAX_K = 576;          // K * DC_CW * 4 bytes <= 144 KiB of LDS
constexpr int DC_GRAD_THREADS = 1024;  // one workgroup per CU (the slab fills its LDS), 16 waves
constexpr int DC_MAX_GRAD_BLOCKS = 256;

struct DcGradArgs {
  const float* W;
  int64_t ldw;
  int K, D;
  const int32_t *rowptr, *t_rowptr, *t_perm;
  const uint4* rec;
  const float* gscale;
  float* dz;
  int64_t lddz, N, nnz;
  int64_t sig_b, sig_ainv, sig_mod;
  unsigned int* queue;
  float* part_csr;   // [chunk*2+slot][2][D]: pieces of long CSR rows       (sum 0: cp W[kc], sum 1: cn W[kc])
  float* part_t;     // [chunk*2+slot][2][D]: pieces of long transposed rows (sum 0: cp W[kr], sum 1: cn W[ksr])
  float* coef_sum;   // [N] or NULL: summed, scaled coefficients of every node's pairs (conditional decoders)
  float* cpart_csr;  // [chunk*2+slot][2]: (sum cp, sum cn) of the pieces of long CSR rows
  float* cpart_t;    // [chunk*2+slot][2]: the same for long transposed rows
  // PAIRS: a fourth list per node - the node-sorted incidence keys of an arbitrary pair list (ceeg_pair_bce_codes_fwd),
  // node n's keys at [p_rowptr[n], p_rowptr[n+1]); coefficient of key k = p_C[code[n] * K + (k & 1023)]; scaled by gscale[1]
  const int32_t* p_rowptr;
  const uint32_t* p_keys;
  const int32_t* p_code;
  const float* p_C;
};

struct DcNode {  // one pipeline stage of dc_grad_rows_kernel: a node's three lists and their first batches
  int A_lo, A_hi, a_hi, B_lo, B_hi, b_hi, C_lo, C_hi, c_hi, tp;
  uint4 ra, rb, rc;
  int P_lo, P_hi, kself;  // PAIRS: the node's incidence keys, its own code, the first two batches of keys and coefficients
  unsigned k0, k1;
  float c0, c1;
};

template <int T>
__device__ __forceinline__ int dc_bcast(int v) {  // lane T of the caller's 16-lane DPP row, in every lane of that row
  return __builtin_amdgcn_mov_dpp(v, 0x150 + T, 0xf, 0xf, false);  // every lane has a source: no `old` value to set up
}
template <int T>
__device__ __forceinline__ float dc_bcastf(float v) { return __int_as_float(dc_bcast<T>(__float_as_int(v))); }

#define DC_FOR16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

__device__ __forceinline__ int dc_clamp_code(int k, int K) { return k < 0 ? 0 : (k >= K ? K - 1 : k); }

__device__ __forceinline__ int64_t dc_sigma_inv(int64_t y, int64_t ainv, int64_t b, int64_t mod) {
  const uint64_t l = (uint64_t)y % (uint64_t)mod;
  const uint64_t t = (l + (uint64_t)mod - (uint64_t)b) % (uint64_t)mod;
  return y - (int64_t)l + (int64_t)(((uint64_t)ainv * t) % (uint64_t)mod);
}

__device__ __forceinline__ bool dc_sorted_has(const int32_t* __restrict__ col, int lo, int end, int32_t s) {
  int hi = end;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < s) lo = mid + 1; else hi = mid;
  }
  return lo < end && col[lo] == s;
}

// literal "+EPS" terms and their derivatives (as decoder.hip: dec_pos / dec_neg)
__device__ __forceinline__ void dc_terms(float vp, float vn, float& tp, float& cp, float& tn, float& cn) {
  const float sp = 1.0f / (1.0f + expf(-vp));
  tp = -logf(sp + DC_EPS);
  cp = -(sp * (1.0f - sp)) / (sp + DC_EPS);
  const float sn = 1.0f / (1.0f + expf(-vn));
  const float om = 1.0f - sn;
  tn = -logf(om + DC_EPS);
  cn = (sn * om) / (om + DC_EPS);
}

// ------------------------------------------------------------------------------------------------ Gram matrix
// G[a][b] = k-ascending fmaf chain from 0 of W[a][k] * W[b][k]  (fmaf(x, y, acc) is symmetric in x, y: G is exactly symmetric)
__global__ __launch_bounds__(256) void dc_gram_kernel(const float* __restrict__ W, int64_t ldw, int K, int D, float* __restrict__ G) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  const int a = blockIdx.y;
  if (b >= K) return;
  const float* wa = W + (int64_t)a * ldw;
  const float* wb = W + (int64_t)b * ldw;
  float acc = 0.f;
  int k = 0;
  for (; k + 32 <= D; k += 32) {  // 8 x 16 bytes of the lane's own row in flight (row a is one broadcast line); same chain
    float4 y[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) y[u] = *reinterpret_cast<const float4*>(wb + k + 4 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 x = *reinterpret_cast<const float4*>(wa + k + 4 * u);
      acc = __fmaf_rn(x.x, y[u].x, acc);
      acc = __fmaf_rn(x.y, y[u].y, acc);
      acc = __fmaf_rn(x.z, y[u].z, acc);
      acc = __fmaf_rn(x.w, y[u].w, acc);
    }
  }
  for (; k < D; k += 4) {
    const float4 x = *reinterpret_cast<const float4*>(wa + k);
    const float4 y = *reinterpret_cast<const float4*>(wb + k);
    acc = __fmaf_rn(x.x, y.x, acc);
    acc = __fmaf_rn(x.y, y.y, acc);
    acc = __fmaf_rn(x.z, y.z, acc);
    acc = __fmaf_rn(x.w, y.w, acc);
  }
  G[(size_t)a * K + b] = acc;
}

// codes as clamped 16-bit values: the array the pair kernels read at random shrinks to 2 bytes per node (2 MB at N = 1 M) and
// stays in L2 next to the 1 MB term table while the pair / key streams and the sampler's bitmap lines flow past
__global__ __launch_bounds__(256) void dc_code16_kernel(const int32_t* __restrict__ code, int64_t N, int K, unsigned short* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < N) out[i] = (unsigned short)dc_clamp_code(code[i], K);
}

// ------------------------------------------------------------------------------------------------ forward: coefficients
// Wave w owns rows [64 w, 64 w + 64): the row pointers, sigma-rows and codes of all 64 arrive in one lane-parallel
// load; group g walks rows 4 it + g, 16 entries at a time.  Static assignment: the loss partial of wave w goes to slot w,
// so the sums do not depend on scheduling.
template <bool NEG>  // NEG = false: positives only (the literal sampler path) - no sigma-row, no second table read, no negative terms
__global__ __launch_bounds__(256) void dc_coef_rows_kernel(const float* __restrict__ G, int K, const unsigned short* __restrict__ code,
                                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N,
                                                           int64_t sig_a, int64_t sig_b, int64_t sig_mod, const float* __restrict__ row_add,
                                                           uint4* __restrict__ rec, float* __restrict__ part) {
  const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row0 = w * 64;
  if (row0 >= N) return;
  const int64_t r = row0 + lane;
  int lo = 0, hi = 0, sr = 0, slo = 0, shi = 0, kr = 0, ksr = 0;
  float radd = 0.f;
  if (r < N) {
    if (row_add) radd = row_add[r];
    lo = rowptr[r];
    hi = rowptr[r + 1];
    kr = code[r];
    if (NEG) {
      sr = (int)ceeg_block_affine(r, sig_a, sig_b, sig_mod);
      slo = rowptr[sr];
      shi = rowptr[sr + 1];
      ksr = code[sr];
    }
    if (hi - lo > DC_CHUNK) hi = lo;  // hub row: dc_coef_long_kernel
  }
  const int cnt = N - row0 < 64 ? (int)(N - row0) : 64;
  float a_pos = 0.f, a_neg = 0.f, a_cnt = 0.f;
  for (int it = 0; it * 4 < cnt; ++it) {
    const int j = it * 4 + grp;
    const int p_lo = __shfl(lo, j, 64), p_hi = __shfl(hi, j, 64);
    const int g_kr = __shfl(kr, j, 64);
    const float g_add = __shfl(radd, j, 64);
    const float* gr = G + (size_t)g_kr * K;
    if (!NEG) {
      for (int base = p_lo; __any(base < p_hi); base += 16) {
        const int p = base + gl;
        if (p < p_hi) {
          const int kc = code[col[p]];
          float tp, cp, tn, cn;
          const float v = gr[kc] + g_add;
          dc_terms(v, v, tp, cp, tn, cn);  // the negative half is dead code here
          rec[p] = make_uint4(__float_as_uint(cp), 0u, (unsigned)g_kr, (unsigned)kc);
          a_pos += tp;
        }
      }
      continue;
    }
    const int g_sr = __shfl(sr, j, 64), g_slo = __shfl(slo, j, 64), g_shi = __shfl(shi, j, 64);
    const int g_ksr = __shfl(ksr, j, 64);
    const int s_n = g_shi - g_slo;
    const bool s_short = s_n <= 16;
    int sc = -1;  // sigma-row's columns across the group's lanes (never equal to a column id when absent)
    if (p_lo < p_hi && s_short && gl < s_n) sc = col[g_slo + gl];
    const float* gs = G + (size_t)g_ksr * K;
    for (int base = p_lo; __any(base < p_hi); base += 16) {
      const int p = base + gl;
      const bool live = p < p_hi;
      int c = 0, kc = 0;
      if (live) {
        c = col[p];
        kc = code[c];
      }
      bool hit = false;
#define DC_HIT(T) hit |= (c == dc_bcast<T>(sc));
      DC_FOR16(DC_HIT)
#undef DC_HIT
      if (live && !s_short) hit = dc_sorted_has(col, g_slo, g_shi, c);
      if (live) {
        float tp, cp, tn, cn;
        dc_terms(gr[kc] + g_add, gs[kc] + g_add, tp, cp, tn, cn);
        const bool valid = c != g_sr && !hit;
        if (!valid) { tn = 0.f; cn = 0.f; }
        rec[p] = make_uint4(__float_as_uint(cp), __float_as_uint(cn), (unsigned)g_kr | ((unsigned)g_ksr << 16), (unsigned)kc);
        a_pos += tp;
        a_neg += tn;
        a_cnt += valid ? 1.0f : 0.f;
      }
    }
  }
  const float t0 = ceeg_wave_sum(a_pos), t1 = ceeg_wave_sum(a_neg), t2 = ceeg_wave_sum(a_cnt);
  if (lane == 0) {
    float* o = part + (size_t)w * 3;
    o[0] = t0; o[1] = t1; o[2] = t2;
  }
}

// pieces of hub rows: one wave per item (lo, hi, chunk*2+slot, row), 64 entries at a time; loss partial -> slot0 + item
__global__ __launch_bounds__(256) void dc_coef_long_kernel(const float* __restrict__ G, int K, const unsigned short* __restrict__ code,
                                                           const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const int4* __restrict__ items, int64_t n_items, int64_t sig_a, int64_t sig_b,
                                                           int64_t sig_mod, const float* __restrict__ row_add, uint4* __restrict__ rec,
                                                           float* __restrict__ part, int64_t slot0, int no_neg) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_items) return;
  const int4 itv = items[i];
  const int lo = itv.x, hi = itv.y, r = itv.w;
  const int sr = (int)ceeg_block_affine((int64_t)r, sig_a, sig_b, sig_mod);
  const int slo = rowptr[sr], shi = rowptr[sr + 1];
  const int kr = code[r], ksr = code[sr];
  const float* gr = G + (size_t)kr * K;
  const float* gs = G + (size_t)ksr * K;
  const float radd = row_add ? row_add[r] : 0.f;
  float a_pos = 0.f, a_neg = 0.f, a_cnt = 0.f;
  for (int base = lo; base < hi; base += 64) {
    const int p = base + lane;
    if (p < hi) {
      const int c = col[p];
      const int kc = code[c];
      float tp, cp, tn, cn;
      dc_terms(gr[kc] + radd, gs[kc] + radd, tp, cp, tn, cn);
      const bool valid = !no_neg && c != sr && !dc_sorted_has(col, slo, shi, c);
      if (!valid) { tn = 0.f; cn = 0.f; }
      rec[p] = make_uint4(__float_as_uint(cp), __float_as_uint(cn), (unsigned)kr | ((unsigned)ksr << 16), (unsigned)kc);
      a_pos += tp;
      a_neg += tn;
      a_cnt += valid ? 1.0f : 0.f;
    }
  }
  const float t0 = ceeg_wave_sum(a_pos), t1 = ceeg_wave_sum(a_neg), t2 = ceeg_wave_sum(a_cnt);
  if (lane == 0) {
    float* o = part + (size_t)(slot0 + i) * 3;
    o[0] = t0; o[1] = t1; o[2] = t2;
  }
}

__global__ __launch_bounds__(1024) void dc_finalize_kernel(const float* __restrict__ part, int64_t nslot, float* __restrict__ out) {
  __shared__ double red[3][1024];
  double s[3] = {0.0, 0.0, 0.0};
  for (int64_t i = threadIdx.x; i < nslot; i += 1024) {
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] += (double)part[(size_t)i * 3 + k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
#pragma unroll
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) out[threadIdx.x] = (float)red[threadIdx.x][0];
}

// ------------------------------------------------------------------------------------------------ backward: gradient
__device__ __forceinline__ void dc_fma4(float c, const float4& x, float4& s) {
  s.x = __fmaf_rn(c, x.x, s.x);
  s.y = __fmaf_rn(c, x.y, s.y);
  s.z = __fmaf_rn(c, x.z, s.z);
  s.w = __fmaf_rn(c, x.w, s.w);
}

// one batch: the group's lanes hold (cf, k) of up to 16 entries (n live; the others carry cf = 0, k = 0, which add 0 * W[0]):
// s += sum_t cf_t W[k_t].  Steps run in blocks of 4 with one wave-uniform early-out per block, so the four LDS reads of a
// block are in flight together (a branch per step serialised them: read -> wait -> fma).
#define DC_STEP1(T)                             \
    {                                           \
      const float ct = dc_bcastf<T>(cf);        \
      const int kt = dc_bcast<T>(k);            \
      dc_fma4(ct, lds4[kt * 16 + gl], s);       \
    }
__device__ __forceinline__ void dc_batch1(const float4* lds4, int gl, float cf, int k, int n, float4& s) {
  do {
    if (!__any(n > 0)) break;
    DC_STEP1(0) DC_STEP1(1) DC_STEP1(2) DC_STEP1(3)
    if (!__any(n > 4)) break;
    DC_STEP1(4) DC_STEP1(5) DC_STEP1(6) DC_STEP1(7)
    if (!__any(n > 8)) break;
    DC_STEP1(8) DC_STEP1(9) DC_STEP1(10) DC_STEP1(11)
    if (!__any(n > 12)) break;
    DC_STEP1(12) DC_STEP1(13) DC_STEP1(14) DC_STEP1(15)
  } while (0);
}
#undef DC_STEP1
// the same with two (coefficient, code) pairs per entry and two sums
#define DC_STEP2(T)                                       \
    {                                                     \
      const float at = dc_bcastf<T>(ca);                  \
      const int kat = dc_bcast<T>(ka);                    \
      const float bt = dc_bcastf<T>(cb);                  \
      const int kbt = dc_bcast<T>(kb);                    \
      dc_fma4(at, lds4[kat * 16 + gl], sa);               \
      dc_fma4(bt, lds4[kbt * 16 + gl], sb);               \
    }
__device__ __forceinline__ void dc_batch2(const float4* lds4, int gl, float ca, int ka, float cb, int kb, int n, float4& sa, float4& sb) {
  do {
    if (!__any(n > 0)) break;
    DC_STEP2(0) DC_STEP2(1) DC_STEP2(2) DC_STEP2(3)
    if (!__any(n > 4)) break;
    DC_STEP2(4) DC_STEP2(5) DC_STEP2(6) DC_STEP2(7)
    if (!__any(n > 8)) break;
    DC_STEP2(8) DC_STEP2(9) DC_STEP2(10) DC_STEP2(11)
    if (!__any(n > 12)) break;
    DC_STEP2(12) DC_STEP2(13) DC_STEP2(14) DC_STEP2(15)
  } while (0);
}
#undef DC_STEP2

__device__ __forceinline__ void dc_fill_slab(const float* __restrict__ W, int64_t ldw, int K, int slab, float4* lds4) {
  for (int i = threadIdx.x; i < K * 16; i += DC_GRAD_THREADS)
    lds4[i] = *reinterpret_cast<const float4*>(W + (int64_t)(i >> 4) * ldw + slab * DC_CW + 4 * (i & 15));
  __syncthreads();
}

__device__ __forceinline__ void dc_add_pieces(const float* __restrict__ part, int lo, int hi, int sum, int D, int col0, float4& s) {
  const int c_lo = lo / DC_CHUNK, c_hi = (hi - 1) / DC_CHUNK;
  for (int c = c_lo; c <= c_hi; ++c) {  // ascending position order
    const int slot = (c == c_lo && lo != c * DC_CHUNK) ? 1 : 0;
    const float4 t = *reinterpret_cast<const float4*>(part + (((size_t)c * 2 + slot) * 2 + sum) * D + col0);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
}

// Coefficient sums (conditional decoders): S[n] = gp * (sum_{p in row n} cp + sum_{q in trow n} cp[tperm q])
//                                                  + gn * (sum_{p in row sigma^-1 n} cn + sum_{q in trow n} cn[tperm q])
// is the factor of every decoder-input column that is CONSTANT over the pairs of a node (a per-graph condition appended to the
// latent).  dc_grad_rows_kernel<., true> adds the coefficients it walks anyway (lane l of a node's group adds entries l, l+16,
// ... in list order, then a fixed butterfly); the pieces of long lists are summed beforehand by dc_coef_sum_long_kernel, one wave
// per item of the graph's static piece lists, four loads in flight per lane.
template <bool TRANSPOSED>
__global__ __launch_bounds__(256) void dc_coef_sum_long_kernel(DcGradArgs a, const int4* __restrict__ items, int64_t n_items) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_items) return;
  const int4 itv = items[i];
  float sp = 0.f, sn = 0.f;
  for (int p = itv.x + lane; p < itv.y; p += 256) {
    float x[4], y[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pu = p + 64 * u;
      x[u] = 0.f; y[u] = 0.f;
      if (pu < itv.y) {
        const uint2 r = *reinterpret_cast<const uint2*>(a.rec + (TRANSPOSED ? a.t_perm[pu] : pu));
        x[u] = __uint_as_float(r.x);
        y[u] = __uint_as_float(r.y);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { sp += x[u]; sn += y[u]; }
  }
  sp = ceeg_wave_sum(sp);
  sn = ceeg_wave_sum(sn);
  if (lane == 0) {
    float* o = (TRANSPOSED ? a.cpart_t : a.cpart_csr) + (size_t)itv.z * 2;
    o[0] = sp;
    o[1] = sn;
  }
}
__device__ __forceinline__ float dc_piece_sums(const float* __restrict__ part, int lo, int hi, int which) {
  const int c_lo = lo / DC_CHUNK, c_hi = (hi - 1) / DC_CHUNK;
  float s = 0.f;
  for (int c = c_lo; c <= c_hi; ++c) {  // ascending position order
    const int slot = (c == c_lo && lo != c * DC_CHUNK) ? 1 : 0;
    s += part[((size_t)c * 2 + slot) * 2 + which];
  }
  return s;
}

// Persistent grid, one workgroup per CU; workgroup b sweeps slab b % nslab.  Every wave draws 64 consecutive nodes from its
// slab's counter (integer atomics only: which wave computes a node does not change the node's result).
// NEG = false: positives only (every cn is 0): the sigma-row list and the cn halves of the transposed list are not walked.
// CSUM: the workgroups of slab 0 also write a.coef_sum.
// PAIRS (with NEG = false): the sampled-pair list of the literal recon_loss is a fourth list of every node - its sum lands in
// the accumulator the structured negatives would use and is scaled by gscale[1] - so both pair sets leave ONE dz write.
template <bool NEG, bool CSUM, bool PAIRS = false>
__global__ __launch_bounds__(DC_GRAD_THREADS) void dc_grad_rows_kernel(DcGradArgs a) {
  static_assert(!(NEG && PAIRS), "the pair list takes the structured negatives' accumulator");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* lds4 = reinterpret_cast<float4*>(smem);
  const int nslab = a.D / DC_CW;
  const int slab = blockIdx.x % nslab;
  dc_fill_slab(a.W, a.ldw, a.K, slab, lds4);
  const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
  const int col0 = slab * DC_CW + 4 * gl;
  const float gp = a.gscale[0], gn = a.gscale[1];
  unsigned int* q = a.queue + slab * 32;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  for (;;) {
    unsigned int dv = 0;
    if (lane == 0) dv = atomicAdd(q, 64u);
    const int64_t base_n = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)dv);
    if (base_n >= a.N) break;
    const int64_t n = base_n + lane;
    int alo = 0, ahi = 0, blo = 0, bhi = 0, clo = 0, chi = 0;
    if (n < a.N) {
      alo = a.rowptr[n];
      ahi = a.rowptr[n + 1];
      if (NEG) {
        const int64_t m = dc_sigma_inv(n, a.sig_ainv, a.sig_b, a.sig_mod);
        blo = a.rowptr[m];
        bhi = a.rowptr[m + 1];
      }
      clo = a.t_rowptr[n];
      chi = a.t_rowptr[n + 1];
    }
    int plo = 0, phi = 0, pks = 0;
    if (PAIRS && n < a.N) {
      plo = a.p_rowptr[n];
      phi = a.p_rowptr[n + 1];
      pks = dc_clamp_code(a.p_code[n], a.K);
    }
    const int cnt = a.N - base_n < 64 ? (int)(a.N - base_n) : 64;
    // Three-stage pipeline over the draw's iterations (4 nodes each): while iteration `it` computes, the records of
    // it+1 are in flight (their t_perm entries were fetched an iteration earlier) and the t_perm entries of it+2 are issued.
    DcNode s0, s1, s2;
    auto bounds = [&](int it, DcNode& s) {
      const int j = (it * 4 + grp) & 63;
      const bool ok = it * 4 < cnt;  // wave-uniform
      s.A_lo = __shfl(alo, j, 64); s.A_hi = __shfl(ahi, j, 64);
      s.B_lo = __shfl(blo, j, 64); s.B_hi = __shfl(bhi, j, 64);
      s.C_lo = __shfl(clo, j, 64); s.C_hi = __shfl(chi, j, 64);
      if (!ok) { s.A_hi = s.A_lo; s.B_hi = s.B_lo; s.C_hi = s.C_lo; }
      s.a_hi = s.A_hi - s.A_lo > DC_CHUNK ? s.A_lo : s.A_hi;  // long lists: their pieces come from dc_grad_long_kernel
      s.b_hi = s.B_hi - s.B_lo > DC_CHUNK ? s.B_lo : s.B_hi;
      s.c_hi = s.C_hi - s.C_lo > DC_CHUNK ? s.C_lo : s.C_hi;
      s.tp = 0;
      if (s.C_lo + gl < s.c_hi) s.tp = a.t_perm[s.C_lo + gl];
      if (PAIRS) {
        s.P_lo = __shfl(plo, j, 64);
        s.P_hi = __shfl(phi, j, 64);
        s.kself = __shfl(pks, j, 64);
        if (!ok) s.P_hi = s.P_lo;
        s.k0 = 0u; s.k1 = 0u;
        if (s.P_lo + gl < s.P_hi) s.k0 = a.p_keys[s.P_lo + gl];
        if (s.P_lo + 16 + gl < s.P_hi) s.k1 = a.p_keys[s.P_lo + 16 + gl];
      }
    };
    auto records = [&](DcNode& s) {  // first batches of the three lists
      s.ra = zero4; s.rb = zero4; s.rc = zero4;
      if (s.C_lo + gl < s.c_hi) s.rc = a.rec[s.tp];
      if (s.A_lo + gl < s.a_hi) s.ra = a.rec[s.A_lo + gl];
      if (NEG && s.B_lo + gl < s.b_hi) s.rb = a.rec[s.B_lo + gl];
      if (PAIRS) {
        const float* row = a.p_C + (size_t)s.kself * a.K;
        s.c0 = 0.f; s.c1 = 0.f;
        if (s.P_lo + gl < s.P_hi) s.c0 = row[s.k0 & 1023u];
        if (s.P_lo + 16 + gl < s.P_hi) s.c1 = row[s.k1 & 1023u];
      }
    };
    bounds(0, s0);
    bounds(1, s1);
    records(s0);
    for (int it = 0; it * 4 < cnt; ++it) {
      bounds(it + 2, s2);
      records(s1);
      const int j = it * 4 + grp;
      uint4 ra = s0.ra, rb = s0.rb, rc = s0.rc;
      float4 s1v = make_float4(0.f, 0.f, 0.f, 0.f), s2v = s1v, s3v = s1v, s4v = s1v;
      float cs_p = 0.f, cs_n = 0.f;  // CSUM: this lane's share of the node's coefficient sums (absent entries carry 0)
      // (1) n as destination of positives: sum_{p in row n} cp W[kc]
      for (int base = s0.A_lo; __any(base < s0.a_hi); base += 16) {
        if (base != s0.A_lo) { ra = zero4; if (base + gl < s0.a_hi) ra = a.rec[base + gl]; }
        const int nb = s0.a_hi - base;
        if (CSUM) cs_p += __uint_as_float(ra.x);
        dc_batch1(lds4, gl, __uint_as_float(ra.x), (int)ra.w, nb < 0 ? 0 : nb, s1v);
      }
      // (3) n as negative endpoint sigma(m): sum_{p in row m} cn W[kc]
      for (int base = s0.B_lo; NEG && __any(base < s0.b_hi); base += 16) {
        if (base != s0.B_lo) { rb = zero4; if (base + gl < s0.b_hi) rb = a.rec[base + gl]; }
        const int nb = s0.b_hi - base;
        if (CSUM) cs_n += __uint_as_float(rb.y);
        dc_batch1(lds4, gl, __uint_as_float(rb.y), (int)rb.w, nb < 0 ? 0 : nb, s3v);
      }
      if (PAIRS) {  // the node's sampled-pair incidences, in sorted (= incidence) order: sum_k C[code n][code_k] W[code_k]
        const int len = s0.P_hi - s0.P_lo;
        dc_batch1(lds4, gl, s0.c0, (int)(s0.k0 & 1023u), len < 0 ? 0 : len, s3v);
        dc_batch1(lds4, gl, s0.c1, (int)(s0.k1 & 1023u), len - 16 < 0 ? 0 : len - 16, s3v);
        const float* row = a.p_C + (size_t)s0.kself * a.K;
        for (int base = s0.P_lo + 32; __any(base < s0.P_hi); base += 16) {  // lists longer than 32: fetched in place
          unsigned k = 0u;
          float c = 0.f;
          if (base + gl < s0.P_hi) {
            k = a.p_keys[base + gl];
            c = row[k & 1023u];
          }
          const int nb = s0.P_hi - base;
          dc_batch1(lds4, gl, c, (int)(k & 1023u), nb < 0 ? 0 : nb, s3v);
        }
      }
      // (2) + (4) n as source: sum_q cp W[kr], sum_q cn W[ksr] over the transposed row
      for (int base = s0.C_lo; __any(base < s0.c_hi); base += 16) {
        if (base != s0.C_lo) {
          rc = zero4;
          if (base + gl < s0.c_hi) rc = a.rec[a.t_perm[base + gl]];
        }
        const int nb = s0.c_hi - base;
        if (CSUM) { cs_p += __uint_as_float(rc.x); cs_n += __uint_as_float(rc.y); }
        if (NEG) dc_batch2(lds4, gl, __uint_as_float(rc.x), (int)(rc.z & 0xffffu), __uint_as_float(rc.y), (int)(rc.z >> 16), nb < 0 ? 0 : nb, s2v, s4v);
        else dc_batch1(lds4, gl, __uint_as_float(rc.x), (int)(rc.z & 0xffffu), nb < 0 ? 0 : nb, s2v);
      }
      if (s0.A_hi != s0.a_hi) dc_add_pieces(a.part_csr, s0.A_lo, s0.A_hi, 0, a.D, col0, s1v);
      if (NEG && s0.B_hi != s0.b_hi) dc_add_pieces(a.part_csr, s0.B_lo, s0.B_hi, 1, a.D, col0, s3v);
      if (s0.C_hi != s0.c_hi) {
        dc_add_pieces(a.part_t, s0.C_lo, s0.C_hi, 0, a.D, col0, s2v);
        if (NEG) dc_add_pieces(a.part_t, s0.C_lo, s0.C_hi, 1, a.D, col0, s4v);
      }
      if (CSUM && slab == 0) {  // block-uniform
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          cs_p += __shfl_xor(cs_p, o, 64);
          cs_n += __shfl_xor(cs_n, o, 64);
        }
        if (s0.A_hi != s0.a_hi) cs_p += dc_piece_sums(a.cpart_csr, s0.A_lo, s0.A_hi, 0);
        if (NEG && s0.B_hi != s0.b_hi) cs_n += dc_piece_sums(a.cpart_csr, s0.B_lo, s0.B_hi, 1);
        if (s0.C_hi != s0.c_hi) {
          cs_p += dc_piece_sums(a.cpart_t, s0.C_lo, s0.C_hi, 0);
          cs_n += dc_piece_sums(a.cpart_t, s0.C_lo, s0.C_hi, 1);
        }
        if (j < cnt && gl == 0) a.coef_sum[base_n + j] = gp * cs_p + gn * cs_n;
      }
      if (j < cnt) {
        float4 o;
        o.x = gn * (s3v.x + s4v.x) + gp * (s1v.x + s2v.x);
        o.y = gn * (s3v.y + s4v.y) + gp * (s1v.y + s2v.y);
        o.z = gn * (s3v.z + s4v.z) + gp * (s1v.z + s2v.z);
        o.w = gn * (s3v.w + s4v.w) + gp * (s1v.w + s2v.w);
        *reinterpret_cast<float4*>(a.dz + (base_n + j) * a.lddz + col0) = o;
      }
      s0 = s1;
      s1 = s2;
    }
  }
}

// pieces of long lists: one 16-lane group per item (lo, hi, chunk*2+slot, row); TRANSPOSED: positions index t_perm
template <bool TRANSPOSED>
__global__ __launch_bounds__(DC_GRAD_THREADS) void dc_grad_long_kernel(DcGradArgs a, const int4* __restrict__ items, int64_t n_items) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* lds4 = reinterpret_cast<float4*>(smem);
  const int nslab = a.D / DC_CW;
  const int slab = blockIdx.x % nslab;
  dc_fill_slab(a.W, a.ldw, a.K, slab, lds4);
  const int gl = threadIdx.x & 15;
  const int col0 = slab * DC_CW + 4 * gl;
  const int64_t stride = (int64_t)(gridDim.x / nslab) * (DC_GRAD_THREADS / 16);
  float* part = TRANSPOSED ? a.part_t : a.part_csr;
  for (int64_t i = (int64_t)(blockIdx.x / nslab) * (DC_GRAD_THREADS / 16) + (threadIdx.x >> 4); __any(i < n_items); i += stride) {
    int lo = 0, hi = 0, slot = 0;
    if (i < n_items) {
      const int4 itv = items[i];
      lo = itv.x; hi = itv.y; slot = itv.z;
    }
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    for (int base = lo; __any(base < hi); base += 16) {
      uint4 r = make_uint4(0u, 0u, 0u, 0u);
      if (base + gl < hi) r = a.rec[TRANSPOSED ? a.t_perm[base + gl] : base + gl];
      const int ka = TRANSPOSED ? (int)(r.z & 0xffffu) : (int)r.w;
      const int kb = TRANSPOSED ? (int)(r.z >> 16) : (int)r.w;
      const int nb = hi - base;
      dc_batch2(lds4, gl, __uint_as_float(r.x), ka, __uint_as_float(r.y), kb, nb < 0 ? 0 : nb, sa, sb);
    }
    if (hi > lo) {
      float* o = part + ((size_t)slot * 2) * a.D + col0;
      *reinterpret_cast<float4*>(o) = sa;
      *reinterpret_cast<float4*>(o + a.D) = sb;
    }
  }
}

bool dc_supported(int n_e, int D) { return n_e >= 1 && n_e <= DC_MAX_K && D >= DC_CW && D <= 256 && (D % DC_CW) == 0; }
size_t dc_cpart_bytes(int64_t nnz, int64_t n_items) {  // scalar (sum cp, sum cn) per piece slot
  if (n_items <= 0) return 0;
  const size_t nchunk = (size_t)((nnz + DC_CHUNK - 1) / DC_CHUNK);
  return ceeg_ws_part((nchunk + 1) * 2 * 2, 4);
}
size_t dc_part_bytes(int64_t nnz, int D, int64_t n_items) {
  if (n_items <= 0) return 0;
  const size_t nchunk = (size_t)((nnz + DC_CHUNK - 1) / DC_CHUNK);
  return ceeg_ws_part((nchunk + 1) * 2 * 2 * (size_t)D, 4);
}

}  // namespace

extern "C" int ceeg_decoder_codes_chunk(void) { return DC_CHUNK; }
extern "C" int ceeg_decoder_codes_supported(int n_e, int D) { return dc_supported(n_e, D) ? 1 : 0; }

extern "C" size_t ceeg_decoder_codes_ws_bytes(int64_t N, int64_t nnz, int n_e, int D, int64_t n_long, int64_t n_tlong) {
  if (N <= 0 || nnz < 0 || !dc_supported(n_e, D) || n_long < 0 || n_tlong < 0) return 0;
  const size_t fwd = ceeg_ws_part((size_t)n_e * n_e, 4) + ceeg_ws_part(((size_t)((N + 63) / 64) + (size_t)n_long) * 3, 4) + ceeg_ws_part((size_t)N, 2);
  const size_t bwd = 1024 + dc_part_bytes(nnz, D, n_long) + dc_part_bytes(nnz, D, n_tlong) + dc_cpart_bytes(nnz, n_long) + dc_cpart_bytes(nnz, n_tlong);
  return fwd > bwd ? fwd : bwd;
}

extern "C" int ceeg_decoder_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* rowptr,
                                      const int32_t* col, int64_t N, int64_t nnz, int64_t sig_a, int64_t sig_b, int64_t sig_mod,
                                      const int32_t* long_items, int64_t n_long, int64_t max_row_nnz, const float* row_add, void* rec,
                                      float* sums, void* ws, size_t ws_bytes, void* stream) {
  if (!W || !code || !rowptr || !sums || N <= 0 || nnz < 0 || (nnz > 0 && (!col || !rec))) return CEEG_EINVAL;
  if (!dc_supported(n_e, D) || N >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (ldw < D || (ldw % 4) != 0 || !ceeg_aligned(W, 16) || (rec && !ceeg_aligned(rec, 16))) return CEEG_EINVAL;
  const int no_neg = sig_a == 0;  // sig_a == 0: positives only (every cn = 0, sums[1] = sums[2] = 0); sigma is then the identity
  if (no_neg) { sig_a = 1; sig_b = 0; }
  if (sig_mod <= 0 || (N % sig_mod) != 0 || sig_a <= 0 || sig_a >= sig_mod + (sig_mod == 1) || sig_b < 0 || sig_b >= sig_mod) return CEEG_EINVAL;
  if (n_long < 0 || (n_long > 0 && (!long_items || !ceeg_aligned(long_items, 16)))) return CEEG_EINVAL;
  if (max_row_nnz < 0 || (max_row_nnz > DC_CHUNK && n_long == 0)) return CEEG_EINVAL;  // hub rows need their piece list
  if (ws_bytes < ceeg_decoder_codes_ws_bytes(N, nnz, n_e, D, n_long, 0) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* G = static_cast<float*>(ws);
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + ceeg_ws_part((size_t)n_e * n_e, 4));
  const int64_t nwave = (N + 63) / 64;
  const int64_t nslot = nwave + n_long;
  unsigned short* code16 = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(part) + ceeg_ws_part((size_t)nslot * 3, 4));
  hipLaunchKernelGGL(dc_code16_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, code, N, n_e, code16);  // 2 bytes per node: L2-resident
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(dc_gram_kernel, dim3((unsigned)((n_e + 255) / 256), (unsigned)n_e), dim3(256), 0, st, W, ldw, n_e, D, G);
  CEEG_CHECK_LAUNCH();
  uint4* r4 = static_cast<uint4*>(rec);
  if (no_neg)
    hipLaunchKernelGGL(dc_coef_rows_kernel<false>, dim3((unsigned)((nwave + 3) / 4)), dim3(256), 0, st, G, n_e, code16, rowptr, col, N, sig_a,
                       sig_b, sig_mod, row_add, r4, part);
  else
    hipLaunchKernelGGL(dc_coef_rows_kernel<true>, dim3((unsigned)((nwave + 3) / 4)), dim3(256), 0, st, G, n_e, code16, rowptr, col, N, sig_a,
                       sig_b, sig_mod, row_add, r4, part);
  CEEG_CHECK_LAUNCH();
  if (n_long > 0) {
    hipLaunchKernelGGL(dc_coef_long_kernel, dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, st, G, n_e, code16, rowptr, col,
                       reinterpret_cast<const int4*>(long_items), n_long, sig_a, sig_b, sig_mod, row_add, r4, part, nwave, no_neg);
    CEEG_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(dc_finalize_kernel, dim3(1), dim3(1024), 0, st, part, nslot, sums);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

// Edge-list form on a quantised latent: <z_i, z_j> = G[code i][code j] (two 4-byte code loads and one float of the L2-resident
// Gram matrix per pair instead of two D*4-byte rows).  One thread per pair, grid-stride; block partials in a fixed order.
namespace {
constexpr int DC_EDGE_BLOCKS = 2048;
__global__ __launch_bounds__(256) void dc_edge_fwd_kernel(const float* __restrict__ G, int K, const int32_t* __restrict__ code,
                                                          const int32_t* __restrict__ ei, const int32_t* __restrict__ ej, int64_t E, int negative,
                                                          float* __restrict__ coef, float* __restrict__ part) {
  __shared__ float red[2][4];
  float acc = 0.f, cnt = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    const int ni = ei[e], nj = ej[e];
    float c = 0.f;
    if (ni >= 0 && nj >= 0) {  // a pair with a negative id is a sample the sampler gave up on: skipped
      const float v = G[(size_t)dc_clamp_code(code[ni], K) * K + dc_clamp_code(code[nj], K)];
      float tp, cp, tn, cn;
      dc_terms(v, v, tp, cp, tn, cn);
      acc += negative ? tn : tp;
      c = negative ? cn : cp;
      cnt += 1.0f;
    }
    if (coef) coef[e] = c;
  }
  acc = ceeg_wave_sum(acc);
  cnt = ceeg_wave_sum(cnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = acc; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + (size_t)blockIdx.x * 3;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    o[2] = 0.f;
  }
}
}  // namespace

extern "C" size_t ceeg_edge_bce_codes_ws_bytes(int64_t E, int n_e) {
  if (E < 0 || n_e <= 0) return 0;
  return ceeg_ws_part((size_t)n_e * n_e, 4) + ceeg_ws_part((size_t)(DC_EDGE_BLOCKS + 1) * 3, 4);
}

extern "C" int ceeg_edge_bce_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* i, const int32_t* j,
                                       int64_t E, int negative, float* coef, float* loss_sum, void* ws, size_t ws_bytes, void* stream) {
  if (!W || !code || !loss_sum || E < 0 || n_e <= 0 || D <= 0 || (E > 0 && (!i || !j))) return CEEG_EINVAL;
  if (n_e > 4096 || (D % 4) != 0 || (ldw % 4) != 0 || ldw < D) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(W, 16)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_edge_bce_codes_ws_bytes(E, n_e) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* G = static_cast<float*>(ws);
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + ceeg_ws_part((size_t)n_e * n_e, 4));
  hipLaunchKernelGGL(dc_gram_kernel, dim3((unsigned)((n_e + 255) / 256), (unsigned)n_e), dim3(256), 0, st, W, ldw, n_e, D, G);
  CEEG_CHECK_LAUNCH();
  int64_t nb = (E + 255) / 256;
  nb = nb < 1 ? 1 : (nb > DC_EDGE_BLOCKS ? DC_EDGE_BLOCKS : nb);
  hipLaunchKernelGGL(dc_edge_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, G, n_e, code, i, j, E, negative & 1, coef, part);
  CEEG_CHECK_LAUNCH();
  // dc_finalize_kernel sums 3 columns; column 2 is zero here: out[0] = sum of terms, out[1] = pairs counted
  float* tmp = part + (size_t)DC_EDGE_BLOCKS * 3;  // 3 floats of scratch behind the partials
  hipLaunchKernelGGL(dc_finalize_kernel, dim3(1), dim3(1024), 0, st, part, nb, tmp);
  CEEG_CHECK_LAUNCH();
  if (hipMemcpyAsync(loss_sum, tmp, 2 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return CEEG_ELAUNCH;
  return CEEG_OK;
}

namespace {
struct DcPairList {  // node-sorted incidence keys of a pair list (NULL p_rowptr: none)
  const int32_t* p_rowptr = nullptr;
  const uint32_t* p_keys = nullptr;
  const int32_t* p_code = nullptr;
  const float* p_C = nullptr;
};
int dc_decoder_bwd_impl(const float* W, int64_t ldw, int n_e, int D, const int32_t* rowptr, const int32_t* t_rowptr,
                        const int32_t* t_perm, int64_t N, int64_t nnz, int64_t sig_a, int64_t sig_b, int64_t sig_ainv,
                        int64_t sig_mod, const int32_t* long_items, int64_t n_long, const int32_t* t_long_items,
                        int64_t n_tlong, int64_t max_row_nnz, int64_t max_trow_nnz, const void* rec, const float* gscale,
                        float* dz, int64_t lddz, float* coef_sum, void* ws, size_t ws_bytes, void* stream, const DcPairList& pl);
}  // namespace

extern "C" int ceeg_decoder_codes_bwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* rowptr, const int32_t* t_rowptr,
                                      const int32_t* t_perm, int64_t N, int64_t nnz, int64_t sig_a, int64_t sig_b, int64_t sig_ainv,
                                      int64_t sig_mod, const int32_t* long_items, int64_t n_long, const int32_t* t_long_items,
                                      int64_t n_tlong, int64_t max_row_nnz, int64_t max_trow_nnz, const void* rec, const float* gscale,
                                      float* dz, int64_t lddz, float* coef_sum, void* ws, size_t ws_bytes, void* stream) {
  return dc_decoder_bwd_impl(W, ldw, n_e, D, rowptr, t_rowptr, t_perm, N, nnz, sig_a, sig_b, sig_ainv, sig_mod, long_items, n_long,
                             t_long_items, n_tlong, max_row_nnz, max_trow_nnz, rec, gscale, dz, lddz, coef_sum, ws, ws_bytes, stream,
                             DcPairList());
}

namespace {
int dc_decoder_bwd_impl(const float* W, int64_t ldw, int n_e, int D, const int32_t* rowptr, const int32_t* t_rowptr,
                        const int32_t* t_perm, int64_t N, int64_t nnz, int64_t sig_a, int64_t sig_b, int64_t sig_ainv,
                        int64_t sig_mod, const int32_t* long_items, int64_t n_long, const int32_t* t_long_items,
                        int64_t n_tlong, int64_t max_row_nnz, int64_t max_trow_nnz, const void* rec, const float* gscale,
                        float* dz, int64_t lddz, float* coef_sum, void* ws, size_t ws_bytes, void* stream, const DcPairList& pl) {
  if (!W || !rowptr || !t_rowptr || !gscale || !dz || N <= 0 || nnz < 0 || (nnz > 0 && (!t_perm || !rec))) return CEEG_EINVAL;
  if (!dc_supported(n_e, D) || N >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (ldw < D || (ldw % 4) != 0 || lddz < D || (lddz % 4) != 0 || !ceeg_aligned(W, 16) || !ceeg_aligned(dz, 16) || (rec && !ceeg_aligned(rec, 16)))
    return CEEG_EINVAL;
  const bool no_neg = sig_a == 0;  // positives only, as ceeg_decoder_codes_fwd: sigma is then the identity
  if (pl.p_rowptr && (!no_neg || coef_sum || !pl.p_code || !pl.p_C)) return CEEG_EINVAL;  // a pair list rides with the positives-only form
  if (no_neg) { sig_a = 1; sig_b = 0; sig_ainv = 1; }
  if (sig_mod <= 0 || (N % sig_mod) != 0 || sig_a <= 0 || sig_b < 0 || sig_b >= sig_mod || sig_ainv <= 0) return CEEG_EINVAL;
  if ((uint64_t)((unsigned __int128)(uint64_t)sig_a * (uint64_t)sig_ainv % (uint64_t)sig_mod) != (uint64_t)(1 % sig_mod)) return CEEG_EINVAL;
  if (n_long < 0 || n_tlong < 0 || (n_long > 0 && (!long_items || !ceeg_aligned(long_items, 16))) ||
      (n_tlong > 0 && (!t_long_items || !ceeg_aligned(t_long_items, 16))))
    return CEEG_EINVAL;
  if (max_row_nnz < 0 || max_trow_nnz < 0 || (max_row_nnz > DC_CHUNK && n_long == 0) || (max_trow_nnz > DC_CHUNK && n_tlong == 0)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_decoder_codes_ws_bytes(N, nnz, n_e, D, n_long, n_tlong) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  DcGradArgs a;
  a.W = W; a.ldw = ldw; a.K = n_e; a.D = D;
  a.rowptr = rowptr; a.t_rowptr = t_rowptr; a.t_perm = t_perm;
  a.rec = static_cast<const uint4*>(rec);
  a.gscale = gscale; a.dz = dz; a.lddz = lddz; a.N = N; a.nnz = nnz;
  a.sig_b = sig_b; a.sig_ainv = sig_ainv; a.sig_mod = sig_mod;
  a.queue = static_cast<unsigned int*>(ws);
  a.part_csr = reinterpret_cast<float*>(static_cast<char*>(ws) + 1024);
  a.part_t = reinterpret_cast<float*>(static_cast<char*>(ws) + 1024 + dc_part_bytes(nnz, D, n_long));
  a.cpart_csr = reinterpret_cast<float*>(reinterpret_cast<char*>(a.part_t) + dc_part_bytes(nnz, D, n_tlong));
  a.cpart_t = reinterpret_cast<float*>(reinterpret_cast<char*>(a.cpart_csr) + dc_cpart_bytes(nnz, n_long));
  a.coef_sum = coef_sum;
  a.p_rowptr = pl.p_rowptr; a.p_keys = pl.p_keys; a.p_code = pl.p_code; a.p_C = pl.p_C;
  if (hipMemsetAsync(ws, 0, 1024, st) != hipSuccess) return CEEG_ELAUNCH;
  const int nslab = D / DC_CW;
  const size_t lds = (size_t)n_e * DC_CW * 4;
  auto grid_for = [&](int64_t groups_of_work) {  // workgroups: a multiple of nslab, at most one per CU
    int64_t per_slab = (groups_of_work + (DC_GRAD_THREADS / 16) - 1) / (DC_GRAD_THREADS / 16);
    if (per_slab < 1) per_slab = 1;
    const int64_t cap = DC_MAX_GRAD_BLOCKS / nslab;
    if (per_slab > cap) per_slab = cap;
    return (unsigned)(per_slab * nslab);
  };
  if (n_long > 0) {
    hipLaunchKernelGGL(dc_grad_long_kernel<false>, dim3(grid_for(n_long)), dim3(DC_GRAD_THREADS), lds, st, a,
                       reinterpret_cast<const int4*>(long_items), n_long);
    CEEG_CHECK_LAUNCH();
  }
  if (n_tlong > 0) {
    hipLaunchKernelGGL(dc_grad_long_kernel<true>, dim3(grid_for(n_tlong)), dim3(DC_GRAD_THREADS), lds, st, a,
                       reinterpret_cast<const int4*>(t_long_items), n_tlong);
    CEEG_CHECK_LAUNCH();
  }
  if (coef_sum) {  // piece sums of the long lists first: dc_grad_rows_kernel adds them
    if (n_long > 0) {
      hipLaunchKernelGGL(dc_coef_sum_long_kernel<false>, dim3((unsigned)((n_long + 3) / 4)), dim3(256), 0, st, a,
                         reinterpret_cast<const int4*>(long_items), n_long);
      CEEG_CHECK_LAUNCH();
    }
    if (n_tlong > 0) {
      hipLaunchKernelGGL(dc_coef_sum_long_kernel<true>, dim3((unsigned)((n_tlong + 3) / 4)), dim3(256), 0, st, a,
                         reinterpret_cast<const int4*>(t_long_items), n_tlong);
      CEEG_CHECK_LAUNCH();
    }
  }
  const dim3 grid(grid_for(N)), block(DC_GRAD_THREADS);
  if (pl.p_rowptr) hipLaunchKernelGGL((dc_grad_rows_kernel<false, false, true>), grid, block, lds, st, a);
  else if (no_neg && coef_sum) hipLaunchKernelGGL((dc_grad_rows_kernel<false, true>), grid, block, lds, st, a);
  else if (no_neg) hipLaunchKernelGGL((dc_grad_rows_kernel<false, false>), grid, block, lds, st, a);
  else if (coef_sum) hipLaunchKernelGGL((dc_grad_rows_kernel<true, true>), grid, block, lds, st, a);
  else hipLaunchKernelGGL((dc_grad_rows_kernel<true, false>), grid, block, lds, st, a);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}
}  // namespace

// ================================================================================================ pair lists on a quantised latent
// recon_loss over an ARBITRARY pair list (the negatives `negative_sampling` draws afresh every step) without building a CSR pair.
// On a quantised latent a pair's term and its coefficient c = d term / dv depend only on the two CODES: v = G[code i][code j].
//   forward : dc_pair_tables_kernel tabulates term and coefficient over the K x K code pairs (G is exactly symmetric, so are
//             the tables); per pair k = (i, j) the loss adds T[code i][code j] and two 32-bit INCIDENCE keys are written,
//             key[2k] = j << 10 | code i and key[2k+1] = i << 10 | code j: node `key >> 10` receives c * W[key & 1023] with
//             c = C[code node][key & 1023]  (codes < 2^10, node <= N < 2^22);
//   backward: ONE stable keys-only radix sort of the 2E keys on their node bits (order inside a node = incidence order: a fixed
//             summation order, no float atomics), a boundary scan gives the row pointers, and dc_pair_grad_kernel sums
//             c * W[code] from the LDS codebook slab, the coefficients looked up in row code[n] of C: dz[n] (+)= gscale * sum.
// Pairs with an id outside [0, N) are skipped (not counted, node field N: sorted behind every node).
namespace {

constexpr int DC_PAIR_NODE_SHIFT = 10;
constexpr int64_t DC_PAIR_MAX_N = ((int64_t)1 << (32 - DC_PAIR_NODE_SHIFT)) - 1;  // the node field also holds the value N

unsigned dc_key_bits(int64_t N) {  // bits that hold the value N itself (the node field of skipped pairs)
  unsigned b = 1;
  while (((int64_t)1 << b) <= N) ++b;
  return b;
}
bool dc_pair_supported(int n_e, int D, int64_t N, int64_t E) {
  return dc_supported(n_e, D) && n_e <= (1 << DC_PAIR_NODE_SHIFT) && N > 0 && N <= DC_PAIR_MAX_N && E >= 0 && E < ((int64_t)1 << 30);
}
size_t dc_pair_sort_temp(size_t M, int64_t N) {
  size_t bytes = 0;
  uint32_t* k = nullptr;
  if (rocprim::radix_sort_keys(nullptr, bytes, k, k, M, (unsigned)DC_PAIR_NODE_SHIFT, (unsigned)DC_PAIR_NODE_SHIFT + dc_key_bits(N), (hipStream_t)0,
                               false) != hipSuccess)
    return 0;
  return bytes < 256 ? 256 : bytes;
}

// T[a][b] = term, C[a][b] = d term / dv at v = G[a][b]
__global__ __launch_bounds__(256) void dc_pair_tables_kernel(const float* __restrict__ G, int K, int negative, float* __restrict__ T,
                                                             float* __restrict__ C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K * K) return;
  const float v = G[i];
  float tp, cp, tn, cn;
  dc_terms(v, v, tp, cp, tn, cn);
  T[i] = negative ? tn : tp;
  C[i] = negative ? cn : cp;
}

__global__ __launch_bounds__(256) void dc_pair_fwd_kernel(const float* __restrict__ T, int K, const unsigned short* __restrict__ code,
                                                          const int32_t* __restrict__ ei, const int32_t* __restrict__ ej, int64_t E, int64_t N,
                                                          uint2* __restrict__ keys, float* __restrict__ part) {
  __shared__ float red[2][4];
  float acc = 0.f, cnt = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (int64_t)gridDim.x * 256) {
    const int ni = __builtin_nontemporal_load(ei + e), nj = __builtin_nontemporal_load(ej + e);  // streams: leave L2 to code[] and T
    unsigned key_i = (unsigned)N, key_j = (unsigned)N, ki = 0, kj = 0;
    if (ni >= 0 && nj >= 0 && ni < N && nj < N) {
      ki = code[ni];
      kj = code[nj];
      acc += T[(size_t)ki * K + kj];
      cnt += 1.0f;
      key_i = (unsigned)ni;
      key_j = (unsigned)nj;
    }
    if (keys) {
      __builtin_nontemporal_store((key_j << DC_PAIR_NODE_SHIFT) | ki, &keys[e].x);
      __builtin_nontemporal_store((key_i << DC_PAIR_NODE_SHIFT) | kj, &keys[e].y);
    }
  }
  acc = ceeg_wave_sum(acc);
  cnt = ceeg_wave_sum(cnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = acc; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + (size_t)blockIdx.x * 3;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    o[2] = 0.f;
  }
}

// negative_sampling fused into the pair-list forward: thread e DRAWS pair e (ceeg_neg_draw: the stand-alone sampler's samples bit
// for bit), writes it out (the caller's neg_edge_index) and goes on as dc_pair_fwd_kernel does - same grid, same per-thread
// strides, same partial layout, so loss and keys equal the two-launch form exactly.  The 128 MB pair list is never re-read.
template <bool FILTER>
__global__ __launch_bounds__(256) void dc_sample_pair_fwd_kernel(const float* __restrict__ T, int K, const unsigned short* __restrict__ code,
                                                                 const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                 int64_t N, int64_t E, uint64_t seed, const uint32_t* __restrict__ filter,
                                                                 unsigned filter_bits, int32_t* __restrict__ neg_src,
                                                                 int32_t* __restrict__ neg_dst, uint2* __restrict__ keys,
                                                                 float* __restrict__ part) {
  constexpr int U = 4;  // samples a thread draws in lockstep (its next U grid-stride iterations): U random reads in flight
  __shared__ float red[2][4];
  float acc = 0.f, cnt = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < E; e0 += U * stride) {
    int64_t ek[U];
    bool live[U];
    int32_t ni[U], nj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { ek[u] = e0 + u * stride; live[u] = ek[u] < E; }
    ceeg_neg_draw_multi<FILTER, U>(rowptr, col, N, seed, ek, live, filter, filter_bits, ni, nj);
    unsigned ki[U], kj[U];
    float t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ki[u] = 0; kj[u] = 0;
      if (live[u] && ni[u] >= 0) { ki[u] = (unsigned)code[ni[u]]; kj[u] = (unsigned)code[nj[u]]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      t[u] = 0.f;
      if (live[u] && ni[u] >= 0) t[u] = T[(size_t)ki[u] * K + kj[u]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {  // sums in ascending-e order, as the one-sample-at-a-time loop adds them
      if (!live[u]) continue;
      unsigned key_i = (unsigned)N, key_j = (unsigned)N;
      if (ni[u] >= 0) {
        acc += t[u];
        cnt += 1.0f;
        key_i = (unsigned)ni[u];
        key_j = (unsigned)nj[u];
      }
      __builtin_nontemporal_store(ni[u], neg_src + ek[u]);
      __builtin_nontemporal_store(nj[u], neg_dst + ek[u]);
      if (keys) {
        __builtin_nontemporal_store((key_j << DC_PAIR_NODE_SHIFT) | ki[u], &keys[ek[u]].x);
        __builtin_nontemporal_store((key_i << DC_PAIR_NODE_SHIFT) | kj[u], &keys[ek[u]].y);
      }
    }
  }
  acc = ceeg_wave_sum(acc);
  cnt = ceeg_wave_sum(cnt);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = acc; red[1][wave] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + (size_t)blockIdx.x * 3;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    o[2] = 0.f;
  }
}

// sorted keys -> row pointers: position p closes every row between node[p-1] and node[p] (node fields >= N count as N); a thread
// takes four consecutive positions from one 16-byte load (the keys array is the 256-byte aligned sort output)
__global__ __launch_bounds__(256) void dc_pair_rowptr_kernel(const uint32_t* __restrict__ keys, int64_t M, int64_t N, int32_t* __restrict__ rowptr) {
  const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 > M) return;
  auto node = [&](uint32_t k) -> int64_t { const int64_t n = (int64_t)(k >> DC_PAIR_NODE_SHIFT); return n < N ? n : N; };
  uint32_t k[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
  if (p0 + 4 <= M) {
    const uint4 u = *reinterpret_cast<const uint4*>(keys + p0);
    k[0] = u.x; k[1] = u.y; k[2] = u.z; k[3] = u.w;
  } else {
    for (int i = 0; i < 4; ++i) if (p0 + i < M) k[i] = keys[p0 + i];
  }
  int64_t prev = p0 > 0 ? node(keys[p0 - 1]) : -1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t p = p0 + i;
    if (p > M) break;
    const int64_t cur = p < M ? node(k[i]) : N;
    for (int64_t r = prev + 1; r <= cur; ++r) rowptr[r] = (int32_t)p;
    prev = cur;
  }
}

struct DcPairGradArgs {
  const float* W;
  int64_t ldw;
  int K, D;
  const int32_t* rowptr;
  const uint32_t* keys;
  const int32_t* code;
  const float* C;
  const float* gscale;
  float* dz;
  int64_t lddz, N;
  int accumulate;
  unsigned int* queue;
};

// Persistent grid as dc_grad_rows_kernel: one workgroup per CU holds a 64-column codebook slab in LDS; a wave draws 64
// consecutive nodes, a 16-lane group owns a node and walks its incidence list 32 keys at a time (two per lane).  Two-deep
// software pipeline over the nodes: the keys of node t+2 are in flight while the coefficients of node t+1 are looked up
// (row code[node] of C, L2-resident) and node t is summed.
struct DcPairNode {
  int lo, hi, kself;
  unsigned k0, k1;  // keys lo + gl and lo + 16 + gl (0 = none; key 0 with coefficient 0 adds 0 * W[0])
  float c0, c1;
};
__global__ __launch_bounds__(DC_GRAD_THREADS) void dc_pair_grad_kernel(DcPairGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* lds4 = reinterpret_cast<float4*>(smem);
  const int nslab = a.D / DC_CW;
  const int slab = blockIdx.x % nslab;
  dc_fill_slab(a.W, a.ldw, a.K, slab, lds4);
  const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
  const int col0 = slab * DC_CW + 4 * gl;
  const float gs = a.gscale[0];
  unsigned int* q = a.queue + slab * 32;
  for (;;) {
    unsigned int dv = 0;
    if (lane == 0) dv = atomicAdd(q, 64u);
    const int64_t base_n = (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)dv);
    if (base_n >= a.N) break;
    const int64_t n = base_n + lane;
    int lo = 0, hi = 0, ks = 0;
    if (n < a.N) {
      lo = a.rowptr[n];
      hi = a.rowptr[n + 1];
      ks = dc_clamp_code(a.code[n], a.K);
    }
    const int cnt = a.N - base_n < 64 ? (int)(a.N - base_n) : 64;
    auto bounds = [&](int it, DcPairNode& s) {  // + the keys of the first two batches
      const int j = (it * 4 + grp) & 63;
      s.lo = __shfl(lo, j, 64);
      s.hi = __shfl(hi, j, 64);
      s.kself = __shfl(ks, j, 64);
      if (it * 4 >= cnt) s.hi = s.lo;  // wave-uniform
      s.k0 = 0u; s.k1 = 0u;
      if (s.lo + gl < s.hi) s.k0 = a.keys[s.lo + gl];
      if (s.lo + 16 + gl < s.hi) s.k1 = a.keys[s.lo + 16 + gl];
    };
    auto coefs = [&](DcPairNode& s) {
      const float* row = a.C + (size_t)s.kself * a.K;
      s.c0 = 0.f; s.c1 = 0.f;
      if (s.lo + gl < s.hi) s.c0 = row[s.k0 & 1023u];
      if (s.lo + 16 + gl < s.hi) s.c1 = row[s.k1 & 1023u];
    };
    DcPairNode s0, s1, s2;
    bounds(0, s0);
    bounds(1, s1);
    coefs(s0);
    for (int it = 0; it * 4 < cnt; ++it) {
      bounds(it + 2, s2);
      coefs(s1);
      const int j = it * 4 + grp;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      const int len = s0.hi - s0.lo;
      dc_batch1(lds4, gl, s0.c0, (int)(s0.k0 & 1023u), len < 0 ? 0 : len, s);
      dc_batch1(lds4, gl, s0.c1, (int)(s0.k1 & 1023u), len - 16 < 0 ? 0 : len - 16, s);
      const float* row = a.C + (size_t)s0.kself * a.K;
      for (int base = s0.lo + 32; __any(base < s0.hi); base += 16) {  // lists longer than 32: keys and coefficients fetched in place
        unsigned k = 0u;
        float c = 0.f;
        if (base + gl < s0.hi) {
          k = a.keys[base + gl];
          c = row[k & 1023u];
        }
        const int nb = s0.hi - base;
        dc_batch1(lds4, gl, c, (int)(k & 1023u), nb < 0 ? 0 : nb, s);
      }
      if (j < cnt) {
        float* op = a.dz + (base_n + j) * a.lddz + col0;
        float4 o = make_float4(gs * s.x, gs * s.y, gs * s.z, gs * s.w);
        if (a.accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(op);
          o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
        }
        *reinterpret_cast<float4*>(op) = o;
      }
      s0 = s1;
      s1 = s2;
    }
  }
}

struct DcPairWs {
  uint32_t* keys_out;
  int32_t* rowptr;
  unsigned int* queue;
  void* temp;
  size_t temp_bytes;
};
size_t dc_pair_bwd_layout(int64_t E, int64_t N, void* ws, size_t ws_bytes, DcPairWs* out) {
  const size_t M = (size_t)(2 * E);
  const size_t tb = dc_pair_sort_temp(M, N);
  if (tb == 0) return 0;
  const size_t need = ceeg_ws_part(M, 4) + ceeg_ws_part((size_t)N + 1, 4) + 1024 + ceeg_align_up(tb, 256);
  if (out) {
    CeegWs a(ws, ws_bytes);
    out->keys_out = a.take<uint32_t>(M);
    out->rowptr = a.take<int32_t>((size_t)N + 1);
    out->queue = a.take<unsigned int>(256);
    out->temp = a.take<char>(tb);
    out->temp_bytes = tb;
    if (!a.ok) return 0;
  }
  return need;
}
size_t dc_pair_fwd_bytes(int n_e, int64_t N) {  // G, T, block partials (+ 3 floats for the finalize), 16-bit codes
  return 2 * ceeg_ws_part((size_t)n_e * n_e, 4) + ceeg_ws_part((size_t)(DC_EDGE_BLOCKS + 1) * 3, 4) + ceeg_ws_part((size_t)N, 2);
}

}  // namespace

extern "C" int ceeg_pair_bce_codes_supported(int n_e, int D, int64_t N, int64_t E) { return dc_pair_supported(n_e, D, N, E) ? 1 : 0; }

extern "C" size_t ceeg_pair_bce_codes_ws_bytes(int64_t E, int64_t N, int n_e) {
  if (E < 0 || N <= 0 || N > DC_PAIR_MAX_N || n_e <= 0 || E >= ((int64_t)1 << 30)) return 0;
  const size_t fwd = dc_pair_fwd_bytes(n_e, N);
  const size_t bwd = dc_pair_bwd_layout(E, N, nullptr, 0, nullptr);
  return fwd > bwd ? fwd : bwd;
}

extern "C" int ceeg_pair_bce_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* i, const int32_t* j,
                                       int64_t E, int64_t N, int negative, uint32_t* keys, float* coef_table, float* loss_sum, void* ws,
                                       size_t ws_bytes, void* stream) {
  if (!W || !code || !loss_sum || !coef_table || E < 0 || N <= 0 || (E > 0 && (!i || !j))) return CEEG_EINVAL;
  if (!dc_pair_supported(n_e, D, N, E)) return CEEG_EUNSUPPORTED;
  if (ldw < D || (ldw % 4) != 0 || !ceeg_aligned(W, 16) || (keys && !ceeg_aligned(keys, 8)) || !ceeg_aligned(coef_table, 4)) return CEEG_EINVAL;
  if (ws_bytes < dc_pair_fwd_bytes(n_e, N) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t kk = ceeg_ws_part((size_t)n_e * n_e, 4);
  float* G = static_cast<float*>(ws);
  float* T = reinterpret_cast<float*>(static_cast<char*>(ws) + kk);
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + 2 * kk);
  unsigned short* code16 = reinterpret_cast<unsigned short*>(static_cast<char*>(ws) + 2 * kk + ceeg_ws_part((size_t)(DC_EDGE_BLOCKS + 1) * 3, 4));
  hipLaunchKernelGGL(dc_code16_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, code, N, n_e, code16);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(dc_gram_kernel, dim3((unsigned)((n_e + 255) / 256), (unsigned)n_e), dim3(256), 0, st, W, ldw, n_e, D, G);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(dc_pair_tables_kernel, dim3((unsigned)((n_e * n_e + 255) / 256)), dim3(256), 0, st, G, n_e, negative & 1, T, coef_table);
  CEEG_CHECK_LAUNCH();
  int64_t nb = (E + 255) / 256;
  nb = nb < 1 ? 1 : (nb > DC_EDGE_BLOCKS ? DC_EDGE_BLOCKS : nb);
  hipLaunchKernelGGL(dc_pair_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, T, n_e, code16, i, j, E, N, reinterpret_cast<uint2*>(keys), part);
  CEEG_CHECK_LAUNCH();
  float* tmp = part + (size_t)DC_EDGE_BLOCKS * 3;
  hipLaunchKernelGGL(dc_finalize_kernel, dim3(1), dim3(1024), 0, st, part, nb, tmp);
  CEEG_CHECK_LAUNCH();
  if (hipMemcpyAsync(loss_sum, tmp, 2 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return CEEG_ELAUNCH;
  return CEEG_OK;
}

extern "C" int ceeg_pair_bce_codes_bwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const uint32_t* keys,
                                       const float* coef_table, int64_t E, int64_t N, const float* gscale, int accumulate, float* dz,
                                       int64_t lddz, void* ws, size_t ws_bytes, void* stream) {
  if (!W || !code || !coef_table || !gscale || !dz || E < 0 || N <= 0 || (E > 0 && !keys)) return CEEG_EINVAL;
  if (!dc_pair_supported(n_e, D, N, E)) return CEEG_EUNSUPPORTED;
  if (ldw < D || (ldw % 4) != 0 || lddz < D || (lddz % 4) != 0 || !ceeg_aligned(W, 16) || !ceeg_aligned(dz, 16)) return CEEG_EINVAL;
  DcPairWs w;
  const size_t need = dc_pair_bwd_layout(E, N, nullptr, 0, nullptr);
  if (need == 0 || ws_bytes < need || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  if (dc_pair_bwd_layout(E, N, ws, ws_bytes, &w) == 0) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t M = 2 * E;
  if (M > 0) {
    size_t tb = w.temp_bytes;
    if (rocprim::radix_sort_keys(w.temp, tb, keys, w.keys_out, (size_t)M, (unsigned)DC_PAIR_NODE_SHIFT,
                                 (unsigned)DC_PAIR_NODE_SHIFT + dc_key_bits(N), st, false) != hipSuccess)
      return CEEG_ELAUNCH;
    hipLaunchKernelGGL(dc_pair_rowptr_kernel, dim3((unsigned)((M / 4 + 1 + 255) / 256)), dim3(256), 0, st, w.keys_out, M, N, w.rowptr);
    CEEG_CHECK_LAUNCH();
  } else if (hipMemsetAsync(w.rowptr, 0, ((size_t)N + 1) * 4, st) != hipSuccess) {  // the empty list: every row is empty
    return CEEG_ELAUNCH;
  }
  if (hipMemsetAsync(w.queue, 0, 1024, st) != hipSuccess) return CEEG_ELAUNCH;
  DcPairGradArgs a;
  a.W = W; a.ldw = ldw; a.K = n_e; a.D = D;
  a.rowptr = w.rowptr; a.keys = w.keys_out; a.code = code; a.C = coef_table; a.gscale = gscale;
  a.dz = dz; a.lddz = lddz; a.N = N; a.accumulate = accumulate ? 1 : 0; a.queue = w.queue;
  const int nslab = D / DC_CW;
  int64_t per_slab = (N + (DC_GRAD_THREADS / 16) - 1) / (DC_GRAD_THREADS / 16);
  const int64_t cap = DC_MAX_GRAD_BLOCKS / nslab;
  per_slab = per_slab < 1 ? 1 : (per_slab > cap ? cap : per_slab);
  hipLaunchKernelGGL(dc_pair_grad_kernel, dim3((unsigned)(per_slab * nslab)), dim3(DC_GRAD_THREADS), (size_t)n_e * DC_CW * 4, st, a);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

// negative_sampling + the pair-list forward in one launch (see dc_sample_pair_fwd_kernel).
extern "C" int ceeg_neg_sample_pair_bce_codes_fwd(const float* W, int64_t ldw, int n_e, int D, const int32_t* code, const int32_t* rowptr,
                                                  const int32_t* col, int64_t N, int64_t num, uint64_t seed, const uint32_t* filter,
                                                  int filter_bits, int32_t* neg_src, int32_t* neg_dst, uint32_t* keys, float* coef_table,
                                                  float* loss_sum, void* ws, size_t ws_bytes, void* stream) {
  if (!W || !code || !rowptr || !col || !loss_sum || !coef_table || num < 0 || N <= 1 || (num > 0 && (!neg_src || !neg_dst))) return CEEG_EINVAL;
  if (filter && (filter_bits < 13 || filter_bits > 32)) return CEEG_EINVAL;
  if (!dc_pair_supported(n_e, D, N, num)) return CEEG_EUNSUPPORTED;
  if (ldw < D || (ldw % 4) != 0 || !ceeg_aligned(W, 16) || (keys && !ceeg_aligned(keys, 8)) || !ceeg_aligned(coef_table, 4)) return CEEG_EINVAL;
  if (ws_bytes < dc_pair_fwd_bytes(n_e, N) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t kk = ceeg_ws_part((size_t)n_e * n_e, 4);
  float* G = static_cast<float*>(ws);
  float* T = reinterpret_cast<float*>(static_cast<char*>(ws) + kk);
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + 2 * kk);
  unsigned short* code16 = reinterpret_cast<unsigned short*>(static_cast<char*>(ws) + 2 * kk + ceeg_ws_part((size_t)(DC_EDGE_BLOCKS + 1) * 3, 4));
  hipLaunchKernelGGL(dc_code16_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, code, N, n_e, code16);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(dc_gram_kernel, dim3((unsigned)((n_e + 255) / 256), (unsigned)n_e), dim3(256), 0, st, W, ldw, n_e, D, G);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(dc_pair_tables_kernel, dim3((unsigned)((n_e * n_e + 255) / 256)), dim3(256), 0, st, G, n_e, 1, T, coef_table);
  CEEG_CHECK_LAUNCH();
  int64_t nb = (num + 255) / 256;
  nb = nb < 1 ? 1 : (nb > DC_EDGE_BLOCKS ? DC_EDGE_BLOCKS : nb);
  if (filter)
    hipLaunchKernelGGL(dc_sample_pair_fwd_kernel<true>, dim3((unsigned)nb), dim3(256), 0, st, T, n_e, code16, rowptr, col, N, num, seed, filter,
                       (unsigned)filter_bits, neg_src, neg_dst, reinterpret_cast<uint2*>(keys), part);
  else
    hipLaunchKernelGGL(dc_sample_pair_fwd_kernel<false>, dim3((unsigned)nb), dim3(256), 0, st, T, n_e, code16, rowptr, col, N, num, seed, filter,
                       0u, neg_src, neg_dst, reinterpret_cast<uint2*>(keys), part);
  CEEG_CHECK_LAUNCH();
  float* tmp = part + (size_t)DC_EDGE_BLOCKS * 3;
  hipLaunchKernelGGL(dc_finalize_kernel, dim3(1), dim3(1024), 0, st, part, nb, tmp);
  CEEG_CHECK_LAUNCH();
  if (hipMemcpyAsync(loss_sum, tmp, 2 * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) return CEEG_ELAUNCH;
  return CEEG_OK;
}

// Backward of BOTH pair sets of the literal recon_loss in one gradient kernel: the positives' three lists (ceeg_decoder_codes_bwd with
// sig_a = 0) and the sampled pairs' node-sorted incidence keys as a fourth list; dz is written once.
extern "C" size_t ceeg_decoder_codes_bwd_pairs_ws_bytes(int64_t N, int64_t nnz, int n_e, int D, int64_t n_long, int64_t n_tlong, int64_t E) {
  const size_t a = ceeg_decoder_codes_ws_bytes(N, nnz, n_e, D, n_long, n_tlong);
  const size_t b = dc_pair_bwd_layout(E, N, nullptr, 0, nullptr);
  if (a == 0 || b == 0) return 0;
  return ceeg_align_up(a, 256) + b;
}

extern "C" int ceeg_decoder_codes_bwd_pairs(const float* W, int64_t ldw, int n_e, int D, const int32_t* rowptr, const int32_t* t_rowptr,
                                            const int32_t* t_perm, int64_t N, int64_t nnz, const int32_t* long_items, int64_t n_long,
                                            const int32_t* t_long_items, int64_t n_tlong, int64_t max_row_nnz, int64_t max_trow_nnz,
                                            const void* rec, const int32_t* code, const uint32_t* keys, const float* coef_table, int64_t E,
                                            const float* gscale, float* dz, int64_t lddz, void* ws, size_t ws_bytes, void* stream) {
  if (!code || !coef_table || E < 0 || (E > 0 && !keys) || N <= 0) return CEEG_EINVAL;
  if (!dc_pair_supported(n_e, D, N, E)) return CEEG_EUNSUPPORTED;
  const size_t need_a = ceeg_align_up(ceeg_decoder_codes_ws_bytes(N, nnz, n_e, D, n_long, n_tlong), 256);
  const size_t need_b = dc_pair_bwd_layout(E, N, nullptr, 0, nullptr);
  if (need_a == 0 || need_b == 0 || ws_bytes < need_a + need_b || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  DcPairWs w;
  if (dc_pair_bwd_layout(E, N, static_cast<char*>(ws) + need_a, ws_bytes - need_a, &w) == 0) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t M = 2 * E;
  if (M > 0) {
    size_t tb = w.temp_bytes;
    if (rocprim::radix_sort_keys(w.temp, tb, keys, w.keys_out, (size_t)M, (unsigned)DC_PAIR_NODE_SHIFT,
                                 (unsigned)DC_PAIR_NODE_SHIFT + dc_key_bits(N), st, false) != hipSuccess)
      return CEEG_ELAUNCH;
    hipLaunchKernelGGL(dc_pair_rowptr_kernel, dim3((unsigned)((M / 4 + 1 + 255) / 256)), dim3(256), 0, st, w.keys_out, M, N, w.rowptr);
    CEEG_CHECK_LAUNCH();
  } else if (hipMemsetAsync(w.rowptr, 0, ((size_t)N + 1) * 4, st) != hipSuccess) {
    return CEEG_ELAUNCH;
  }
  DcPairList pl;
  pl.p_rowptr = w.rowptr; pl.p_keys = w.keys_out; pl.p_code = code; pl.p_C = coef_table;
  return dc_decoder_bwd_impl(W, ldw, n_e, D, rowptr, t_rowptr, t_perm, N, nnz, 0, 0, 1, N, long_items, n_long, t_long_items, n_tlong,
                             max_row_nnz, max_trow_nnz, rec, gscale, dz, lddz, nullptr, ws, need_a, stream, pl);
}
