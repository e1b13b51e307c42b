// InnerProductDecoder + GAE.recon_loss for gfx950 (upstream PyG semantics), literal
// "+EPS" form:  v = <z_i, z_j>;  pos: -log(sigmoid(v)+EPS);  neg: -log((1-sigmoid(v))+EPS), EPS = 1e-15.
//
// HBM/L2-bound gathers: each 16-byte piece of a latent row is one lane's load, the dot product is a
// cross-lane butterfly, the per-edge derivative d(term)/dv is stored so the backward is a pair of
// deterministic SpMMs (spmm.hip) — no float atomics anywhere.  Loss sums: thread -> wave butterfly ->
// block -> one fp64 pass over the block partials (fixed order).
#include <stdlib.h>
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

__global__ __launch_bounds__(1024) void dec_finalize_kernel(const float* __restrict__ part, int nblk, int nout, float* __restrict__ out) {
  __shared__ double red[3][1024];
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblk; i += 1024) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (k < nout) s[k] += (double)part[(size_t)i * 3 + k];
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
  if ((int)threadIdx.x < nout) out[threadIdx.x] = (float)red[threadIdx.x][0];
}

// ------------------------------------------------------------------ edge-list form (any order)
template <int LPR>
__global__ __launch_bounds__(DEC_WAVES * 64) void edge_bce_fwd_kernel(
    const float* __restrict__ z, int64_t ldz, const int32_t* __restrict__ zidx, const int32_t* __restrict__ ei,
    const int32_t* __restrict__ ej, int64_t E, int D, int negative, float* __restrict__ coef, float* __restrict__ prob,
    float* __restrict__ part) {
  __shared__ float red[3 * DEC_WAVES];
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int fl = lane % LPR, sub = lane / LPR;
  const bool fl_ok = 4 * fl < D;
  const int64_t wave = (int64_t)blockIdx.x * DEC_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * DEC_WAVES;
  float acc = 0.f, cnt = 0.f;
  for (int64_t e0 = wave * G; e0 < E; e0 += nwaves * G) {
    const int64_t e = e0 + sub;
    int ni = -1, nj = -1;
    if (e < E) { ni = ei[e]; nj = ej[e]; }
    const bool ok = ni >= 0 && nj >= 0;  // a pair with a negative id is a sample the sampler gave up on: skipped
    float d = 0.f;
    if (ok && fl_ok) {
      if (zidx) { ni = zidx[ni]; nj = zidx[nj]; }  // quantised latent: node n's row is z[zidx[n]] (z = the codebook)
      const float4 a = *reinterpret_cast<const float4*>(z + (int64_t)ni * ldz + 4 * fl);
      const float4 b = *reinterpret_cast<const float4*>(z + (int64_t)nj * ldz + 4 * fl);
      d = dot4(a, b);
    }
    d = group_sum<LPR>(d);
    if (e < E && fl == 0) {
      float t = 0.f, c = 0.f;
      if (ok) {
        if (negative & 1) dec_neg(d, t, c); else dec_pos(d, t, c);
        cnt += 1.0f;
      }
      if (coef) coef[e] = c;
      if (prob) prob[e] = ok ? ((negative & 2) ? d : dec_sigmoid(d)) : 0.f;  // bit 1: the raw inner product (sigmoid=False)
      acc += t;
    }
  }
  dec_block_out(acc, cnt, 0.f, red, part);
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

// Transposed butterfly: every lane enters with NV partial sums (one per (step, kind) of a round) that must each be
// summed over the LPR lanes of its group.  Instead of NV full butterflies (NV * log2(LPR) exchanges) each exchange
// step hands HALF of the values to the partner lane and keeps the other half, so the count halves every step:
// NV/2 + NV/4 + ... exchanges.  After H = min(log2 LPR, log2 NV) halvings a lane holds NVF = NV >> H finished
// values (slot i = value (fl >> (M-H)) * NVF + i); left-over strides finish as plain butterflies.
template <int STRIDE, int NV>
struct DecTRed {
  static __device__ __forceinline__ void run(float* v, int fl) {
    if constexpr (STRIDE >= 1) {
      if constexpr (NV > 1) {
        const bool up = (fl & STRIDE) != 0;
#pragma unroll
        for (int i = 0; i < NV / 2; ++i) {
          const float send = up ? v[i] : v[i + NV / 2];
          const float keep = up ? v[i + NV / 2] : v[i];
          v[i] = keep + __shfl_xor(send, STRIDE, 64);
        }
        DecTRed<STRIDE / 2, NV / 2>::run(v, fl);
      } else {
        v[0] += __shfl_xor(v[0], STRIDE, 64);
        DecTRed<STRIDE / 2, 1>::run(v, fl);
      }
    }
  }
};
constexpr int dec_log2(int x) { return x <= 1 ? 0 : 1 + dec_log2(x / 2); }

struct DecRowState {
  float4 zr, zs;      // this lane's piece of z[r] and z[sigma(r)]
  float4 ga, gb;      // running sum_p cp[p] z[col p] / sum_p cn[p] z[col p] over this lane's sub-group of entries
  float s_pos, s_neg, s_cnt;
};

// One round = T (<= 4: keeps the kernel near 112 VGPRs, 4 waves per SIMD) gather steps (T*G entries of the current 64-entry batch, starting at step i0): gathers in flight together,
// 2T dot products reduced by the transposed butterfly, ONE evaluation of the sigmoid/log/derivative per 2T*G values
// (spread over the lanes instead of repeated per step), then the coefficients go back to the lanes that still hold
// the gathered pieces for the row-side gradient sums.
template <int LPR, int T, bool GRAD>
__device__ __forceinline__ void dec_round(const float* __restrict__ z, int64_t ldz, int my_g, int my_valid, int base, int i0, int ns, int n,
                                          int lane, bool fl_ok, float* __restrict__ coef_pos, float* __restrict__ coef_neg, int cstride,
                                          DecRowState& st) {
  constexpr int G = 64 / LPR;
  constexpr int M = dec_log2(LPR);
  constexpr int NV = 2 * T;
  constexpr int H = M < dec_log2(NV) ? M : dec_log2(NV);
  constexpr int NVF = NV >> H;
  const int fl = lane % LPR, sub = lane / LPR;
  float4 x[T];
  float v[NV];
#pragma unroll
  for (int k = 0; k < T; ++k) {
    const int j = (i0 + k) * G + sub;
    const int c = __shfl(my_g, j & 63, 64);
    x[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 + k < ns && j < n && fl_ok) x[k] = *reinterpret_cast<const float4*>(z + (int64_t)c * ldz + 4 * fl);
  }
#pragma unroll
  for (int k = 0; k < T; ++k) {
    v[k] = dot4(x[k], st.zr);
    v[T + k] = dot4(x[k], st.zs);
  }
  DecTRed<LPR / 2, NV>::run(v, fl);
  const int top = fl >> (M - H);
  const bool writer = (fl & ((1 << (M - H)) - 1)) == 0;
  float cf[NVF];
#pragma unroll
  for (int i = 0; i < NVF; ++i) {
    const int idx = top * NVF + i;
    const int kind = idx / T, k = idx % T;
    const int j = (i0 + k) * G + sub;
    const bool ok = (i0 + k < ns) && j < n;
    const int vld = __shfl(my_valid, j & 63, 64);
    const float s = dec_sigmoid(v[i]);
    const float om = 1.0f - s;
    const float a = kind ? om : s;
    const float sso = s * om;
    float term = -logf(a + DEC_EPS);
    float coef = (kind ? sso : -sso) / (a + DEC_EPS);
    const bool live = ok && (kind == 0 || vld != 0);
    if (!live) { term = 0.f; coef = 0.f; }
    if (ok && writer) {
      (kind ? coef_neg : coef_pos)[(int64_t)(base + j) * cstride] = coef;
      if (kind == 0) st.s_pos += term;
      else if (vld) { st.s_neg += term; st.s_cnt += 1.0f; }
    }
    cf[i] = coef;
  }
  if constexpr (GRAD) {
    const int gbase = lane & ~(LPR - 1);
#pragma unroll
    for (int k = 0; k < T; ++k) {
      const int ip = k, in = T + k;
      const float cp = __shfl(cf[ip % NVF], gbase | ((ip / NVF) << (M - H)), 64);
      const float cn = __shfl(cf[in % NVF], gbase | ((in / NVF) << (M - H)), 64);
      st.ga.x = __fmaf_rn(cp, x[k].x, st.ga.x); st.ga.y = __fmaf_rn(cp, x[k].y, st.ga.y);
      st.ga.z = __fmaf_rn(cp, x[k].z, st.ga.z); st.ga.w = __fmaf_rn(cp, x[k].w, st.ga.w);
      st.gb.x = __fmaf_rn(cn, x[k].x, st.gb.x); st.gb.y = __fmaf_rn(cn, x[k].y, st.gb.y);
      st.gb.z = __fmaf_rn(cn, x[k].z, st.gb.z); st.gb.w = __fmaf_rn(cn, x[k].w, st.gb.w);
    }
  }
}

// first row r in [0, N] with rowptr[r] >= target
__device__ __forceinline__ int64_t dec_row_lower_bound(const int32_t* __restrict__ rowptr, int64_t N, int64_t target) {
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] < target) lo = mid + 1; else hi = mid;
  }
  return lo;
}

constexpr int DEC_ROWS = 16;          // consecutive rows a wave draws at a time
constexpr int DEC_BLOCKS_PER_XCD = 256;
constexpr size_t DEC_QUEUE_BYTES = 8 * 128;

struct DecRow {     // everything a row needs before its gathers, fetched one row ahead
  int p_lo, p_hi, s_lo, s_hi, sr;
  int bc, sc;       // lane's entry of the row's first 64 columns / of row sigma(r)'s first 64 columns
  int bg;           // row of z to gather for bc (bc itself, or zidx[bc] when z is a table)
  float4 zr, zs;
};

// positions [p_lo, p_hi) of ONE row (a whole row, or a DEC_CHUNK-aligned piece of a hub row): dots, coefficients,
// loss terms and - with GRAD - this piece's share of the row-side gradient sums, all into `st`.
// (bc, bg): the lane's column / gather row of the first batch when the caller prefetched it (have_first).
template <int LPR, bool GRAD>
__device__ __forceinline__ void dec_fwd_segment(const float* __restrict__ z, int64_t ldz, const int32_t* __restrict__ zidx,
                                                const int32_t* __restrict__ col, int p_lo, int p_hi, int s_lo, int s_hi, int sr, int sc,
                                                bool have_first, int bc, int bg, int lane, bool fl_ok, float* __restrict__ coef_pos,
                                                float* __restrict__ coef_neg, int cstride, DecRowState& st) {
  constexpr int G = 64 / LPR;
  const int s_n = s_hi - s_lo;
  for (int base = p_lo; base < p_hi; base += 64) {
    const int n = p_hi - base < 64 ? p_hi - base : 64;
    int my_c = bc, my_g = bg;
    if (!have_first || base != p_lo) {
      my_c = lane < n ? col[base + lane] : 0;
      my_g = (zidx && lane < n) ? zidx[my_c] : my_c;
    }
    bool hit = false;
    if (s_n <= 64) {
      for (int t = 0; t < s_n; ++t) hit |= (my_c == __builtin_amdgcn_readlane(sc, t));
    } else if (lane < n) {
      hit = sorted_has(col, s_lo, s_hi, my_c);
    }
    const int my_valid = (lane < n && my_c != sr && !hit) ? 1 : 0;
    const int ns = (n + G - 1) / G;  // steps in this batch (wave-uniform)
    int i0 = 0;
    for (; i0 + 4 <= ns; i0 += 4) dec_round<LPR, 4, GRAD>(z, ldz, my_g, my_valid, base, i0, ns, n, lane, fl_ok, coef_pos, coef_neg, cstride, st);
    const int rem = ns - i0;  // one round of the next power of two covers the tail
    if (rem > 2) dec_round<LPR, 4, GRAD>(z, ldz, my_g, my_valid, base, i0, ns, n, lane, fl_ok, coef_pos, coef_neg, cstride, st);
    else if (rem == 2) dec_round<LPR, 2, GRAD>(z, ldz, my_g, my_valid, base, i0, ns, n, lane, fl_ok, coef_pos, coef_neg, cstride, st);
    else if (rem == 1) dec_round<LPR, 1, GRAD>(z, ldz, my_g, my_valid, base, i0, ns, n, lane, fl_ok, coef_pos, coef_neg, cstride, st);
  }
  if constexpr (GRAD && G > 1) {
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) {
      st.ga.x += __shfl_xor(st.ga.x, o, 64); st.ga.y += __shfl_xor(st.ga.y, o, 64);
      st.ga.z += __shfl_xor(st.ga.z, o, 64); st.ga.w += __shfl_xor(st.ga.w, o, 64);
      st.gb.x += __shfl_xor(st.gb.x, o, 64); st.gb.y += __shfl_xor(st.gb.y, o, 64);
      st.gb.z += __shfl_xor(st.gb.z, o, 64); st.gb.w += __shfl_xor(st.gb.w, o, 64);
    }
  }
}

__device__ __forceinline__ int dec_row_of(const int32_t* __restrict__ rowptr, int64_t N, int p) {
  int64_t lo = 0, hi = N;  // last r in [0,N) with rowptr[r] <= p
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] <= p) lo = mid; else hi = mid;
  }
  return (int)lo;
}

constexpr int DEC_CHUNK = 256;  // a row longer than this is cut at the GLOBAL positions k*DEC_CHUNK (as spmm.hip)

// Pieces of hub rows (in-degree > DEC_CHUNK): wave w handles global positions [w*DEC_CHUNK, (w+1)*DEC_CHUNK) - one wave
// per whole row would serialise a 1e5-entry row.  Each piece leaves its coefficients, its loss partial (slot
// lslot0 + 2w + slot) and, with GRAD, its share of the row-side sums in `gpart`; the row kernel adds the shares in order.
template <int LPR, bool GRAD>
__global__ __launch_bounds__(DEC_WAVES * 64) void decoder_fused_fwd_long_kernel(
    const float* __restrict__ z, int64_t ldz, const int32_t* __restrict__ zidx, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, int64_t N, int D, int64_t sig_a, int64_t sig_b, int64_t sig_mod, float* __restrict__ coef_pos,
    float* __restrict__ coef_neg, int cstride, float* __restrict__ gpart, float* __restrict__ part, int64_t lslot0) {
  const int lane = threadIdx.x & 63;
  const int fl = lane % LPR;
  const bool fl_ok = 4 * fl < D;
  const int64_t w = (int64_t)blockIdx.x * DEC_WAVES + (threadIdx.x >> 6);
  const int nnz = rowptr[N];
  const int64_t p0l = w * DEC_CHUNK;
  if (p0l >= nnz) return;
  const int p0 = (int)p0l;
  const int p1 = p0 + DEC_CHUNK < nnz ? p0 + DEC_CHUNK : nnz;
  const int r0 = dec_row_of(rowptr, N, p0);
  const int r1 = dec_row_of(rowptr, N, p1 - 1);
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto piece = [&](int r, int lo, int hi, int slot) {
    DecRowState st;
    st.s_pos = st.s_neg = st.s_cnt = 0.f;
    st.ga = zero4; st.gb = zero4; st.zr = zero4; st.zs = zero4;
    const int sr = (int)ceeg_block_affine((int64_t)r, sig_a, sig_b, sig_mod);
    const int s_lo = rowptr[sr], s_hi = rowptr[sr + 1];
    const int sc = lane < s_hi - s_lo ? col[s_lo + lane] : 0;
    if (fl_ok) {
      st.zr = *reinterpret_cast<const float4*>(z + (int64_t)(zidx ? zidx[r] : r) * ldz + 4 * fl);
      st.zs = *reinterpret_cast<const float4*>(z + (int64_t)(zidx ? zidx[sr] : sr) * ldz + 4 * fl);
    }
    dec_fwd_segment<LPR, GRAD>(z, ldz, zidx, col, lo, hi, s_lo, s_hi, sr, sc, false, 0, 0, lane, fl_ok, coef_pos, coef_neg, cstride, st);
    if constexpr (GRAD) {
      if (lane < LPR && fl_ok) {
        float* o = gpart + ((size_t)w * 2 + slot) * 2 * (4 * LPR) + 4 * fl;
        *reinterpret_cast<float4*>(o) = st.ga;
        *reinterpret_cast<float4*>(o + 4 * LPR) = st.gb;
      }
    }
    const float t0 = ceeg_wave_sum(st.s_pos), t1 = ceeg_wave_sum(st.s_neg), t2 = ceeg_wave_sum(st.s_cnt);
    if (lane == 0) {
      float* o = part + (size_t)(lslot0 + w * 2 + slot) * 3;
      o[0] = t0; o[1] = t1; o[2] = t2;
    }
  };
  if (rowptr[r0 + 1] - rowptr[r0] > DEC_CHUNK) piece(r0, p0, rowptr[r0 + 1] < p1 ? rowptr[r0 + 1] : p1, 0);
  if (r1 != r0 && rowptr[r1 + 1] - rowptr[r1] > DEC_CHUNK) piece(r1, rowptr[r1], p1, 1);
}

// Row r of the dst-sorted CSR: positives (r, col p), negatives (col p, sigma(r)) - both share the gather of z[col p].
// GRAD also leaves the row-side gradient sums (unscaled):  grow_pos[r] = sum_p cp[p] z[col p],
// grow_neg[sigma(r)] = sum_p cn[p] z[col p]  (sigma is a bijection: every row of both tables is written once).
//
// Scheduling (as spmm.hip): persistent grid, the workgroups sharing blockIdx%8 (one XCD, one L2) own a contiguous,
// nonzero-balanced range of rows and sweep it in order - every WAVE draws DEC_ROWS consecutive rows from the XCD's
// counter (next draw in flight while the current rows are processed; the counter only grows, so every wave leaves
// once its draw passes the end).  Per draw, the row pointers of the DEC_ROWS rows and of their sigma-rows come in
// one lane-parallel load; per row, the column batch, sigma(r)'s column batch and the two fixed latent rows are
// fetched one row ahead, so a row costs ~one gather latency instead of the
// rowptr -> col -> binary search -> gather chain.  The collision test (is col p -> sigma(r) an edge?) compares
// against sigma(r)'s columns held across the lanes (memory binary search only for sigma-rows longer than 64).
template <int LPR, bool GRAD>
__global__ __launch_bounds__(DEC_WAVES * 64) void decoder_fused_fwd_kernel(
    const float* __restrict__ z, int64_t ldz, const int32_t* __restrict__ zidx, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, int64_t N, int D, int64_t sig_a, int64_t sig_b, int64_t sig_mod, float* __restrict__ coef_pos, float* __restrict__ coef_neg, int cstride, float* __restrict__ grow_pos,
    float* __restrict__ grow_neg, int64_t ldg, unsigned int* __restrict__ queue /* 8 counters, 128 B apart, zeroed */,
    const float* __restrict__ gpart /* shares of hub rows */, float* __restrict__ part) {
  const int lane = threadIdx.x & 63;
  const int fl = lane % LPR;
  const bool fl_ok = 4 * fl < D;
  const int xcd = blockIdx.x & 7;
  const int64_t nnz = rowptr[N];
  const int64_t r_begin = xcd == 0 ? 0 : dec_row_lower_bound(rowptr, N, nnz * xcd / 8);
  const int64_t r_end = xcd == 7 ? N : dec_row_lower_bound(rowptr, N, nnz * (xcd + 1) / 8);
  unsigned int* q = queue + xcd * 32;
  DecRowState st;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  unsigned int dv = 0;
  if (lane == 0) dv = atomicAdd(q, (unsigned)DEC_ROWS);
  for (;;) {
    const int64_t base_r = r_begin + (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)dv);
    if (base_r >= r_end) break;
    if (lane == 0) dv = atomicAdd(q, (unsigned)DEC_ROWS);  // the next draw lands while these rows run
    const int cnt = r_end - base_r < DEC_ROWS ? (int)(r_end - base_r) : DEC_ROWS;
    st.s_pos = st.s_neg = st.s_cnt = 0.f;
    int rp = 0, sri = 0, sp_lo = 0, sp_hi = 0, gri = 0, gsi = 0;
    if (lane <= cnt) rp = rowptr[base_r + lane];
    if (lane < cnt) {
      sri = (int)ceeg_block_affine(base_r + lane, sig_a, sig_b, sig_mod);
      sp_lo = rowptr[sri];
      sp_hi = rowptr[sri + 1];
      gri = (int)(base_r + lane);
      gsi = sri;
      if (zidx) { gri = zidx[gri]; gsi = zidx[gsi]; }
    }
    auto fetch = [&](int i) -> DecRow {
      DecRow w;
      w.p_lo = __builtin_amdgcn_readlane(rp, i);
      w.p_hi = __builtin_amdgcn_readlane(rp, i + 1);
      w.s_lo = __builtin_amdgcn_readlane(sp_lo, i);
      w.s_hi = __builtin_amdgcn_readlane(sp_hi, i);
      w.sr = __builtin_amdgcn_readlane(sri, i);
      w.bc = 0; w.sc = 0; w.bg = 0; w.zr = zero4; w.zs = zero4;
      if (w.p_lo != w.p_hi && w.p_hi - w.p_lo <= DEC_CHUNK) {
        if (lane < w.p_hi - w.p_lo) {
          w.bc = col[w.p_lo + lane];
          w.bg = zidx ? zidx[w.bc] : w.bc;
        }
        if (lane < w.s_hi - w.s_lo) w.sc = col[w.s_lo + lane];
        if (fl_ok) {
          w.zr = *reinterpret_cast<const float4*>(z + (int64_t)__builtin_amdgcn_readlane(gri, i) * ldz + 4 * fl);
          w.zs = *reinterpret_cast<const float4*>(z + (int64_t)__builtin_amdgcn_readlane(gsi, i) * ldz + 4 * fl);
        }
      }
      return w;
    };
    DecRow cur = fetch(0);
    for (int i = 0; i < cnt; ++i) {
      DecRow nxt = cur;
      if (i + 1 < cnt) nxt = fetch(i + 1);
      const int64_t r = base_r + i;
      st.ga = zero4;
      st.gb = zero4;
      st.zr = cur.zr;
      st.zs = cur.zs;
      if (cur.p_hi - cur.p_lo <= DEC_CHUNK) {
        dec_fwd_segment<LPR, GRAD>(z, ldz, zidx, col, cur.p_lo, cur.p_hi, cur.s_lo, cur.s_hi, cur.sr, cur.sc, true, cur.bc, cur.bg,
                                   lane, fl_ok, coef_pos, coef_neg, cstride, st);
      } else if constexpr (GRAD) {  // hub row: its pieces were computed by decoder_fused_fwd_long_kernel
        const int c_lo = cur.p_lo / DEC_CHUNK, c_hi = (cur.p_hi - 1) / DEC_CHUNK;
        if (lane < LPR && fl_ok) {
          for (int cc = c_lo; cc <= c_hi; ++cc) {  // shares in ascending position order
            const int slot = (cc == c_lo && cur.p_lo != cc * DEC_CHUNK) ? 1 : 0;
            const float* o = gpart + ((size_t)cc * 2 + slot) * 2 * (4 * LPR) + 4 * fl;
            const float4 t1 = *reinterpret_cast<const float4*>(o);
            const float4 t2 = *reinterpret_cast<const float4*>(o + 4 * LPR);
            st.ga.x += t1.x; st.ga.y += t1.y; st.ga.z += t1.z; st.ga.w += t1.w;
            st.gb.x += t2.x; st.gb.y += t2.y; st.gb.z += t2.z; st.gb.w += t2.w;
          }
        }
      }
      if constexpr (GRAD) {
        if (lane < LPR && fl_ok) {
          *reinterpret_cast<float4*>(grow_pos + r * ldg + 4 * fl) = st.ga;
          *reinterpret_cast<float4*>(grow_neg + (int64_t)cur.sr * ldg + 4 * fl) = st.gb;
        }
      }
      cur = nxt;
    }
    // loss partials per DRAW, not per wave: which wave ran a draw must not change the sums.  Slots are unique:
    // draws of one XCD are DEC_ROWS apart, and the +xcd keeps neighbouring ranges from sharing a slot.
    const float t0 = ceeg_wave_sum(st.s_pos), t1 = ceeg_wave_sum(st.s_neg), t2 = ceeg_wave_sum(st.s_cnt);
    if (lane == 0) {
      float* o = part + (size_t)(base_r / DEC_ROWS + xcd) * 3;
      o[0] = t0; o[1] = t1; o[2] = t2;
    }
  }
}

// ------------------------------------------------------------------ fused backward: both transposed passes in one
// Row n of the transposed CSR lists the entries q whose SOURCE is n (tcol q = the destination r, tperm q = the entry's
// position p in the dst-sorted CSR).  One kernel accumulates both column-side halves of the gradient
//     s1 = sum_q cp[p] z[r]          s2 = sum_q cn[p] z[sigma(r)]
// and finishes   dz[n] = gn*(grow_neg[n] + s2) + gp*(grow_pos[n] + s1).
// The (tcol, tperm) streams and the coefficient pair (interleaved: one sector per entry) are read once instead of
// once per pass and dz is written once.  Summation order and rounding are exactly those of the two-pass form over
// spmm.hip (lane group g adds entries g, g+G, ... ascending, fixed butterfly, long rows cut at global positions
// k*DEC_CHUNK with partials added in ascending order), so both forms give bitwise equal results.
struct DecBwdArgs {
  const int32_t *t_rowptr, *t_col, *t_perm, *zidx;
  const float *coef_pos, *coef_neg, *z, *grow_pos, *grow_neg, *gscale;
  float* dz;
  float* partial;        // [chunk][slot 0/1][sum 1/2][4*LPR]
  unsigned int* queue;
  int64_t ldz, ldg, lddz, N, sig_a, sig_b, sig_mod;
  int cstride, D;
};

struct DecBwdBatch {   // lane's entry of a 64-entry batch, ready to gather
  int g1, g2;
  float c1, c2;
};

__device__ __forceinline__ void dec_bwd_raw(const DecBwdArgs& a, int base, int p_hi, int lane, int& rc, int& rp) {
  rc = 0;
  rp = 0;
  if (base + lane < p_hi) {
    rc = a.t_col[base + lane];
    rp = a.t_perm[base + lane];
  }
}
__device__ __forceinline__ DecBwdBatch dec_bwd_resolve(const DecBwdArgs& a, int base, int p_hi, int lane, int rc, int rp) {
  DecBwdBatch b;
  b.g1 = 0; b.g2 = 0; b.c1 = 0.f; b.c2 = 0.f;
  if (base + lane < p_hi) {
    const int s = (int)ceeg_block_affine((int64_t)rc, a.sig_a, a.sig_b, a.sig_mod);
    b.g1 = a.zidx ? a.zidx[rc] : rc;
    b.g2 = a.zidx ? a.zidx[s] : s;
    b.c1 = a.coef_pos[(int64_t)rp * a.cstride];
    b.c2 = a.coef_neg[(int64_t)rp * a.cstride];
  }
  return b;
}

template <int LPR, int K>
__device__ __forceinline__ void dec_bwd_steps(const DecBwdArgs& a, const DecBwdBatch& b, int i0, int n, int lane, bool fl_ok,
                                              float4& s1, float4& s2) {
  constexpr int G = 64 / LPR;
  const int fl = lane % LPR, sub = lane / LPR;
  float4 x1[K], x2[K];
  float v1[K], v2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int j = (i0 + k) * G + sub;
    const int g1 = __shfl(b.g1, j & 63, 64), g2 = __shfl(b.g2, j & 63, 64);
    v1[k] = __shfl(b.c1, j & 63, 64);
    v2[k] = __shfl(b.c2, j & 63, 64);
    x1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    x2[k] = x1[k];
    if (j < n && fl_ok) {
      x1[k] = *reinterpret_cast<const float4*>(a.z + (int64_t)g1 * a.ldz + 4 * fl);
      x2[k] = *reinterpret_cast<const float4*>(a.z + (int64_t)g2 * a.ldz + 4 * fl);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if ((i0 + k) * G + sub >= n) continue;  // lanes past the batch end add nothing
    s1.x = __fmaf_rn(v1[k], x1[k].x, s1.x); s1.y = __fmaf_rn(v1[k], x1[k].y, s1.y);
    s1.z = __fmaf_rn(v1[k], x1[k].z, s1.z); s1.w = __fmaf_rn(v1[k], x1[k].w, s1.w);
    s2.x = __fmaf_rn(v2[k], x2[k].x, s2.x); s2.y = __fmaf_rn(v2[k], x2[k].y, s2.y);
    s2.z = __fmaf_rn(v2[k], x2[k].z, s2.z); s2.w = __fmaf_rn(v2[k], x2[k].w, s2.w);
  }
}

// both sums over positions [p_lo, p_hi); `first` holds the resolved first batch
template <int LPR>
__device__ __forceinline__ void dec_bwd_seg(const DecBwdArgs& a, int p_lo, int p_hi, int lane, bool fl_ok, DecBwdBatch b,
                                            float4& s1, float4& s2) {
  constexpr int G = 64 / LPR;
  s1 = make_float4(0.f, 0.f, 0.f, 0.f);
  s2 = s1;
  for (int base = p_lo; base < p_hi; base += 64) {
    const int n = p_hi - base < 64 ? p_hi - base : 64;
    if (base != p_lo) {
      int rc, rp;
      dec_bwd_raw(a, base, p_hi, lane, rc, rp);
      b = dec_bwd_resolve(a, base, p_hi, lane, rc, rp);
    }
    const int ns = (n + G - 1) / G;
    int i0 = 0;
    for (; i0 + 4 <= ns; i0 += 4) dec_bwd_steps<LPR, 4>(a, b, i0, n, lane, fl_ok, s1, s2);
    const int rem = ns - i0;
    if (rem & 2) { dec_bwd_steps<LPR, 2>(a, b, i0, n, lane, fl_ok, s1, s2); i0 += 2; }
    if (rem & 1) dec_bwd_steps<LPR, 1>(a, b, i0, n, lane, fl_ok, s1, s2);
  }
  if constexpr (G > 1) {
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) {
      s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64); s1.z += __shfl_xor(s1.z, o, 64); s1.w += __shfl_xor(s1.w, o, 64);
      s2.x += __shfl_xor(s2.x, o, 64); s2.y += __shfl_xor(s2.y, o, 64); s2.z += __shfl_xor(s2.z, o, 64); s2.w += __shfl_xor(s2.w, o, 64);
    }
  }
}

// pieces of long rows: wave w handles global positions [w*DEC_CHUNK, (w+1)*DEC_CHUNK)
template <int LPR>
__global__ __launch_bounds__(DEC_WAVES * 64) void dec_bwd_long_kernel(DecBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * DEC_WAVES + (threadIdx.x >> 6);
  const int nnz = a.t_rowptr[a.N];
  const int64_t p0l = w * DEC_CHUNK;
  if (p0l >= nnz) return;
  const int p0 = (int)p0l;
  const int p1 = p0 + DEC_CHUNK < nnz ? p0 + DEC_CHUNK : nnz;
  const int fl = lane % LPR;
  const bool fl_ok = 4 * fl < a.D;
  const int r0 = dec_row_of(a.t_rowptr, a.N, p0);
  const int r1 = dec_row_of(a.t_rowptr, a.N, p1 - 1);
  auto piece = [&](int lo, int hi, int slot) {
    int rc, rp;
    dec_bwd_raw(a, lo, hi, lane, rc, rp);
    const DecBwdBatch b = dec_bwd_resolve(a, lo, hi, lane, rc, rp);
    float4 s1, s2;
    dec_bwd_seg<LPR>(a, lo, hi, lane, fl_ok, b, s1, s2);
    if (lane < LPR && fl_ok) {
      float* o = a.partial + ((size_t)w * 2 + slot) * 2 * (4 * LPR) + 4 * fl;
      *reinterpret_cast<float4*>(o) = s1;
      *reinterpret_cast<float4*>(o + 4 * LPR) = s2;
    }
  };
  if (a.t_rowptr[r0 + 1] - a.t_rowptr[r0] > DEC_CHUNK) piece(p0, a.t_rowptr[r0 + 1] < p1 ? a.t_rowptr[r0 + 1] : p1, 0);
  if (r1 != r0 && a.t_rowptr[r1 + 1] - a.t_rowptr[r1] > DEC_CHUNK) piece(a.t_rowptr[r1], p1, 1);
}

// rows: scheduling as the forward kernel (per-XCD in-order queue, every wave draws DEC_ROWS consecutive rows).  Inside a
// draw the loads that feed later rows are issued BEFORE the current row's gathers - dependent half of row i+1 (its
// inputs landed an iteration ago), raw half of row i+2, then the gathers of row i - so one wait covers all three.
template <int LPR>
__global__ __launch_bounds__(DEC_WAVES * 64) void dec_bwd_rows_kernel(DecBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int fl = lane % LPR;
  const bool fl_ok = 4 * fl < a.D;
  const int xcd = blockIdx.x & 7;
  const int64_t N = a.N;
  const int64_t nnz = a.t_rowptr[N];
  const int64_t r_begin = xcd == 0 ? 0 : dec_row_lower_bound(a.t_rowptr, N, nnz * xcd / 8);
  const int64_t r_end = xcd == 7 ? N : dec_row_lower_bound(a.t_rowptr, N, nnz * (xcd + 1) / 8);
  unsigned int* q = a.queue + xcd * 32;
  const float gp = a.gscale[0], gn = a.gscale[1];
  unsigned int dv = 0;
  if (lane == 0) dv = atomicAdd(q, (unsigned)DEC_ROWS);
  for (;;) {
    const int64_t base_r = r_begin + (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)dv);
    if (base_r >= r_end) break;
    if (lane == 0) dv = atomicAdd(q, (unsigned)DEC_ROWS);
    const int cnt = r_end - base_r < DEC_ROWS ? (int)(r_end - base_r) : DEC_ROWS;
    int rpv = 0;
    if (lane <= cnt) rpv = a.t_rowptr[base_r + lane];
    auto lo_of = [&](int i) { return __builtin_amdgcn_readlane(rpv, i < cnt ? i : cnt); };
    auto hi_of = [&](int i) { return __builtin_amdgcn_readlane(rpv, i < cnt ? i + 1 : cnt); };
    auto short_row = [&](int i) { return hi_of(i) - lo_of(i) <= DEC_CHUNK; };
    // fill: row 0 resolved, row 1 raw
    int rc0, rp0, rc1 = 0, rp1 = 0;
    dec_bwd_raw(a, lo_of(0), short_row(0) ? hi_of(0) : lo_of(0), lane, rc0, rp0);
    if (cnt > 1) dec_bwd_raw(a, lo_of(1), short_row(1) ? hi_of(1) : lo_of(1), lane, rc1, rp1);
    DecBwdBatch cur = dec_bwd_resolve(a, lo_of(0), short_row(0) ? hi_of(0) : lo_of(0), lane, rc0, rp0);
    for (int i = 0; i < cnt; ++i) {
      const int p_lo = lo_of(i), p_hi = hi_of(i);
      DecBwdBatch nxt = cur;
      if (i + 1 < cnt) nxt = dec_bwd_resolve(a, lo_of(i + 1), short_row(i + 1) ? hi_of(i + 1) : lo_of(i + 1), lane, rc1, rp1);
      int rc2 = 0, rp2 = 0;
      if (i + 2 < cnt) dec_bwd_raw(a, lo_of(i + 2), short_row(i + 2) ? hi_of(i + 2) : lo_of(i + 2), lane, rc2, rp2);
      const int64_t r = base_r + i;
      float4 s1, s2;
      if (p_hi - p_lo <= DEC_CHUNK) {
        dec_bwd_seg<LPR>(a, p_lo, p_hi, lane, fl_ok, cur, s1, s2);
      } else {
        s1 = make_float4(0.f, 0.f, 0.f, 0.f);
        s2 = s1;
        const int c_lo = p_lo / DEC_CHUNK, c_hi = (p_hi - 1) / DEC_CHUNK;
        if (lane < LPR && fl_ok) {
          for (int c = c_lo; c <= c_hi; ++c) {  // pieces in ascending position order
            const int slot = (c == c_lo && p_lo != c * DEC_CHUNK) ? 1 : 0;
            const float* o = a.partial + ((size_t)c * 2 + slot) * 2 * (4 * LPR) + 4 * fl;
            const float4 t1 = *reinterpret_cast<const float4*>(o);
            const float4 t2 = *reinterpret_cast<const float4*>(o + 4 * LPR);
            s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
            s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
          }
        }
      }
      if (lane < LPR && fl_ok) {
        if (a.grow_pos) {
          const float4 A = *reinterpret_cast<const float4*>(a.grow_pos + r * a.ldg + 4 * fl);
          const float4 B = *reinterpret_cast<const float4*>(a.grow_neg + r * a.ldg + 4 * fl);
          s1.x += A.x; s1.y += A.y; s1.z += A.z; s1.w += A.w;
          s2.x += B.x; s2.y += B.y; s2.z += B.z; s2.w += B.w;
        }
        s1.x *= gp; s1.y *= gp; s1.z *= gp; s1.w *= gp;
        s2.x *= gn; s2.y *= gn; s2.z *= gn; s2.w *= gn;
        float4 o;  // (second pass) + (first pass), as the accumulating two-pass form rounds it
        o.x = s2.x + s1.x; o.y = s2.y + s1.y; o.z = s2.z + s1.z; o.w = s2.w + s1.w;
        *reinterpret_cast<float4*>(a.dz + r * a.lddz + 4 * fl) = o;
      }
      cur = nxt;
      rc1 = rc2;
      rp1 = rp2;
    }
  }
}

// Row kernel for D <= 128: every LANE GROUP (LPR lanes) owns its own row, so a wave has G = 64/LPR rows in flight instead
// of one (the one-row form is latency-bound: ~2 GB of traffic in 1.5 ms).  A group walks its row LPR entries at a
// time (metadata one entry per lane, broadcast inside the group), 4 dual gathers in flight, sums in entry order -
// a different (still fixed) summation order than the one-row form, so equal to it only to rounding.
template <int LPR>
__global__ __launch_bounds__(DEC_WAVES * 64) void dec_bwd_rows_grouped_kernel(DecBwdArgs a) {
  constexpr int G = 64 / LPR;
  static_assert(G >= 2, "grouped form needs at least two rows per wave");
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPR, grp = lane / LPR, gbase = lane - gl;
  const bool fl_ok = 4 * gl < a.D;
  const int xcd = blockIdx.x & 7;
  const int64_t N = a.N;
  const int64_t nnz = a.t_rowptr[N];
  const int64_t r_begin = xcd == 0 ? 0 : dec_row_lower_bound(a.t_rowptr, N, nnz * xcd / 8);
  const int64_t r_end = xcd == 7 ? N : dec_row_lower_bound(a.t_rowptr, N, nnz * (xcd + 1) / 8);
  unsigned int* q = a.queue + xcd * 32;
  const float gp = a.gscale[0], gn = a.gscale[1];
  constexpr int DRAW = DEC_ROWS * G;  // rows per wave draw
  unsigned int dv = 0;
  if (lane == 0) dv = atomicAdd(q, (unsigned)DRAW);
  for (;;) {
    const int64_t base_r = r_begin + (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)dv);
    if (base_r >= r_end) break;
    if (lane == 0) dv = atomicAdd(q, (unsigned)DRAW);
    const int cnt = r_end - base_r < DRAW ? (int)(r_end - base_r) : DRAW;
    for (int i0 = 0; i0 < cnt; i0 += 64) {   // row pointers 64 at a time (DRAW may exceed 63)
      const int have = cnt - i0 < 64 ? cnt - i0 : 64;
      int rlo = 0, rhi = 0;
      if (lane < have) { rlo = a.t_rowptr[base_r + i0 + lane]; rhi = a.t_rowptr[base_r + i0 + lane + 1]; }
      // bounds of this group's row in iteration i (empty past the batch); exchanges by ALL lanes: a masked-off source reads 0
      auto bounds = [&](int i, int& lo, int& hi) {
        const int ri = i + grp;
        lo = __shfl(rlo, ri & 63, 64);
        const int h = __shfl(rhi, ri & 63, 64);
        hi = (i < have && ri < have) ? h : lo;
      };
      // first LPR entries of a (short) row: raw = (tcol, tperm); resolved = gather rows + coefficient pair
      auto raw = [&](int lo, int hi, int& rc, int& rp) {
        rc = 0; rp = 0;
        if (hi - lo <= DEC_CHUNK && lo + gl < hi) { rc = a.t_col[lo + gl]; rp = a.t_perm[lo + gl]; }
      };
      auto resolve = [&](int lo, int hi, int rc, int rp, int& g1, int& g2, float& c1, float& c2) {
        g1 = 0; g2 = 0; c1 = 0.f; c2 = 0.f;
        if (hi - lo <= DEC_CHUNK && lo + gl < hi) {
          const int s = (int)ceeg_block_affine((int64_t)rc, a.sig_a, a.sig_b, a.sig_mod);
          g1 = a.zidx ? a.zidx[rc] : rc;
          g2 = a.zidx ? a.zidx[s] : s;
          c1 = a.coef_pos[(int64_t)rp * a.cstride];
          c2 = a.coef_neg[(int64_t)rp * a.cstride];
        }
      };
      // pipeline: iteration i gathers; the resolve loads of i+G and the raw loads of i+2G are issued first (one wait covers all)
      int lo0, hi0, lo1, hi1, lo2, hi2;
      bounds(0, lo0, hi0);
      bounds(G, lo1, hi1);
      int rc0, rp0, rc1, rp1;
      raw(lo0, hi0, rc0, rp0);
      raw(lo1, hi1, rc1, rp1);
      int g1, g2;
      float c1, c2;
      resolve(lo0, hi0, rc0, rp0, g1, g2, c1, c2);
      for (int i = 0; i < have; i += G) {
        int g1n, g2n;
        float c1n, c2n;
        resolve(lo1, hi1, rc1, rp1, g1n, g2n, c1n, c2n);
        bounds(i + 2 * G, lo2, hi2);
        int rc2, rp2;
        raw(lo2, hi2, rc2, rp2);
        const int p_lo = lo0, p_hi = hi0;
        const bool live = i + grp < have;
        const int64_t r = base_r + i0 + i + grp;
        float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
        if (p_hi - p_lo <= DEC_CHUNK) {
          for (int base = p_lo; __any(base < p_hi); base += LPR) {
            const int n = p_hi - base < LPR ? (p_hi - base > 0 ? p_hi - base : 0) : LPR;
            if (base != p_lo) {   // rows longer than LPR entries: later batches are fetched in line (wave-uniform branch per lane group)
              int rc, rp;
              raw(base, p_hi, rc, rp);
              resolve(base, p_hi, rc, rp, g1, g2, c1, c2);
            }
            for (int j0 = 0; __any(j0 < n); j0 += 4) {
              float4 x1[4], x2[4];
              float v1[4], v2[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int src = gbase + ((j0 + k) & (LPR - 1));
                const int e1 = __shfl(g1, src, 64), e2 = __shfl(g2, src, 64);
                v1[k] = __shfl(c1, src, 64);
                v2[k] = __shfl(c2, src, 64);
                x1[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                x2[k] = x1[k];
                if (j0 + k < n && fl_ok) {
                  x1[k] = *reinterpret_cast<const float4*>(a.z + (int64_t)e1 * a.ldz + 4 * gl);
                  x2[k] = *reinterpret_cast<const float4*>(a.z + (int64_t)e2 * a.ldz + 4 * gl);
                }
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (j0 + k < n) {
                  s1.x = __fmaf_rn(v1[k], x1[k].x, s1.x); s1.y = __fmaf_rn(v1[k], x1[k].y, s1.y);
                  s1.z = __fmaf_rn(v1[k], x1[k].z, s1.z); s1.w = __fmaf_rn(v1[k], x1[k].w, s1.w);
                  s2.x = __fmaf_rn(v2[k], x2[k].x, s2.x); s2.y = __fmaf_rn(v2[k], x2[k].y, s2.y);
                  s2.z = __fmaf_rn(v2[k], x2[k].z, s2.z); s2.w = __fmaf_rn(v2[k], x2[k].w, s2.w);
                }
              }
            }
          }
        } else if (fl_ok) {  // hub row: pieces from dec_bwd_long_kernel<LPR>, ascending
          const int c_lo = p_lo / DEC_CHUNK, c_hi = (p_hi - 1) / DEC_CHUNK;
          for (int c = c_lo; c <= c_hi; ++c) {
            const int slot = (c == c_lo && p_lo != c * DEC_CHUNK) ? 1 : 0;
            const float* o = a.partial + ((size_t)c * 2 + slot) * 2 * (4 * LPR) + 4 * gl;
            const float4 t1 = *reinterpret_cast<const float4*>(o);
            const float4 t2 = *reinterpret_cast<const float4*>(o + 4 * LPR);
            s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
            s2.x += t2.x; s2.y += t2.y; s2.z += t2.z; s2.w += t2.w;
          }
        }
        if (live && fl_ok) {
          if (a.grow_pos) {
            const float4 A = *reinterpret_cast<const float4*>(a.grow_pos + r * a.ldg + 4 * gl);
            const float4 B = *reinterpret_cast<const float4*>(a.grow_neg + r * a.ldg + 4 * gl);
            s1.x += A.x; s1.y += A.y; s1.z += A.z; s1.w += A.w;
            s2.x += B.x; s2.y += B.y; s2.z += B.z; s2.w += B.w;
          }
          s1.x *= gp; s1.y *= gp; s1.z *= gp; s1.w *= gp;
          s2.x *= gn; s2.y *= gn; s2.z *= gn; s2.w *= gn;
          float4 o;
          o.x = s2.x + s1.x; o.y = s2.y + s1.y; o.z = s2.z + s1.z; o.w = s2.w + s1.w;
          *reinterpret_cast<float4*>(a.dz + r * a.lddz + 4 * gl) = o;
        }
        lo0 = lo1; hi0 = hi1; lo1 = lo2; hi1 = hi2;
        g1 = g1n; g2 = g2n; c1 = c1n; c2 = c2n;
        rc1 = rc2; rp1 = rp2;
      }
    }
  }
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

extern "C" int ceeg_edge_bce_fwd(const float* z, int64_t ldz, const int32_t* zidx, const int32_t* i, const int32_t* j, int64_t E,
                                 int D, int negative, float* coef, float* prob, float* loss_sum,
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
    hipLaunchKernelGGL(edge_bce_fwd_kernel<L>, dim3(nblk), dim3(DEC_WAVES * 64), 0, st, z, ldz, zidx, i, j, E, D, negative, coef, prob, part); \
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
  hipLaunchKernelGGL(dec_finalize_kernel, dim3(1), dim3(1024), 0, st, part, nblk, 2, loss_sum);  // [sum of terms, pairs counted]
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" size_t ceeg_decoder_fused_ws_bytes(int64_t N, int64_t nnz, int D) {
  if (N <= 0 || nnz < 0 || D <= 0) return 0;
  const size_t nchunk = (size_t)((nnz + DEC_CHUNK - 1) / DEC_CHUNK);
  // forward: queue | loss partials (one slot per draw + two per chunk of hub rows) | hub-row shares of the row-side sums
  const size_t a = DEC_QUEUE_BYTES + ceeg_ws_part(((size_t)(N / DEC_ROWS + 16) + 2 * (nchunk + 1)) * 3, 4) +
                   ceeg_ws_part((nchunk + 1) * 2 * 2 * 4 * (size_t)dec_lpr(D), 4);
  const size_t a0 = ceeg_ws_part((size_t)DEC_MAX_BLOCKS * 3, 4);
  const size_t c0 = DEC_QUEUE_BYTES + ceeg_ws_part((nchunk + 1) * 2 * 2 * 4 * (size_t)dec_lpr(D), 4);  // fused backward: queue + long-row partials
  const size_t b = ceeg_spmm_ws_bytes(N, nnz, D);
  const size_t m = (a > b ? a : b) > a0 ? (a > b ? a : b) : a0;
  return m > c0 ? m : c0;
}

extern "C" int ceeg_decoder_fused_fwd(const float* z, int64_t ldz, const int32_t* zidx, const int32_t* rowptr, const int32_t* col,
                                      int64_t N, int64_t nnz, int D, int64_t sig_a, int64_t sig_b, int64_t sig_mod,
                                      float* coef_pos, float* coef_neg, int coef_stride, float* grow_pos, float* grow_neg, int64_t ldg,
                                      float* sums, int64_t max_row_nnz, void* ws, size_t ws_bytes, void* stream) {
  if (coef_stride != 1 && coef_stride != 2) return CEEG_EINVAL;
  if ((grow_pos == nullptr) != (grow_neg == nullptr)) return CEEG_EINVAL;
  if (grow_pos && (ldg < D || (ldg % 4) != 0 || !ceeg_aligned(grow_pos, 16) || !ceeg_aligned(grow_neg, 16))) return CEEG_EINVAL;
  if (!z || !rowptr || !sums || N <= 0 || nnz < 0 || D <= 0 || (nnz > 0 && (!col || !coef_pos || !coef_neg))) return CEEG_EINVAL;
  if ((D % 4) != 0 || D > 256 || (ldz % 4) != 0 || ldz < D || N >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (sig_mod <= 0 || (N % sig_mod) != 0 || sig_a <= 0 || sig_a >= sig_mod + (sig_mod == 1) || sig_b < 0 || sig_b >= sig_mod) return CEEG_EINVAL;
  if (!ceeg_aligned(z, 16)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_decoder_fused_ws_bytes(N, nnz, D) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(ws);
  unsigned int* queue = static_cast<unsigned int*>(ws);  // queue counters, then one loss-partial slot per draw (unused slots stay 0)
  part = reinterpret_cast<float*>(static_cast<char*>(ws) + DEC_QUEUE_BYTES);
  const int64_t nchunk = (nnz + DEC_CHUNK - 1) / DEC_CHUNK;
  const int64_t lslot0 = N / DEC_ROWS + 16;                       // first loss slot of the hub-row pieces
  const int nslots = (int)(lslot0 + 2 * (nchunk + 1));
  float* gpart = reinterpret_cast<float*>(static_cast<char*>(ws) + DEC_QUEUE_BYTES + ceeg_ws_part((size_t)nslots * 3, 4));
  const int lpr = dec_lpr(D);
  const bool has_long = nchunk > 0 && !(max_row_nnz >= 0 && max_row_nnz <= DEC_CHUNK);  // hint from the graph build skips the pass
  const unsigned lblk = (unsigned)((nchunk + DEC_WAVES - 1) / DEC_WAVES);
  const int nsum = has_long ? nslots : (int)lslot0;               // slots that can hold anything
  if (hipMemsetAsync(ws, 0, DEC_QUEUE_BYTES + (size_t)nsum * 3 * 4, st) != hipSuccess) return CEEG_ELAUNCH;
  int64_t per_xcd = (N + 8 * DEC_WAVES * DEC_ROWS - 1) / (8 * DEC_WAVES * DEC_ROWS);  // enough blocks for one draw per wave ...
  if (per_xcd > DEC_BLOCKS_PER_XCD) per_xcd = DEC_BLOCKS_PER_XCD;                     // ... capped at a persistent grid
  const int nblk = (int)(8 * per_xcd);
#define CEEG_DEC_CASE(L)                                                                                                              \
  case L:                                                                                                                             \
    if (grow_pos) {                                                                                                                   \
      if (has_long)                                                                                                                   \
        hipLaunchKernelGGL((decoder_fused_fwd_long_kernel<L, true>), dim3(lblk), dim3(DEC_WAVES * 64), 0, st, z, ldz, zidx, rowptr, col, N, D, \
                           sig_a, sig_b, sig_mod, coef_pos, coef_neg, coef_stride, gpart, part, lslot0);                                \
      hipLaunchKernelGGL((decoder_fused_fwd_kernel<L, true>), dim3(nblk), dim3(DEC_WAVES * 64), 0, st, z, ldz, zidx, rowptr, col, N, D, sig_a, \
                         sig_b, sig_mod, coef_pos, coef_neg, coef_stride, grow_pos, grow_neg, ldg, queue, gpart, part);               \
    } else {                                                                                                                          \
      if (has_long)                                                                                                                   \
        hipLaunchKernelGGL((decoder_fused_fwd_long_kernel<L, false>), dim3(lblk), dim3(DEC_WAVES * 64), 0, st, z, ldz, zidx, rowptr, col, N, D, \
                           sig_a, sig_b, sig_mod, coef_pos, coef_neg, coef_stride, gpart, part, lslot0);                                \
      hipLaunchKernelGGL((decoder_fused_fwd_kernel<L, false>), dim3(nblk), dim3(DEC_WAVES * 64), 0, st, z, ldz, zidx, rowptr, col, N, D, sig_a, \
                         sig_b, sig_mod, coef_pos, coef_neg, coef_stride, grow_pos, grow_neg, ldg, queue, gpart, part);               \
    }                                                                                                                                 \
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
  hipLaunchKernelGGL(dec_finalize_kernel, dim3(1), dim3(1024), 0, st, part, nsum, 3, sums);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_decoder_fused_bwd(const float* z, int64_t ldz, const int32_t* zidx, const int32_t* rowptr, const int32_t* col,
                                      const int32_t* t_rowptr, const int32_t* t_col, const int32_t* t_perm,
                                      int64_t N, int64_t nnz, int D, int64_t sig_a, int64_t sig_b, int64_t sig_ainv, int64_t sig_mod,
                                      const float* coef_pos, const float* coef_neg, int coef_stride, const float* grow_pos,
                                      const float* grow_neg, int64_t ldg, const float* gscale, float* dz, int64_t lddz, int64_t max_row_nnz, int64_t max_trow_nnz,
                                      void* ws, size_t ws_bytes, void* stream) {
  if (!z || !rowptr || !t_rowptr || !gscale || !dz || N <= 0 || nnz < 0) return CEEG_EINVAL;
  if (nnz > 0 && (!col || !t_col || !t_perm || !coef_pos || !coef_neg)) return CEEG_EINVAL;
  if (sig_mod <= 0 || (N % sig_mod) != 0 || sig_a <= 0 || sig_b < 0 || sig_b >= sig_mod || sig_ainv <= 0) return CEEG_EINVAL;
  if ((uint64_t)((unsigned __int128)(uint64_t)sig_a * (uint64_t)sig_ainv % (uint64_t)sig_mod) != (uint64_t)(1 % sig_mod)) return CEEG_EINVAL;
  if (ws_bytes < ceeg_decoder_fused_ws_bytes(N, nnz, D)) return CEEG_EWORKSPACE;
  if ((grow_pos == nullptr) != (grow_neg == nullptr)) return CEEG_EINVAL;
  if (grow_pos && (ldg < D || (ldg % 4) != 0 || !ceeg_aligned(grow_pos, 16) || !ceeg_aligned(grow_neg, 16))) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc;
  if (coef_stride != 1 && coef_stride != 2) return CEEG_EINVAL;
  const char* two_pass_env = getenv("CEEG_DEC_BWD_TWO_PASS");  // checker's switch: the same sums as two SpMM passes
  if (grow_pos && two_pass_env && two_pass_env[0] == '1') {
    CeegSpmmEx t2; t2.mod = sig_mod; t2.scale = gscale + 0; t2.max_row_nnz = max_trow_nnz; t2.addend = grow_pos; t2.ld_add = ldg;
    t2.xidx = zidx; t2.val_stride = coef_stride;
    rc = ceeg_spmm_launch(t_rowptr, t_col, coef_pos, t_perm, z, ldz, N, nnz, D, nullptr, 0, 0, dz, lddz, ws, ws_bytes, st, t2);
    if (rc != CEEG_OK) return rc;
    CeegSpmmEx t4 = t2; t4.scale = gscale + 1; t4.col_a = sig_a; t4.col_b = sig_b; t4.addend = grow_neg;
    return ceeg_spmm_launch(t_rowptr, t_col, coef_neg, t_perm, z, ldz, N, nnz, D, nullptr, 0, 1, dz, lddz, ws, ws_bytes, st, t4);
  }
  if (grow_pos) {
    // the forward already holds the row-side sums (terms (1) and (3) below); ONE kernel pair adds both transposed terms:
    //   dz[n] = gn * (grow_neg[n] + sum_{q in trow n} cn[tperm q] z[sigma(tcol q)]) + gp * (grow_pos[n] + sum_q cp[tperm q] z[tcol q])
    if ((D % 4) != 0 || D > 256 || (ldz % 4) != 0 || ldz < D || (lddz % 4) != 0 || lddz < D || N >= ((int64_t)1 << 31) || nnz >= ((int64_t)1 << 31))
      return CEEG_EUNSUPPORTED;
    if (!ceeg_aligned(z, 16) || !ceeg_aligned(dz, 16) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EINVAL;
    const int lpr = dec_lpr(D);
    DecBwdArgs a;
    a.t_rowptr = t_rowptr; a.t_col = t_col; a.t_perm = t_perm; a.zidx = zidx;
    a.coef_pos = coef_pos; a.coef_neg = coef_neg; a.z = z; a.grow_pos = grow_pos; a.grow_neg = grow_neg; a.gscale = gscale;
    a.dz = dz;
    a.queue = static_cast<unsigned int*>(ws);
    a.partial = reinterpret_cast<float*>(static_cast<char*>(ws) + DEC_QUEUE_BYTES);
    a.ldz = ldz; a.ldg = ldg; a.lddz = lddz; a.N = N; a.sig_a = sig_a; a.sig_b = sig_b; a.sig_mod = sig_mod;
    a.cstride = coef_stride; a.D = D;
    if (hipMemsetAsync(a.queue, 0, DEC_QUEUE_BYTES, st) != hipSuccess) return CEEG_ELAUNCH;
    const int64_t nchunk = (nnz + DEC_CHUNK - 1) / DEC_CHUNK;
    const bool has_long = nchunk > 0 && !(max_trow_nnz >= 0 && max_trow_nnz <= DEC_CHUNK);
    int64_t per_xcd = (N + 8 * DEC_WAVES * DEC_ROWS - 1) / (8 * DEC_WAVES * DEC_ROWS);
    if (per_xcd > DEC_BLOCKS_PER_XCD) per_xcd = DEC_BLOCKS_PER_XCD;
    const char* genv = getenv("CEEG_DEC_BWD_GROUPED");
    const bool grouped = !(genv && genv[0] == '0');
#define CEEG_DECB_CASE(L)                                                                                                         \
  case L:                                                                                                                         \
    if (has_long)                                                                                                                 \
      hipLaunchKernelGGL(dec_bwd_long_kernel<L>, dim3((unsigned)((nchunk + DEC_WAVES - 1) / DEC_WAVES)), dim3(DEC_WAVES * 64), 0, st, a); \
    if constexpr (L <= 32) {                                                                                                      \
      if (grouped) hipLaunchKernelGGL(dec_bwd_rows_grouped_kernel<L>, dim3((unsigned)(8 * per_xcd)), dim3(DEC_WAVES * 64), 0, st, a); \
      else hipLaunchKernelGGL(dec_bwd_rows_kernel<L>, dim3((unsigned)(8 * per_xcd)), dim3(DEC_WAVES * 64), 0, st, a);               \
    } else {                                                                                                                      \
      hipLaunchKernelGGL(dec_bwd_rows_kernel<L>, dim3((unsigned)(8 * per_xcd)), dim3(DEC_WAVES * 64), 0, st, a);                   \
    }                                                                                                                             \
    break;
    switch (lpr) {
      CEEG_DECB_CASE(1)
      CEEG_DECB_CASE(2)
      CEEG_DECB_CASE(4)
      CEEG_DECB_CASE(8)
      CEEG_DECB_CASE(16)
      CEEG_DECB_CASE(32)
      CEEG_DECB_CASE(64)
      default: return CEEG_EUNSUPPORTED;
    }
#undef CEEG_DECB_CASE
    CEEG_CHECK_LAUNCH();
    return CEEG_OK;
  }
  // (1) n as dst of positives:            dz[n]  = gp * sum_{p in row n} cp[p] z[col p]
  CeegSpmmEx e1; e1.mod = sig_mod; e1.scale = gscale + 0; e1.max_row_nnz = max_row_nnz; e1.xidx = zidx; e1.val_stride = coef_stride;
  CeegSpmmEx e2 = e1; e2.max_row_nnz = max_trow_nnz;
  rc = ceeg_spmm_launch(rowptr, col, coef_pos, nullptr, z, ldz, N, nnz, D, nullptr, 0, 0, dz, lddz, ws, ws_bytes, st, e1);
  if (rc != CEEG_OK) return rc;
  // (2) n as src of positives:            dz[n] += gp * sum_{q in trow n} cp[tperm q] z[tcol q]
  rc = ceeg_spmm_launch(t_rowptr, t_col, coef_pos, t_perm, z, ldz, N, nnz, D, nullptr, 0, 1, dz, lddz, ws, ws_bytes, st, e2);
  if (rc != CEEG_OK) return rc;
  // (3) n = sigma(r) as negative endpoint: dz[sigma(r)] += gn * sum_{p in row r} cn[p] z[col p]
  CeegSpmmEx e3; e3.mod = sig_mod; e3.scale = gscale + 1; e3.row_a = sig_a; e3.row_b = sig_b; e3.max_row_nnz = max_row_nnz; e3.xidx = zidx; e3.val_stride = coef_stride;
  rc = ceeg_spmm_launch(rowptr, col, coef_neg, nullptr, z, ldz, N, nnz, D, nullptr, 0, 1, dz, lddz, ws, ws_bytes, st, e3);
  if (rc != CEEG_OK) return rc;
  // (4) n as src of negatives:            dz[n] += gn * sum_{q in trow n} cn[tperm q] z[sigma(tcol q)]
  CeegSpmmEx e4; e4.mod = sig_mod; e4.scale = gscale + 1; e4.col_a = sig_a; e4.col_b = sig_b; e4.max_row_nnz = max_trow_nnz; e4.xidx = zidx; e4.val_stride = coef_stride;
  rc = ceeg_spmm_launch(t_rowptr, t_col, coef_neg, t_perm, z, ldz, N, nnz, D, nullptr, 0, 1, dz, lddz, ws, ws_bytes, st, e4);
  return rc;
}
