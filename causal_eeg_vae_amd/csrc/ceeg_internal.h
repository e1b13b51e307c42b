// Shared device/host helpers for libceeg.so (gfx950 / CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ceeg.h"

#define CEEG_WAVE 64

#define CEEG_CHECK_LAUNCH()                          \
  do {                                               \
    if (hipGetLastError() != hipSuccess) return CEEG_ELAUNCH; \
  } while (0)

static inline size_t ceeg_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline bool ceeg_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Bump allocator over the caller's workspace (256-byte aligned carves). Host-side only.
struct CeegWs {
  char* base;
  size_t off, cap;
  bool ok;
  CeegWs(void* p, size_t bytes) : base(static_cast<char*>(p)), off(0), cap(bytes), ok(true) {}
  template <class T>
  T* take(size_t n) {
    size_t need = ceeg_align_up(n * sizeof(T), 256);
    if (base == nullptr || off + need > cap) { ok = false; return nullptr; }
    T* r = reinterpret_cast<T*>(base + off);
    off += need;
    return r;
  }
};
static inline size_t ceeg_ws_part(size_t n, size_t elem) { return ceeg_align_up(n * elem, 256); }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float ceeg_bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even fp32 -> bf16 (NaN kept a NaN)
__device__ __forceinline__ unsigned short ceeg_f32_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

__device__ __forceinline__ float ceeg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;  // fixed butterfly order: identical in every lane, reproducible
}
__device__ __forceinline__ double ceeg_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Philox4x32-10, restated bit for bit in oracle/ceeg_oracle.c
#define CEEG_TAG_EPS 0x45505300u
#define CEEG_TAG_NEG 0x4E454700u
__device__ __forceinline__ uint4 ceeg_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
// two uint32 -> two N(0,1) (Box-Muller); u = (x+1)*2^-32 in (0,1]
__device__ __forceinline__ float2 ceeg_box_muller(uint32_t x0, uint32_t x1) {
  float u0 = ((float)x0 + 1.0f) * 2.3283064365386963e-10f;
  float u1 = ((float)x1 + 1.0f) * 2.3283064365386963e-10f;
  u0 = fminf(u0, 1.0f);
  float r = sqrtf(-2.0f * logf(u0));
  float t = 6.283185307179586f * u1;
  float s, c;
  sincosf(t, &s, &c);
  return make_float2(r * c, r * s);
}

__device__ __forceinline__ bool csr_has(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int32_t d, int32_t s) {
  int lo = rowptr[d];
  const int end = rowptr[d + 1];
  int hi = end;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < s) lo = mid + 1; else hi = mid;
  }
  return lo < end && col[lo] == s;
}

// Edge-set prefilter of the sampler: a blocked Bloom filter - every (src, dst) sets FOUR bits of ONE 32-bit word, so a probe is still
// one random read, and a word that lacks any of the four bits proves "not an edge" without the row-pointer pair and the binary
// search (exact: full matches still go through the search).  At >= 16 bits per edge (two edges per word) about 0.4 % of the
// non-edges match, against 6 % for one bit per edge: the dependent search chain that nearly every wave waited for becomes rare.
struct NegProbe {
  uint32_t word, mask;
};
__device__ __forceinline__ NegProbe neg_filter_probe(int32_t s, int32_t d, unsigned bits) {
  uint64_t x = (((uint64_t)(uint32_t)d << 32) | (uint64_t)(uint32_t)s) * 0x9E3779B97F4A7C15ull;
  x ^= x >> 29;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 32;
  NegProbe p;
  p.word = (uint32_t)(x >> (64 - (bits - 5)));
  const uint32_t m = (uint32_t)x;
  p.mask = (1u << (m & 31)) | (1u << ((m >> 5) & 31)) | (1u << ((m >> 10) & 31)) | (1u << ((m >> 15) & 31));
  return p;
}

// The k-th uniform negative sample (include/ceeg.h, ceeg_neg_sample_uniform; restated in oracle/ceeg_oracle.c): up to 16 Philox
// candidates, rejected when src == dst or src -> dst is an edge; (-1, -1) when all collide.  One definition for the stand-alone
// sampler (graph.hip) and the sampler fused into the pair-list forward (decoder_codes.hip): the samples are the same bit for bit.
template <bool FILTER>
__device__ __forceinline__ void ceeg_neg_draw(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N, uint64_t seed,
                                              int64_t k, const uint32_t* __restrict__ filter, unsigned filter_bits, int32_t& s, int32_t& d) {
  s = -1;
  d = -1;
  for (uint32_t a = 0; a < 16; ++a) {  // bounded: every lane leaves after at most 16 attempts
    const uint4 x = ceeg_philox(seed, (uint32_t)k, (uint32_t)((uint64_t)k >> 32), CEEG_TAG_NEG, a);
    const int32_t cs = (int32_t)(((uint64_t)x.x * (uint64_t)N) >> 32);
    const int32_t cd = (int32_t)(((uint64_t)x.y * (uint64_t)N) >> 32);
    if (cs == cd) continue;
    if (FILTER) {
      const NegProbe pr = neg_filter_probe(cs, cd, filter_bits);
      if ((filter[pr.word] & pr.mask) != pr.mask) {  // a clear bit: not an edge
        s = cs;
        d = cd;
        break;
      }
    }
    if (csr_has(rowptr, col, cd, cs)) continue;
    s = cs;
    d = cd;
    break;
  }
}

// U samples (k[0..U)) drawn in LOCKSTEP by one thread: the same candidates, tests and results as U calls of ceeg_neg_draw, but
// the dependent loads of the U samples (filter word -> row pointers -> binary-search probes) are issued side by side, so a
// thread keeps U random reads in flight instead of one (the sampler is latency-bound: ~6 % of the candidates hit a set filter
// bit and take a row search, and with 64 lanes nearly every wave waits for one of them).  live[u] = false skips sample u.
template <bool FILTER, int U>
__device__ __forceinline__ void ceeg_neg_draw_multi(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, int64_t N,
                                                    uint64_t seed, const int64_t (&k)[U], const bool (&live)[U],
                                                    const uint32_t* __restrict__ filter, unsigned filter_bits, int32_t (&s)[U],
                                                    int32_t (&d)[U]) {
  bool open[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { s[u] = -1; d[u] = -1; open[u] = live[u]; }
  for (uint32_t a = 0; a < 16; ++a) {  // bounded: every lane leaves after at most 16 attempts
    bool any_open = false;
#pragma unroll
    for (int u = 0; u < U; ++u) any_open |= open[u];
    if (!any_open) break;
    int32_t cs[U], cd[U];
    uint32_t fw[U], fm[U];
    bool search[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // candidates + filter words, all U loads issued before any is used
      const uint4 x = ceeg_philox(seed, (uint32_t)k[u], (uint32_t)((uint64_t)k[u] >> 32), CEEG_TAG_NEG, a);
      cs[u] = (int32_t)(((uint64_t)x.x * (uint64_t)N) >> 32);
      cd[u] = (int32_t)(((uint64_t)x.y * (uint64_t)N) >> 32);
      search[u] = open[u] && cs[u] != cd[u];
      fw[u] = ~0u;
      fm[u] = 1u;
      if (FILTER && search[u]) {
        const NegProbe pr = neg_filter_probe(cs[u], cd[u], filter_bits);
        fm[u] = pr.mask;
        fw[u] = __builtin_nontemporal_load(filter + pr.word);  // single-use lines of a 32 MiB bitmap: keep L2 for the codes
      }
    }
    int lo[U], hi[U], end[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (search[u] && (fw[u] & fm[u]) != fm[u]) {  // a clear bit: not an edge
        s[u] = cs[u]; d[u] = cd[u];
        open[u] = false;
        search[u] = false;
      }
      lo[u] = 0; hi[u] = 0; end[u] = 0;
      if (search[u]) { lo[u] = rowptr[cd[u]]; end[u] = rowptr[cd[u] + 1]; hi[u] = end[u]; }
    }
    for (;;) {  // the U binary searches of csr_has, one probe of each per round
      bool more = false;
#pragma unroll
      for (int u = 0; u < U; ++u) more |= search[u] && lo[u] < hi[u];
      if (!more) break;
      int c[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        c[u] = 0;
        if (search[u] && lo[u] < hi[u]) c[u] = col[(lo[u] + hi[u]) >> 1];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (search[u] && lo[u] < hi[u]) {
          const int mid = (lo[u] + hi[u]) >> 1;
          if (c[u] < cs[u]) lo[u] = mid + 1; else hi[u] = mid;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (search[u]) {
        const bool is_edge = lo[u] < end[u] && col[lo[u]] == cs[u];
        if (!is_edge) { s[u] = cs[u]; d[u] = cd[u]; open[u] = false; }
      }
    }
  }
}

// ---------------------------------------------------------------- internal cross-file launchers
// Extra addressing for the SpMM used by the fused decoder backward: affine (bijective) maps on the
// gathered column index and on the output row, and a device-side scale applied to the row sum.
// sigma(x) = (x - x % mod) + (a * (x % mod) + b) % mod: an affine permutation inside every block of `mod` consecutive ids
// (mod = N: one permutation of all nodes; mod = nodes per graph: negatives stay inside their graph of a collated batch)
__host__ __device__ __forceinline__ int64_t ceeg_block_affine(int64_t x, int64_t a, int64_t b, int64_t mod) {
  const uint64_t l = (uint64_t)x % (uint64_t)mod;
  return x - (int64_t)l + (int64_t)(((uint64_t)a * l + (uint64_t)b) % (uint64_t)mod);
}
struct CeegSpmmEx {
  int64_t row_a = 1, row_b = 0, col_a = 1, col_b = 0, mod = 1;  // id maps through ceeg_block_affine
  int64_t max_row_nnz = -1;  // longest row if the caller knows it (<0: unknown -> the long-row pass always runs)
  const float* scale = nullptr;
  const float* addend = nullptr;  // [N, ld_add]: added to row r's sum before the scale (row index BEFORE the row map)
  int64_t ld_add = 0;
  int64_t val_stride = 1;         // element stride of val[] (2: one half of an interleaved pair)
  const int32_t* long_items = nullptr;  // (lo, hi, chunk*2+slot, row) pieces of the rows longer than the chunk (non-NULL: used INSTEAD of the scan)
  int64_t n_long_items = 0;
  int nt = 0;                     // bit 0: non-temporal stores of out; bit 1: non-temporal loads of the col / val streams
  const int32_t* xidx = nullptr;  // row indirection for X (after the column map): gather X[xidx[c]] - X is then a small table
  const int2* colval = nullptr;   // (col, val bits) interleaved: one 8-byte stream instead of two 4-byte ones (col / val arguments then unused)
  int bf16 = 0;                   // X, mask and out hold bf16 (sums fp32); no accumulate / addend / table in this form
  const float* mask = nullptr;    // [N, ld_mask]: the finished row is multiplied by (mask[r, f] > 0) - a ReLU's backward mask, applied
  int64_t ld_mask = 0;            // to the gradient the pass produces (the mask row is fetched before the gathers, off the critical path)
};
int ceeg_spmm_launch(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                     const void* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                     int accumulate, void* out, int64_t ldo, void* ws, size_t ws_bytes, hipStream_t st,
                     const CeegSpmmEx& ex);
