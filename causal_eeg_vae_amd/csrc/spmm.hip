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
#include <stdlib.h>
#include "ceeg_internal.h"

namespace {

constexpr int SPMM_CHUNK = 256;  // nonzeros per long-row piece
constexpr int SPMM_WAVES = 4;    // rows (waves) per workgroup
constexpr int SPMM_BLOCKS_PER_XCD = 256;  // persistent grid: 32 CUs x 8 workgroups per XCD
constexpr int SPMM_GRAB = 4;     // consecutive rows per queue draw (one per wave: keeps the XCD window of live rows tight)
constexpr int SPMM_SUBQ = 4;      // interleaved sub-queues per XCD (spreads the returning atomics over 4 words)
constexpr size_t SPMM_QUEUE_BYTES = 8 * SPMM_SUBQ * 128;  // 8 counters on 128-byte lines of their own, one memset per pass

// K consecutive steps of a batch: step i covers entries j = i*G + g of the batch (g = lane group).
// All K gathers are issued before the first FMA (K x 1 KiB per wave in flight), the FMAs then run in
// ascending entry order, so the summation order does not depend on K.
// Row pieces: every lane owns 16 BYTES of a feature row - 4 fp32 features, or 8 features of bf16 storage (BF = true: the gathered
// matrix, the ReLU mask and the output hold bf16, sums stay fp32; a row is half the bytes AND half the lanes, so a wave
// instruction gathers twice as many rows - the opt-in reduced-precision step).  SpAcc<BF> = the lane's 4 / 8 running sums.
template <bool BF>
struct SpAcc {
  float4 a;
};
template <>
struct SpAcc<true> {
  float4 a, b;
};
template <bool BF>
constexpr int sp_epl() { return BF ? 8 : 4; }  // elements per lane
template <bool BF>
__device__ __forceinline__ SpAcc<BF> sp_zero() {
  SpAcc<BF> r;
  r.a = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (BF) r.b = r.a;
  return r;
}
template <bool BF>
__device__ __forceinline__ uint4 sp_load_raw(const void* __restrict__ base, int64_t elem) {
  return *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + elem * (BF ? 2 : 4));
}
template <bool BF>
__device__ __forceinline__ SpAcc<BF> sp_unpack(const uint4& u) {
  SpAcc<BF> r;
  if constexpr (BF) {
    r.a = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    r.b = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
  } else {
    r.a = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
  }
  return r;
}
__device__ __forceinline__ void sp_fma4(float v, const float4& x, float4& acc) {
  acc.x = __fmaf_rn(v, x.x, acc.x);
  acc.y = __fmaf_rn(v, x.y, acc.y);
  acc.z = __fmaf_rn(v, x.z, acc.z);
  acc.w = __fmaf_rn(v, x.w, acc.w);
}
template <bool BF>
__device__ __forceinline__ void sp_fma(float v, const uint4& raw, SpAcc<BF>& acc) {
  const SpAcc<BF> x = sp_unpack<BF>(raw);
  sp_fma4(v, x.a, acc.a);
  if constexpr (BF) sp_fma4(v, x.b, acc.b);
}
__device__ __forceinline__ void sp_add4(float4& d, const float4& s) { d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w; }
template <bool BF>
__device__ __forceinline__ void sp_add(SpAcc<BF>& d, const SpAcc<BF>& s) {
  sp_add4(d.a, s.a);
  if constexpr (BF) sp_add4(d.b, s.b);
}
__device__ __forceinline__ float4 sp_xor4(const float4& v, int o) {
  return make_float4(__shfl_xor(v.x, o, 64), __shfl_xor(v.y, o, 64), __shfl_xor(v.z, o, 64), __shfl_xor(v.w, o, 64));
}
template <bool BF>
__device__ __forceinline__ SpAcc<BF> sp_load_f32(const float* __restrict__ p) {  // 4 / 8 consecutive floats (partials, bias)
  SpAcc<BF> r;
  r.a = *reinterpret_cast<const float4*>(p);
  if constexpr (BF) r.b = *reinterpret_cast<const float4*>(p + 4);
  return r;
}
template <bool BF>
__device__ __forceinline__ void sp_store_f32(float* __restrict__ p, const SpAcc<BF>& v) {
  *reinterpret_cast<float4*>(p) = v.a;
  if constexpr (BF) *reinterpret_cast<float4*>(p + 4) = v.b;
}
__device__ __forceinline__ unsigned sp_pack2(float lo, float hi) { return (unsigned)ceeg_f32_to_bf16(lo) | ((unsigned)ceeg_f32_to_bf16(hi) << 16); }

// The finished 16 bytes of the previous row, stored only after the next row's first gathers are in flight: on gfx9 stores and
// loads share vmcnt and retire in order, so a store issued BEFORE those gathers sits in front of them in every wait
// (F = 256 aggregation of config 2: 1.42 -> 1.38 ms; profiles/r03_spmm_bound.txt).
struct SpPend {
  uint4 raw;
  void* addr = nullptr;  // per lane; nullptr = nothing pending
  __device__ __forceinline__ void flush() {
    if (addr) *reinterpret_cast<uint4*>(addr) = raw;
    addr = nullptr;
  }
};

template <int LPR, int K, bool BF = false>
__device__ __forceinline__ void gather_steps(int my_c, float my_v, int i0, int n, int sub, const void* __restrict__ X,
                                             int64_t ldx, int foff, bool fl_ok, SpAcc<BF>& acc, SpPend* pend = nullptr) {
  constexpr int G = 64 / LPR;
  uint4 x[K];
  float v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int i = i0 + k;
    int c;
    bool ld = fl_ok;
    if constexpr (G == 1) {
      c = __builtin_amdgcn_readlane(my_c, i);
      v[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_v), i));
    } else {
      const int j = i * G + sub;
      c = __shfl(my_c, j, 64);
      v[k] = __shfl(my_v, j, 64);
      ld = ld && j < n;
    }
    x[k] = make_uint4(0u, 0u, 0u, 0u);
    if (ld) x[k] = sp_load_raw<BF>(X, (int64_t)c * ldx + foff);
  }
  if (pend) pend->flush();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if constexpr (G > 1) {
      if (!(i0 + k < (n + G - 1) / G && (i0 + k) * G + sub < n)) continue;  // lanes past the batch end add nothing
    }
    sp_fma<BF>(v[k], x[k], acc);
  }
}

// one lane-parallel load of up to 64 (column, value) pairs starting at position `base`
__device__ __forceinline__ void load_batch(const int32_t* __restrict__ col, const float* __restrict__ val,
                                           const int32_t* __restrict__ vperm, int base, int p_hi, int lane, const CeegSpmmEx& ex,
                                           int& my_c, float& my_v) {
  my_c = 0;
  my_v = 0.f;
  if (ex.colval) {
    if (base + lane < p_hi) {
      const int2 cv = ex.colval[base + lane];
      my_c = cv.x;
      my_v = __int_as_float(cv.y);
    }
    return;
  }
  if (base + lane < p_hi) {
    my_c = (ex.nt & 2) ? __builtin_nontemporal_load(col + base + lane) : col[base + lane];
    if (ex.col_a != 1 || ex.col_b != 0) my_c = (int)ceeg_block_affine((int64_t)my_c, ex.col_a, ex.col_b, ex.mod);
    if (ex.xidx) my_c = ex.xidx[my_c];  // X is a table: node c's row is X[xidx[c]]
    const float* vp = val + (int64_t)(vperm ? vperm[base + lane] : base + lane) * ex.val_stride;
    my_v = ((ex.nt & 2) && !vperm) ? __builtin_nontemporal_load(vp) : *vp;
  }
}

// sum over positions [p_lo, p_hi); (my_c, my_v) hold the already-loaded first batch (positions p_lo .. p_lo+63)
template <int LPR, bool BF = false>
__device__ __forceinline__ SpAcc<BF> seg_sum_pre(const int32_t* __restrict__ col, const float* __restrict__ val,
                                              const int32_t* __restrict__ vperm, const void* __restrict__ X, int64_t ldx,
                                              int p_lo, int p_hi, int lane, int foff, bool fl_ok, const CeegSpmmEx& ex,
                                              int my_c, float my_v, SpPend* pend = nullptr) {
  constexpr int G = 64 / LPR;
  const int sub = lane / LPR;
  SpAcc<BF> acc = sp_zero<BF>();
  for (int base = p_lo; base < p_hi; base += 64) {
    const int n = p_hi - base < 64 ? p_hi - base : 64;
    if (base != p_lo) load_batch(col, val, vperm, base, p_hi, lane, ex, my_c, my_v);
    const int ns = (n + G - 1) / G;  // steps in this batch (wave-uniform)
    int i0 = 0;
    for (; i0 + 8 <= ns; i0 += 8) gather_steps<LPR, 8, BF>(my_c, my_v, i0, n, sub, X, ldx, foff, fl_ok, acc, pend);
    const int rem = ns - i0;
    if (rem & 4) { gather_steps<LPR, 4, BF>(my_c, my_v, i0, n, sub, X, ldx, foff, fl_ok, acc, pend); i0 += 4; }
    if (rem & 2) { gather_steps<LPR, 2, BF>(my_c, my_v, i0, n, sub, X, ldx, foff, fl_ok, acc, pend); i0 += 2; }
    if (rem & 1) { gather_steps<LPR, 1, BF>(my_c, my_v, i0, n, sub, X, ldx, foff, fl_ok, acc, pend); }
  }
  if constexpr (G > 1) {
#pragma unroll
    for (int o = 32; o >= LPR; o >>= 1) {
      sp_add4(acc.a, sp_xor4(acc.a, o));
      if constexpr (BF) sp_add4(acc.b, sp_xor4(acc.b, o));
    }
  }
  return acc;
}

template <int LPR, bool BF = false>
__device__ __forceinline__ SpAcc<BF> seg_sum(const int32_t* __restrict__ col, const float* __restrict__ val,
                                          const int32_t* __restrict__ vperm, const void* __restrict__ X, int64_t ldx,
                                          int p_lo, int p_hi, int lane, int foff, bool fl_ok, const CeegSpmmEx& ex) {
  int my_c;
  float my_v;
  load_batch(col, val, vperm, p_lo, p_hi, lane, ex, my_c, my_v);
  return seg_sum_pre<LPR, BF>(col, val, vperm, X, ldx, p_lo, p_hi, lane, foff, fl_ok, ex, my_c, my_v);
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
template <int LPR, bool BF = false>
__global__ __launch_bounds__(SPMM_WAVES * 64) void spmm_long_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ vperm, const void* __restrict__ X, int64_t ldx, int64_t N, int F, int fbase,
    float* __restrict__ partial, CeegSpmmEx ex) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * SPMM_WAVES + (threadIdx.x >> 6);
  const int nnz = rowptr[N];
  const int64_t p0l = w * SPMM_CHUNK;
  if (p0l >= nnz) return;
  const int p0 = (int)p0l;
  const int p1 = p0 + SPMM_CHUNK < nnz ? p0 + SPMM_CHUNK : nnz;
  constexpr int EPL = sp_epl<BF>();
  const int fl = lane % LPR;
  const int foff = fbase + EPL * fl;
  const bool fl_ok = foff < F;
  const int r0 = row_of(rowptr, N, p0);
  const int r1 = row_of(rowptr, N, p1 - 1);
  if (rowptr[r0 + 1] - rowptr[r0] > SPMM_CHUNK) {  // slot 0: the long row that covers p0
    const int e = rowptr[r0 + 1] < p1 ? rowptr[r0 + 1] : p1;
    const SpAcc<BF> a = seg_sum<LPR, BF>(col, val, vperm, X, ldx, p0, e, lane, foff, fl_ok, ex);
    if (lane < LPR && fl_ok) sp_store_f32<BF>(partial + ((size_t)w * 2 + 0) * (EPL * LPR) + EPL * fl, a);
  }
  if (r1 != r0 && rowptr[r1 + 1] - rowptr[r1] > SPMM_CHUNK) {  // slot 1: a long row that starts inside
    const SpAcc<BF> a = seg_sum<LPR, BF>(col, val, vperm, X, ldx, rowptr[r1], p1, lane, foff, fl_ok, ex);
    if (lane < LPR && fl_ok) sp_store_f32<BF>(partial + ((size_t)w * 2 + 1) * (EPL * LPR) + EPL * fl, a);
  }
}

// the same pieces from a list the caller built once per graph: item = (lo, hi, chunk*2 + slot, row); one wave per item,
// no row search and no waves for the chunks that hold no long row (config 2's transposed graph: 3.4 k items in 66 k chunks)
template <int LPR, bool BF = false>
__global__ __launch_bounds__(SPMM_WAVES * 64) void spmm_long_items_kernel(
    const int4* __restrict__ items, int64_t n_items, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ vperm, const void* __restrict__ X, int64_t ldx, int F, int fbase, float* __restrict__ partial,
    CeegSpmmEx ex) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * SPMM_WAVES + (threadIdx.x >> 6);
  if (w >= n_items) return;
  const int4 it = items[w];
  constexpr int EPL = sp_epl<BF>();
  const int fl = lane % LPR;
  const int foff = fbase + EPL * fl;
  const bool fl_ok = foff < F;
  const SpAcc<BF> a = seg_sum<LPR, BF>(col, val, vperm, X, ldx, it.x, it.y, lane, foff, fl_ok, ex);
  if (lane < LPR && fl_ok) sp_store_f32<BF>(partial + (size_t)it.z * (EPL * LPR) + EPL * fl, a);
}

// first row r in [0, N] with rowptr[r] >= target
__device__ __forceinline__ int64_t row_lower_bound(const int32_t* __restrict__ rowptr, int64_t N, int64_t target) {
  int64_t lo = 0, hi = N;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (rowptr[mid] < target) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Persistent row kernel with one in-order work queue per XCD.
//  * XCD-aware AND nonzero-balanced: the workgroups sharing blockIdx%8 (= one XCD, one L2) own one
//    CONTIGUOUS range of rows holding ~1/8 of the nonzeros (hub rows of the transposed event graph sit at
//    low ids: equal row counts would leave XCD 0 with 3x the work).
//  * Inside the range, workgroups draw the next SPMM_GRAB consecutive rows from the XCD's counter
//    (one returning atomic per workgroup per grab), so the XCD sweeps its range strictly in order with a
//    tight window of live rows: time-local events re-use their gathered neighbours from that L2, and
//    heavy rows cannot unbalance the waves.  Which wave computes a row never changes its value.
//  Every workgroup leaves when its draw passes the end of the range (the counter only grows).
template <int LPR, bool BF = false>
__global__ __launch_bounds__(SPMM_WAVES * 64) void spmm_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ val,
    const int32_t* __restrict__ vperm, const void* __restrict__ X, int64_t ldx, int64_t N, int F, int fbase,
    const float* __restrict__ bias, int relu, int accumulate, void* __restrict__ out, int64_t ldo,
    const float* __restrict__ partial, unsigned int* __restrict__ queue /* 8 counters, 128 B apart, zeroed before the launch */, CeegSpmmEx ex) {
  static_assert(SPMM_GRAB == SPMM_WAVES, "one row per wave per draw");
  __shared__ unsigned int s_draw[2];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7;
  const int64_t nnz = rowptr[N];
  const int64_t r_begin = xcd == 0 ? 0 : row_lower_bound(rowptr, N, nnz * xcd / 8);
  const int64_t r_end = xcd == 7 ? N : row_lower_bound(rowptr, N, nnz * (xcd + 1) / 8);
  constexpr int EPL = sp_epl<BF>();
  const int fl = lane % LPR;
  const int foff = fbase + EPL * fl;
  const bool fl_ok = foff < F;
  const int subq = (blockIdx.x >> 3) % SPMM_SUBQ;
  unsigned int* q = queue + (xcd * SPMM_SUBQ + subq) * 32;
  // draw value d (a multiple of GRAB) of sub-queue s -> row set (d/GRAB)*SUBQ + s: the sub-queues interleave
  auto set_base = [&](unsigned d) -> int64_t { return r_begin + ((int64_t)(d / SPMM_GRAB) * SPMM_SUBQ + subq) * SPMM_GRAB; };

  // Software pipeline over the workgroup's draws (one row per wave per draw):
  //   iteration `it` GATHERS row(it) while the (col,val) batch of row(it+1), the row pointers of row(it+2)
  //   and the queue draw for it+2 are in flight - each dependent load gets a whole iteration to land, so a
  //   row costs one gather latency instead of the rowptr -> col/val -> gather chain.
  // Only metadata runs ahead; the rows being gathered stay one consecutive band per XCD.
  auto row_ptrs = [&](int64_t base) -> int {  // lanes 0,1: rowptr[r], rowptr[r+1] (0,0 past the range)
    const int64_t r = base + wave;
    int v = 0;
    if (lane < 2 && r < r_end) v = rowptr[r + lane];
    return v;
  };
  if (threadIdx.x == 0) {
    const unsigned d0 = atomicAdd(q, (unsigned)SPMM_GRAB);
    const unsigned d1 = atomicAdd(q, (unsigned)SPMM_GRAB);
    s_draw[0] = d0;
    s_draw[1] = d1;
  }
  __syncthreads();
  int64_t base_cur = set_base(s_draw[0]), base_nxt = set_base(s_draw[1]);
  __syncthreads();
  int rp = row_ptrs(base_cur);
  int p_lo = __builtin_amdgcn_readlane(rp, 0), p_hi = __builtin_amdgcn_readlane(rp, 1);
  int bc = 0;
  float bv = 0.f;
  if (p_hi - p_lo <= SPMM_CHUNK) load_batch(col, val, vperm, p_lo, p_hi, lane, ex, bc, bv);
  int rp_n = row_ptrs(base_nxt);
  unsigned int draw = 0;
  SpPend pend;
  for (int it = 0; base_cur < r_end; ++it) {
    if (threadIdx.x == 0) draw = atomicAdd(q, (unsigned)SPMM_GRAB);  // draw it+2
    // (col,val) batch of the next row
    const int p_lo_n = __builtin_amdgcn_readlane(rp_n, 0), p_hi_n = __builtin_amdgcn_readlane(rp_n, 1);
    int bc_n = 0;
    float bv_n = 0.f;
    if (p_hi_n - p_lo_n <= SPMM_CHUNK) load_batch(col, val, vperm, p_lo_n, p_hi_n, lane, ex, bc_n, bv_n);
    // gather + store the current row
    const int64_t r = base_cur + wave;
    if (r < r_end) {
      SpAcc<BF> acc;
      uint4 mk_raw = make_uint4(0u, 0u, 0u, 0u);
      if (ex.mask && lane < LPR && fl_ok) mk_raw = sp_load_raw<BF>(ex.mask, r * ex.ld_mask + foff);  // in flight under the gathers
      if (p_hi - p_lo <= SPMM_CHUNK) {
        acc = seg_sum_pre<LPR, BF>(col, val, vperm, X, ldx, p_lo, p_hi, lane, foff, fl_ok, ex, bc, bv, &pend);
      } else {
        acc = sp_zero<BF>();
        const int c_lo = p_lo / SPMM_CHUNK, c_hi = (p_hi - 1) / SPMM_CHUNK;
        if (lane < LPR && fl_ok) {
          for (int c = c_lo; c <= c_hi; ++c) {  // pieces in ascending position order
            const int slot = (c == c_lo && p_lo != c * SPMM_CHUNK) ? 1 : 0;
            sp_add<BF>(acc, sp_load_f32<BF>(partial + ((size_t)c * 2 + slot) * (EPL * LPR) + EPL * fl));
          }
        }
      }
      if (lane < LPR && fl_ok) {
        if constexpr (!BF) {  // the decoder backward's extras exist in the fp32 form only (ceeg_spmm_launch refuses them for bf16)
          if (ex.addend) sp_add4(acc.a, *reinterpret_cast<const float4*>(ex.addend + r * ex.ld_add + foff));
          if (ex.scale) {
            const float sc = ex.scale[0];
            acc.a.x *= sc; acc.a.y *= sc; acc.a.z *= sc; acc.a.w *= sc;
          }
        }
        if (bias) sp_add<BF>(acc, sp_load_f32<BF>(bias + foff));
        int64_t ro = r;
        if (ex.row_a != 1 || ex.row_b != 0) ro = ceeg_block_affine(r, ex.row_a, ex.row_b, ex.mod);
        if constexpr (!BF) {
          if (accumulate) sp_add4(acc.a, *reinterpret_cast<const float4*>(static_cast<const float*>(out) + ro * ldo + foff));
        }
        auto relu4 = [](float4& v) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); };
        auto mask4 = [](float4& v, const float4& m) {  // g * (y > 0): exactly what ceeg_relu_bwd_colsum writes
          v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        };
        if (relu) {
          relu4(acc.a);
          if constexpr (BF) relu4(acc.b);
        }
        if (ex.mask) {
          const SpAcc<BF> mk = sp_unpack<BF>(mk_raw);
          mask4(acc.a, mk.a);
          if constexpr (BF) mask4(acc.b, mk.b);
        }
        if constexpr (BF) {  // rounded to nearest-even once, on the store
          const uint4 u = make_uint4(sp_pack2(acc.a.x, acc.a.y), sp_pack2(acc.a.z, acc.a.w), sp_pack2(acc.b.x, acc.b.y), sp_pack2(acc.b.z, acc.b.w));
          pend.flush();
          pend.raw = u;
          pend.addr = static_cast<unsigned short*>(out) + ro * ldo + foff;
        } else {
          float* o = static_cast<float*>(out) + ro * ldo + foff;
          if (ex.nt & 1) {  // once-written output rows: keep them from evicting gathered rows
            __builtin_nontemporal_store(acc.a.x, o); __builtin_nontemporal_store(acc.a.y, o + 1);
            __builtin_nontemporal_store(acc.a.z, o + 2); __builtin_nontemporal_store(acc.a.w, o + 3);
          } else {
            pend.flush();  // a row without gathers in between (empty, or summed from pieces)
            pend.raw = make_uint4(__float_as_uint(acc.a.x), __float_as_uint(acc.a.y), __float_as_uint(acc.a.z), __float_as_uint(acc.a.w));
            pend.addr = o;
          }
        }
      }
    }
    // publish draw it+2 (slot it&1 was last read two barriers ago), start the row-pointer load of row(it+2)
    if (threadIdx.x == 0) s_draw[it & 1] = draw;
    __syncthreads();
    const int64_t base_nn = set_base(s_draw[it & 1]);
    base_cur = base_nxt;
    base_nxt = base_nn;
    p_lo = p_lo_n; p_hi = p_hi_n; bc = bc_n; bv = bv_n;
    rp_n = row_ptrs(base_nn);
  }
  pend.flush();
}


int lpr_for(int F, int epl = 4) {
  const int need = (F + epl - 1) / epl;
  int l = 1;
  while (l < need && l < 64) l <<= 1;
  return l;
}

template <int LPR, bool BF = false>
int launch_pass(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm, const void* X,
                int64_t ldx, int64_t N, int64_t nnz, int F, int fbase, const float* bias, int relu, int accumulate,
                void* out, int64_t ldo, float* partial, unsigned int* queue, hipStream_t st, const CeegSpmmEx& ex) {
  const int64_t nchunk = (nnz + SPMM_CHUNK - 1) / SPMM_CHUNK;
  if (nchunk > 0 && ex.long_items != nullptr) {  // the caller's piece list covers every row longer than SPMM_CHUNK
    if (ex.n_long_items > 0) {
      hipLaunchKernelGGL((spmm_long_items_kernel<LPR, BF>), dim3((unsigned)((ex.n_long_items + SPMM_WAVES - 1) / SPMM_WAVES)), dim3(SPMM_WAVES * 64),
                         0, st, reinterpret_cast<const int4*>(ex.long_items), ex.n_long_items, col, val, vperm, X, ldx, F, fbase, partial, ex);
      CEEG_CHECK_LAUNCH();
    }
  } else if (nchunk > 0 && !(ex.max_row_nnz >= 0 && ex.max_row_nnz <= SPMM_CHUNK)) {  // skipped when the caller knows there is no long row
    hipLaunchKernelGGL((spmm_long_kernel<LPR, BF>), dim3((unsigned)((nchunk + SPMM_WAVES - 1) / SPMM_WAVES)), dim3(SPMM_WAVES * 64), 0, st,
                       rowptr, col, val, vperm, X, ldx, N, F, fbase, partial, ex);
    CEEG_CHECK_LAUNCH();
  }
  if (hipMemsetAsync(queue, 0, SPMM_QUEUE_BYTES, st) != hipSuccess) return CEEG_ELAUNCH;
  int64_t per_xcd = (N + 8 * SPMM_GRAB - 1) / (8 * SPMM_GRAB);  // blocks per XCD: enough for one draw each ...
  static const int bpx_cap = getenv("CEEG_SPMM_BPX") ? atoi(getenv("CEEG_SPMM_BPX")) : SPMM_BLOCKS_PER_XCD;  // tuning knob
  if (per_xcd > bpx_cap) per_xcd = bpx_cap;  // ... capped at 8 resident blocks on each of 32 CUs
  if (per_xcd < SPMM_SUBQ) per_xcd = SPMM_SUBQ;  // every interleaved sub-queue needs a workgroup that drains it (tiny graphs)
  hipLaunchKernelGGL((spmm_rows_kernel<LPR, BF>), dim3((unsigned)(8 * per_xcd)), dim3(SPMM_WAVES * 64), 0, st, rowptr, col, val, vperm, X, ldx,
                     N, F, fbase, bias, relu, accumulate, out, ldo, partial, queue, ex);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}


// ---------------------------------------------------------------------------------------------- gather probe
// Measurement kernel (ceeg_spmm_gather_probe): the row kernel's traversal (per-XCD in-order queue, one wave per row, 64
// column indices per lane-parallel load, 8 row gathers in flight) with everything but the gathers removed - no value stream,
// no FMA chain, no epilogue, no output matrix.  What it takes is the empirical ceiling of ANY SpMM on this graph that
// gathers whole feature rows through the cache hierarchy.  `mode` selects the entries gathered: bit 0 = near ones
// (|col - row| <= window), bit 1 = far ones; the split shows how much of the time the time-local edges (which an LDS-staged
// band could serve) and the hub edges (single-use rows from all over X) each account for.
template <int LPR>
__global__ __launch_bounds__(SPMM_WAVES * 64) void spmm_gather_probe_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ X, int64_t ldx, int64_t N,
    int F, int mode, int window, unsigned int* __restrict__ sink, unsigned int* __restrict__ queue) {
  __shared__ unsigned int s_draw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7;
  const int64_t nnz = rowptr[N];
  const int64_t r_begin = xcd == 0 ? 0 : row_lower_bound(rowptr, N, nnz * xcd / 8);
  const int64_t r_end = xcd == 7 ? N : row_lower_bound(rowptr, N, nnz * (xcd + 1) / 8);
  constexpr int G = 64 / LPR;
  const int fl = lane % LPR, sub = lane / LPR;
  const int foff = 4 * fl;
  const bool fl_ok = foff < F;
  unsigned int* q = queue + xcd * 32;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (;;) {
    if (threadIdx.x == 0) s_draw = atomicAdd(q, (unsigned)SPMM_GRAB);
    __syncthreads();
    const int64_t r = r_begin + s_draw + wave;
    __syncthreads();
    if (r - wave >= r_end) break;
    if (r >= r_end) continue;
    const int p_lo = rowptr[r], p_hi = rowptr[r + 1];
    if (p_hi - p_lo > SPMM_CHUNK) continue;  // hub rows: the real kernel takes them by pieces in a launch of its own
    for (int base = p_lo; base < p_hi; base += 64) {
      const int n = p_hi - base < 64 ? p_hi - base : 64;
      int my_c = -1;
      if (base + lane < p_hi) {
        my_c = col[base + lane];
        const int d = my_c - (int)r;
        const bool near = (d < 0 ? -d : d) <= window;
        if (!((mode & 1) && near) && !((mode & 2) && !near)) my_c = -1;  // filtered out: no gather
      }
      const int ns = (n + G - 1) / G;
      for (int i0 = 0; i0 < ns; i0 += 8) {
        uint4 x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int j = (i0 + k) * G + sub;
          const int c = __shfl(my_c, j < 64 ? j : 63, 64);
          x[k] = make_uint4(0, 0, 0, 0);
          if (i0 + k < ns && j < n && c >= 0 && fl_ok) x[k] = *reinterpret_cast<const uint4*>(X + (int64_t)c * ldx + foff);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { acc.x ^= x[k].x; acc.y ^= x[k].y; acc.z ^= x[k].z; acc.w ^= x[k].w; }
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = acc.x;  // keeps the loads alive; practically never taken
}

// The same probe for a FEATURE-SLICED schedule: XCD x walks ALL rows but gathers only columns [x F/8, (x+1) F/8) of every
// source row (128 bytes at F = 256), so each L2 holds an eighth of every row it caches - eight times as many distinct rows -
// at the price of eight passes over the (col) stream.  An 8-lane group owns a row (eight rows per wave, eight gathers in
// flight per lane).  No product kernel is built on it; the measurement decides whether one would pay.
__global__ __launch_bounds__(SPMM_WAVES * 64) void spmm_gather_probe_sliced_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ X, int64_t ldx, int64_t N,
    int Fs, int mode, int window, unsigned int* __restrict__ sink, unsigned int* __restrict__ queue) {
  __shared__ unsigned int s_draw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7;
  const float* Xs = X + (int64_t)xcd * Fs;
  const int fl = lane & 7, sub = lane >> 3;
  const int foff = 4 * fl;
  const bool fl_ok = foff < Fs;
  unsigned int* q = queue + xcd * 32;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (;;) {
    if (threadIdx.x == 0) s_draw = atomicAdd(q, (unsigned)(SPMM_WAVES * 8));
    __syncthreads();
    const int64_t r0 = (int64_t)s_draw;
    __syncthreads();
    if (r0 >= N) break;
    const int64_t r = r0 + wave * 8 + sub;
    int p_lo = 0, p_hi = 0;
    if (r < N) {
      p_lo = rowptr[r];
      p_hi = rowptr[r + 1];
      if (p_hi - p_lo > SPMM_CHUNK) p_hi = p_lo;  // hub rows excluded, as in the whole-row probe
    }
    for (int base = p_lo; __any(base < p_hi); base += 8) {
      int my_c = -1;
      if (base + fl < p_hi) {
        my_c = col[base + fl];
        const int d = my_c - (int)r;
        const bool near = (d < 0 ? -d : d) <= window;
        if (!((mode & 1) && near) && !((mode & 2) && !near)) my_c = -1;
      }
      uint4 x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = __shfl(my_c, (lane & ~7) + k, 64);
        x[k] = make_uint4(0, 0, 0, 0);
        if (c >= 0 && fl_ok) x[k] = *reinterpret_cast<const uint4*>(Xs + (int64_t)c * ldx + foff);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) { acc.x ^= x[k].x; acc.y ^= x[k].y; acc.z ^= x[k].z; acc.w ^= x[k].w; }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = acc.x;
}

}  // namespace

extern "C" int ceeg_spmm_gather_probe(const int32_t* rowptr, const int32_t* col, const float* X, int64_t ldx, int64_t N, int64_t nnz,
                                      int F, int mode, int window, void* ws, size_t ws_bytes, void* stream) {
  if (N <= 0 || nnz < 0 || !rowptr || !col || !X || !ws) return CEEG_EINVAL;
  if (F < 4 || F > 256 || (F % 4) != 0 || (ldx % 4) != 0 || ldx < F || !ceeg_aligned(X, 16)) return CEEG_EUNSUPPORTED;
  if (ws_bytes < SPMM_QUEUE_BYTES + 256 || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned int* queue = static_cast<unsigned int*>(ws);
  unsigned int* sink = queue + SPMM_QUEUE_BYTES / 4;
  if (hipMemsetAsync(queue, 0, SPMM_QUEUE_BYTES, st) != hipSuccess) return CEEG_ELAUNCH;
  const int lpr = lpr_for(F);
  const dim3 grid(8 * SPMM_BLOCKS_PER_XCD), block(SPMM_WAVES * 64);
  if (mode & 4) {  // feature-sliced schedule: XCD x gathers columns [x F/8, (x+1) F/8) of every row
    if ((F % 32) != 0) return CEEG_EUNSUPPORTED;
    hipLaunchKernelGGL(spmm_gather_probe_sliced_kernel, grid, block, 0, st, rowptr, col, X, ldx, N, F / 8, mode, window, sink, queue);
    CEEG_CHECK_LAUNCH();
    return CEEG_OK;
  }
  switch (lpr) {
    case 64: hipLaunchKernelGGL(spmm_gather_probe_kernel<64>, grid, block, 0, st, rowptr, col, X, ldx, N, F, mode, window, sink, queue); break;
    case 32: hipLaunchKernelGGL(spmm_gather_probe_kernel<32>, grid, block, 0, st, rowptr, col, X, ldx, N, F, mode, window, sink, queue); break;
    case 16: hipLaunchKernelGGL(spmm_gather_probe_kernel<16>, grid, block, 0, st, rowptr, col, X, ldx, N, F, mode, window, sink, queue); break;
    default: return CEEG_EUNSUPPORTED;
  }
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

namespace {
}  // namespace

extern "C" size_t ceeg_spmm_ws_bytes(int64_t N, int64_t nnz, int F) {
  (void)N;
  if (nnz < 0 || F <= 0) return 0;
  const int lpr = lpr_for(F);
  const int64_t nchunk = (nnz + SPMM_CHUNK - 1) / SPMM_CHUNK;
  return SPMM_QUEUE_BYTES + ceeg_ws_part((size_t)(nchunk + 1) * 2 * 4 * lpr, 4);
}

int ceeg_spmm_launch(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                     const void* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                     int accumulate, void* out, int64_t ldo, void* ws, size_t ws_bytes, hipStream_t st,
                     const CeegSpmmEx& ex) {
  if (N <= 0 || nnz < 0 || !rowptr || !X || !out || (nnz > 0 && !ex.colval && (!col || !val))) return CEEG_EINVAL;
  if (ex.colval && (vperm || ex.xidx || ex.val_stride != 1 || ex.col_a != 1 || ex.col_b != 0 || !ceeg_aligned(ex.colval, 8))) return CEEG_EINVAL;
  if (F < 4 || F > 1024 || (F % 4) != 0 || (ldx % 4) != 0 || (ldo % 4) != 0 || ldx < F || ldo < F) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(X, 16) || !ceeg_aligned(out, 16) || (bias && !ceeg_aligned(bias, 16))) return CEEG_EINVAL;
  if (ex.bf16 && (accumulate || ex.addend || ex.xidx || (ex.nt & 1))) return CEEG_EUNSUPPORTED;
  if (nnz >= ((int64_t)1 << 31) || N >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (ws_bytes < ceeg_spmm_ws_bytes(N, nnz, F) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  unsigned int* queue = static_cast<unsigned int*>(ws);
  float* partial = reinterpret_cast<float*>(static_cast<char*>(ws) + SPMM_QUEUE_BYTES);
  const int epl = ex.bf16 ? 8 : 4;
  if (ex.bf16 && ((F % 8) != 0 || (ldx % 8) != 0 || (ldo % 8) != 0 || (ex.mask && (ex.ld_mask % 8) != 0))) return CEEG_EUNSUPPORTED;  // 16-byte lanes
  const int lpr = lpr_for(F, epl);
  int rc = CEEG_OK;
  for (int fbase = 0; fbase < F && rc == CEEG_OK; fbase += epl * lpr) {
#define CEEG_SPMM_CASE(L)                                                                                                       \
  case L:                                                                                                                       \
    rc = ex.bf16 ? launch_pass<L, true>(rowptr, col, val, vperm, X, ldx, N, nnz, F, fbase, bias, relu, accumulate, out, ldo, partial, queue, st, ex) \
                 : launch_pass<L, false>(rowptr, col, val, vperm, X, ldx, N, nnz, F, fbase, bias, relu, accumulate, out, ldo, partial, queue, st, ex); \
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

// measurement switch (read per call): CEEG_SPMM_NT bit 0 = non-temporal stores of `out`, bit 1 = non-temporal col/val loads
static int spmm_nt_env() {
  const char* e = getenv("CEEG_SPMM_NT");
  return e ? atoi(e) : 0;
}

extern "C" int ceeg_spmm_csr(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                             const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                             int accumulate, float* out, int64_t ldo, int64_t max_row_nnz, void* ws, size_t ws_bytes, void* stream) {
  CeegSpmmEx ex;
  ex.mod = N;
  ex.max_row_nnz = max_row_nnz;
  ex.nt = spmm_nt_env();
  return ceeg_spmm_launch(rowptr, col, val, vperm, X, ldx, N, nnz, F, bias, relu, accumulate, out, ldo, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), ex);
}

extern "C" int ceeg_spmm_csr_table(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                                   const float* table, int64_t ldt, const int32_t* xidx, int64_t N, int64_t nnz, int F, int accumulate,
                                   float* out, int64_t ldo, int64_t max_row_nnz, void* ws, size_t ws_bytes, void* stream) {
  if (!xidx) return CEEG_EINVAL;
  CeegSpmmEx ex;
  ex.mod = N;
  ex.max_row_nnz = max_row_nnz;
  ex.xidx = xidx;
  return ceeg_spmm_launch(rowptr, col, val, vperm, table, ldt, N, nnz, F, nullptr, 0, accumulate, out, ldo, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), ex);
}

extern "C" int ceeg_spmm_csr_items(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                                   const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* bias, int relu,
                                   int accumulate, float* out, int64_t ldo, int64_t max_row_nnz, const int32_t* long_items,
                                   int64_t n_long, void* ws, size_t ws_bytes, void* stream) {
  if (n_long < 0 || (n_long > 0 && (!long_items || !ceeg_aligned(long_items, 16)))) return CEEG_EINVAL;
  if (max_row_nnz < 0 || (max_row_nnz > SPMM_CHUNK && n_long == 0)) return CEEG_EINVAL;  // long rows need their pieces listed
  static const int32_t no_items[4] = {0, 0, 0, 0};
  CeegSpmmEx ex;
  ex.mod = N;
  ex.max_row_nnz = max_row_nnz;
  ex.nt = spmm_nt_env();
  ex.long_items = n_long > 0 ? long_items : no_items;  // non-NULL: "the list is authoritative" (host pointer never read when empty)
  ex.n_long_items = n_long;
  return ceeg_spmm_launch(rowptr, col, val, vperm, X, ldx, N, nnz, F, bias, relu, accumulate, out, ldo, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), ex);
}

extern "C" int ceeg_spmm_csr_masked(const int32_t* rowptr, const int32_t* col, const float* val, const int32_t* vperm,
                                    const float* X, int64_t ldx, int64_t N, int64_t nnz, int F, const float* mask, int64_t ld_mask,
                                    float* out, int64_t ldo, int64_t max_row_nnz, const int32_t* long_items, int64_t n_long,
                                    void* ws, size_t ws_bytes, void* stream) {
  if (!mask || ld_mask < F || (ld_mask % 4) != 0 || !ceeg_aligned(mask, 16)) return CEEG_EINVAL;
  if (n_long < 0 || (n_long > 0 && (!long_items || !ceeg_aligned(long_items, 16)))) return CEEG_EINVAL;
  if (max_row_nnz < 0 || (max_row_nnz > SPMM_CHUNK && n_long == 0)) return CEEG_EINVAL;
  static const int32_t no_items[4] = {0, 0, 0, 0};
  CeegSpmmEx ex;
  ex.mod = N;
  ex.max_row_nnz = max_row_nnz;
  ex.long_items = n_long > 0 ? long_items : no_items;
  ex.n_long_items = n_long;
  ex.mask = mask;
  ex.ld_mask = ld_mask;
  return ceeg_spmm_launch(rowptr, col, val, vperm, X, ldx, N, nnz, F, nullptr, 0, 0, out, ldo, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), ex);
}

// ceeg_spmm_csr_items / ceeg_spmm_csr_masked with the (column, value) pairs of the CSR interleaved into one 8-byte stream
extern "C" int ceeg_spmm_csr_cv(const int32_t* rowptr, const void* colval, const float* X, int64_t ldx, int64_t N, int64_t nnz, int F,
                                const float* bias, int relu, const float* mask, int64_t ld_mask, float* out, int64_t ldo,
                                int64_t max_row_nnz, const int32_t* long_items, int64_t n_long, void* ws, size_t ws_bytes, void* stream) {
  if (!colval && nnz > 0) return CEEG_EINVAL;
  if (mask && (ld_mask < F || (ld_mask % 4) != 0 || !ceeg_aligned(mask, 16))) return CEEG_EINVAL;
  if (n_long < 0 || (n_long > 0 && (!long_items || !ceeg_aligned(long_items, 16)))) return CEEG_EINVAL;
  if (max_row_nnz < 0 || (max_row_nnz > SPMM_CHUNK && n_long == 0)) return CEEG_EINVAL;
  static const int32_t no_items[4] = {0, 0, 0, 0};
  static const int2 no_cv = {0, 0};
  CeegSpmmEx ex;
  ex.mod = N;
  ex.max_row_nnz = max_row_nnz;
  ex.long_items = n_long > 0 ? long_items : no_items;
  ex.n_long_items = n_long;
  ex.mask = mask;
  ex.ld_mask = ld_mask;
  ex.colval = colval ? static_cast<const int2*>(colval) : &no_cv;
  return ceeg_spmm_launch(rowptr, nullptr, nullptr, nullptr, X, ldx, N, nnz, F, bias, relu, 0, out, ldo, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), ex);
}

// The reduced-precision form of ceeg_spmm_csr_cv: X, mask and out hold bf16 (2-byte elements, row strides in elements), sums in fp32.
extern "C" int ceeg_spmm_csr_cv_bf16(const int32_t* rowptr, const void* colval, const void* X, int64_t ldx, int64_t N, int64_t nnz, int F,
                                     const float* bias, int relu, const void* mask, int64_t ld_mask, void* out, int64_t ldo,
                                     int64_t max_row_nnz, const int32_t* long_items, int64_t n_long, void* ws, size_t ws_bytes, void* stream) {
  if (!colval && nnz > 0) return CEEG_EINVAL;
  if (mask && (ld_mask < F || (ld_mask % 8) != 0 || !ceeg_aligned(mask, 16))) return CEEG_EINVAL;
  if (n_long < 0 || (n_long > 0 && (!long_items || !ceeg_aligned(long_items, 16)))) return CEEG_EINVAL;
  if (max_row_nnz < 0 || (max_row_nnz > SPMM_CHUNK && n_long == 0)) return CEEG_EINVAL;
  static const int32_t no_items[4] = {0, 0, 0, 0};
  static const int2 no_cv = {0, 0};
  CeegSpmmEx ex;
  ex.mod = N;
  ex.max_row_nnz = max_row_nnz;
  ex.long_items = n_long > 0 ? long_items : no_items;
  ex.n_long_items = n_long;
  ex.mask = static_cast<const float*>(mask);
  ex.ld_mask = ld_mask;
  ex.colval = colval ? static_cast<const int2*>(colval) : &no_cv;
  ex.bf16 = 1;
  return ceeg_spmm_launch(rowptr, nullptr, nullptr, nullptr, X, ldx, N, nnz, F, bias, relu, 0, out, ldo, ws, ws_bytes,
                          static_cast<hipStream_t>(stream), ex);
}
