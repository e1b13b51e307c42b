// VectorQuantizer forward/backward for gfx950 (MI355X).
//
// Stands behind scripts/quantizer.py:25-67 (reference) — see include/ceeg.h for the contract.
//
// Forward = 4 launches on the caller's stream:
//   vq_prep_kernel   : ee[c] (fp32: |W_c|^2 fmaf chain; bf16: -|bf16(W_c)|^2/2) + the codebook image in
//                      exactly the (XOR-swizzled) order the main kernel wants in LDS        (tiny)
//   vq_main_*        : distance GEMM on MFMA + running argmin in registers + codebook-row gather
//                      (z_q), squared error partial per block, idx                          (dominant)
//   vq_hist_kernel   : per-slice integer histograms of idx in LDS -> partial table (no atomics in HBM)
//   vq_finalize      : fixed-order reduction -> loss, perplexity, hist
//
// MFMA mapping (16x16 output tiles; C/D: col = lane&15 = code, row = 4*(lane>>4)+reg = latent row):
//   fp32 : v_mfma_f32_16x16x4_f32, A lane(r,q) = z[r][4s+q], B lane(c,q) = W[c][4s+q]; one accumulator
//          chain per (row tile), k ascending from 0 => dot is the sequential fmaf chain of the oracle.
//   bf16 : v_mfma_f32_16x16x32_bf16, A lane(r,q) = z[r][32s+8q .. +8], B likewise from the codebook;
//          C is seeded with -ee/2 so the accumulator IS the score (no epilogue arithmetic).
// The codebook is streamed through LDS in chunks of CT codes; z tiles live in registers for the whole
// kernel (each z element is read from HBM exactly once).
#include <stdlib.h>
#include <mutex>
#include "ceeg_internal.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

namespace {

constexpr int VQ_WAVES = 4;        // waves per block
constexpr int VQ_THREADS = VQ_WAVES * 64;
constexpr int VQ_HIST_ROWS = 1024; // rows per histogram slice (one wave of the stable scatter walks a slice)
constexpr int VQ_FWD_SLICES = 128; // histogram slices of the forward (the backward's stable scatter takes VQ_HIST_ROWS rows per slice)
constexpr int VQ_SEG = 256;        // sorted rows per backward chunk

__host__ __device__ constexpr int vq_min(int a, int b) { return a < b ? a : b; }

// ------------------------------------------------------------------------------------------ prep
// ee / image.  One thread per code computes the norm chain; all threads then write image elements.
// Image layout (what the main kernel copies verbatim into LDS, one chunk of CT codes at a time):
//   chunk ch at byte offset ch*CHB, CHB = CT*DP*elt + CT*4:
//     [CT rows x DP elements, 16-byte pieces XOR-swizzled inside each row][CT floats: ee (fp32) or -ee/2 (bf16)]
template <bool BF16>
__global__ void vq_prep_kernel(const float* __restrict__ W, int n_e, int D, int n_ep, int DP, int CT,
                               float* __restrict__ ee, char* __restrict__ img) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t elt = BF16 ? 2 : 4;
  const size_t chb = (size_t)CT * DP * elt + (size_t)CT * 4;
  if (tid < n_ep) {
    const int c = (int)tid;
    float e;
    if (c < n_e) {
      e = 0.f;
      int k = 0;
      for (; k + 32 <= D && (D & 3) == 0; k += 32) {  // 8 x 16 bytes in flight; the chain stays k-ascending (W is 16-byte aligned)
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(W + (size_t)c * D + k + 4 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          float w4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float w = w4[t];
            if (BF16) w = ceeg_bf16_to_f32(ceeg_f32_to_bf16(w));
            e = __fmaf_rn(w, w, e);
          }
        }
      }
      for (; k < D; ++k) {
        float w = W[(size_t)c * D + k];
        if (BF16) w = ceeg_bf16_to_f32(ceeg_f32_to_bf16(w));
        e = __fmaf_rn(w, w, e);
      }
      if (BF16) e = -0.5f * e;
    } else {
      e = BF16 ? -INFINITY : INFINITY;  // padded codes can never win
    }
    ee[c] = e;
    *reinterpret_cast<float*>(img + (size_t)(c / CT) * chb + (size_t)CT * DP * elt + (size_t)(c % CT) * 4) = e;
  }
  const int64_t total = (int64_t)n_ep * DP;
  for (int64_t i = tid; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / DP), k = (int)(i % DP);
    const float w = (c < n_e && k < D) ? W[(size_t)c * D + k] : 0.f;
    char* rows = img + (size_t)(c / CT) * chb;
    const int cl = c % CT;
    if (BF16) {
      const int cpr = DP / 8;                       // 16-byte chunks per row
      const int swz = c & (vq_min(16, cpr) - 1);
      const int chunk = k >> 3;
      reinterpret_cast<unsigned short*>(rows)[(size_t)cl * DP + ((chunk ^ swz) << 3) + (k & 7)] = ceeg_f32_to_bf16(w);
    } else {
      const int q = k & 3, s = k >> 2;              // de-interleave by k mod 4 (the MFMA's k index)
      const int cpq = DP / 16;                      // 16-byte chunks per quarter row
      const int swz = c & (vq_min(16, cpq) - 1);
      const int chunk = s >> 2, e = s & 3;
      reinterpret_cast<float*>(rows)[(size_t)cl * DP + q * (DP / 4) + ((chunk ^ swz) << 2) + e] = w;
    }
  }
}

// Asynchronous global -> LDS copy of one image chunk (PIECES x 16 bytes) with global_load_lds_dwordx4: no
// VGPR staging, the data lands while the MFMAs of the current chunk run.  The LDS address of a piece is
// (wave-uniform base) + lane*16, so the base passed is lane 0's; tail pieces go to lanes 0..k of one wave.
template <int PIECES>
__device__ __forceinline__ void vq_stage_chunk(const char* __restrict__ gsrc, char* ldst) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i0 = 0; i0 < PIECES; i0 += VQ_THREADS) {
    const int i = i0 + (int)threadIdx.x;
    if (i < PIECES) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t)i * 16),
                                       (__attribute__((address_space(3))) void*)(ldst + (size_t)(i - lane) * 16), 16, 0, 0);
    }
  }
}

__device__ __forceinline__ float pick4(const float4& v, int q) {
  return q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w));
}

// block-level fixed-order sum of one float per thread -> result valid in thread 0
__device__ __forceinline__ float vq_block_sum(float v, float* red /* [VQ_WAVES] LDS */) {
  v = ceeg_wave_sum(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < VQ_WAVES; ++w) t += red[w];
  }
  return t;
}

// ------------------------------------------------------------------------------------------ fp32 main
// EXACT: D == DP (every 16-byte access is in bounds and aligned).  Otherwise the ragged tail is handled
// with clamped scalar loads + selects (branch-free) and guarded scalar stores.
template <int DP, bool EXACT>
__global__ __launch_bounds__(VQ_THREADS, (EXACT && DP <= 128) ? 3 : 2) void vq_main_f32(
    const float* __restrict__ z, const float* __restrict__ W, const float* __restrict__ ee,
    const float* __restrict__ img, int64_t B, int n_e, int D, int n_ep,
    float* __restrict__ z_q, int64_t* __restrict__ idx_out, float* __restrict__ err_part, const int* __restrict__ run_flag) {
  if (run_flag && *run_flag == 0) return;  // behind vq_screen_f32: only if that kernel raised its flag (uniform: no barrier is skipped)
  constexpr int RT = 2;                 // 16-row tiles per wave
  constexpr int ROWS_W = RT * 16;       // rows per wave
  constexpr int BM = VQ_WAVES * ROWS_W; // rows per block
  constexpr int CT = 32;                // codes per LDS stage
  constexpr int NU = DP / 4;            // float4 per row == MFMA k-steps
  constexpr int CPQ = DP / 16;
  constexpr int SWZ_MASK = vq_min(16, CPQ) - 1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CHB = CT * DP * 4 + CT * 4;                      // one image chunk: rows + ee
  int* lds_idx = reinterpret_cast<int*>(smem + 2 * CHB);         // [BM]
  float* lds_red = reinterpret_cast<float*>(smem + 2 * CHB + BM * 4);  // [VQ_WAVES]
  const char* gimg = reinterpret_cast<const char*>(img);
  vq_stage_chunk<CHB / 16>(gimg, smem);                          // chunk 0 flies while the z tile is loaded

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * BM + wave * ROWS_W;

  // ---- z tile -> registers: lane (r,q) keeps z[r][4u+q] (the MFMA A operand of k-step u); zz[r] is the
  // natural-order fmaf chain over the whole row, computed redundantly by the row's 4 lanes.
  // Rows past B are clamped to B-1 for the loads (their results are never stored).
  constexpr bool VIA_LDS = EXACT && DP <= 128;  // wider rows: no LDS to spare at two blocks per CU
  float a[RT][NU];
  float zz[RT];
  if constexpr (VIA_LDS) {
    // A wave's 16-row tile is one contiguous 16*DP*4-byte block: fetch it with fully coalesced 16-byte loads
    // (every 128-byte line touched once), park it in the wave's own LDS strip (row pitch DP+4 floats: the
    // strided reads below hit 64 distinct banks), and pick the MFMA layout out of LDS.
    constexpr int PITCH = DP + 4;
    constexpr int PPR = DP / 4;            // 16-byte pieces per row
    constexpr int NPC = 16 * PPR / 64;     // pieces per lane per tile
    // the strips lie over chunk buffer 1 and the idx/red words (first written after the barrier that ends the prologue)
    float* strip = reinterpret_cast<float*>(smem + CHB) + wave * 16 * PITCH;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float4 pc[NPC];
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const int p = j * 64 + lane;
        int64_t row = row0 + rt * 16 + p / PPR;
        row = row < B ? row : B - 1;
        pc[j] = *reinterpret_cast<const float4*>(z + row * D + 4 * (p % PPR));
      }
#pragma unroll
      for (int j = 0; j < NPC; ++j) {
        const int p = j * 64 + lane;
        *reinterpret_cast<float4*>(strip + (p / PPR) * PITCH + 4 * (p % PPR)) = pc[j];
      }
      asm volatile("" ::: "memory");
      // same wave wrote and reads the strip: LDS operations of one wave complete in order
      const float* zr = strip + r * PITCH;
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const float4 v = *reinterpret_cast<const float4*>(zr + 4 * u);
        acc = __fmaf_rn(v.x, v.x, acc);
        acc = __fmaf_rn(v.y, v.y, acc);
        acc = __fmaf_rn(v.z, v.z, acc);
        acc = __fmaf_rn(v.w, v.w, acc);
        a[rt][u] = zr[4 * u + q];
        if ((u & 7) == 7) asm volatile("" ::: "memory");  // register budget: <= 8 LDS reads in flight
      }
      zz[rt] = acc;
      asm volatile("" ::: "memory");  // the next tile overwrites the strip only after these reads were issued
    }
  } else {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int64_t row = row0 + rt * 16 + r;
    row = row < B ? row : B - 1;
    const float* zr = z + row * D;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if constexpr (EXACT) {
        a[rt][u] = zr[4 * u + q];
      } else {
        const int k = 4 * u + q;
        const float t = zr[k < D ? k : D - 1];
        a[rt][u] = k < D ? t : 0.f;
      }
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      float4 v;
      if constexpr (EXACT) {
        v = *reinterpret_cast<const float4*>(zr + 4 * u);
      } else {
        const int k = 4 * u, dm = D - 1;
        v.x = zr[k + 0 < D ? k + 0 : dm]; v.x = k + 0 < D ? v.x : 0.f;
        v.y = zr[k + 1 < D ? k + 1 : dm]; v.y = k + 1 < D ? v.y : 0.f;
        v.z = zr[k + 2 < D ? k + 2 : dm]; v.z = k + 2 < D ? v.z : 0.f;
        v.w = zr[k + 3 < D ? k + 3 : dm]; v.w = k + 3 < D ? v.w : 0.f;
      }
      acc = __fmaf_rn(v.x, v.x, acc);
      acc = __fmaf_rn(v.y, v.y, acc);
      acc = __fmaf_rn(v.z, v.z, acc);
      acc = __fmaf_rn(v.w, v.w, acc);
      if ((u & 7) == 7) asm volatile("" ::: "memory");  // keep <= 8 float4 loads in flight (register budget)
    }
    zz[rt] = acc;  // identical in the 4 q-lanes of row r
  }
  }
  // zz of the rows this lane owns in the C layout (row = 4q + i)
  float zzc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) zzc[rt][i] = __shfl(zz[rt], 4 * q + i, 64);

  float best[RT][4];
  int bidx[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { best[rt][i] = INFINITY; bidx[rt][i] = 0; }

  const int nch = n_ep / CT;
  __syncthreads();  // chunk 0 landed (the barrier waits for the outstanding LDS-DMA)
  for (int ch = 0; ch < nch; ++ch) {
    // double buffer: chunk ch+1 streams into the other buffer while this chunk feeds the MFMAs
    if (ch + 1 < nch) vq_stage_chunk<CHB / 16>(gimg + (size_t)(ch + 1) * CHB, smem + ((ch + 1) & 1) * CHB);
    const float* lds_w = reinterpret_cast<const float*>(smem + (ch & 1) * CHB);
    const float* lds_ee = lds_w + CT * DP;
#pragma unroll
    for (int ct = 0; ct < CT / 16; ++ct) {
      f32x4 acc[RT];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* wrow = lds_w + (ct * 16 + r) * DP + q * (DP / 4);
      const int swz = r & SWZ_MASK;
#pragma unroll
      for (int t = 0; t < CPQ; ++t) {
        const float4 b = *reinterpret_cast<const float4*>(wrow + ((t ^ swz) << 2));
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][4 * t + e], bb[e], acc[rt], 0, 0, 0);
      }
      const int code = ch * CT + ct * 16 + r;
      const float eev = lds_ee[ct * 16 + r];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = __fmaf_rn(-2.0f, acc[rt][i], zzc[rt][i] + eev);  // fl(fl(zz+ee) - 2 dot)
          const bool better = d < best[rt][i];                             // strict: lowest code wins ties
          best[rt][i] = better ? d : best[rt][i];
          bidx[rt][i] = better ? code : bidx[rt][i];
        }
    }
    __syncthreads();  // next chunk landed + every wave is done reading this one
  }
  // ---- argmin across the 16 lanes (codes) that share this lane's rows; ties -> lowest index
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float d = best[rt][i];
      int bi = bidx[rt][i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float od = __shfl_xor(d, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool take = (od < d) || (od == d && oi < bi);
        d = take ? od : d;
        bi = take ? oi : bi;
      }
      if (r == 0) lds_idx[wave * ROWS_W + rt * 16 + 4 * q + i] = bi;
    }
  __syncthreads();

  // ---- z_q gather, squared error, idx
  float err = 0.f;
  if constexpr (VIA_LDS) {
    // the wave's ROWS_W output rows are one contiguous block: lane-consecutive 16-byte pieces, each the matching
    // piece of its row's codebook row (L1/L2-resident) - stores and the z re-read are fully coalesced
    constexpr int PPR = DP / 4;
    constexpr int NP = ROWS_W * PPR / 64;
    if (lane < ROWS_W && row0 + lane < B) idx_out[row0 + lane] = lds_idx[wave * ROWS_W + lane];
#pragma unroll 4
    for (int j = 0; j < NP; ++j) {
      const int p = j * 64 + lane;
      const int lr = p / PPR, c4 = 4 * (p % PPR);
      const int64_t row = row0 + lr;
      if (row < B) {
        const int id = lds_idx[wave * ROWS_W + lr];
        const float4 wv = *reinterpret_cast<const float4*>(W + (size_t)id * D + c4);
        const float4 zv = *reinterpret_cast<const float4*>(z + row * D + c4);
        *reinterpret_cast<float4*>(z_q + row * D + c4) = wv;
        const float dx = wv.x - zv.x, dy = wv.y - zv.y, dz = wv.z - zv.z, dw = wv.w - zv.w;
        err = __fmaf_rn(dx, dx, err);
        err = __fmaf_rn(dy, dy, err);
        err = __fmaf_rn(dz, dz, err);
        err = __fmaf_rn(dw, dw, err);
      }
    }
  } else {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t row = row0 + rt * 16 + r;
    const bool live = row < B;
    const int id = lds_idx[wave * ROWS_W + rt * 16 + r];
    const float* wr = W + (size_t)id * D;
    float* out = z_q + (live ? row : 0) * D;
    if (live && q == 0) idx_out[row] = id;
    float e = 0.f;
#pragma unroll
    for (int u = 0; u < NU; ++u) {  // squared error on the lane's own k-slice (k = 4u+q)
      float wv;
      if constexpr (EXACT) {
        wv = wr[4 * u + q];
      } else {
        const int k = 4 * u + q;
        wv = wr[k < D ? k : D - 1];
        wv = k < D ? wv : 0.f;
      }
      const float df = wv - a[rt][u];  // padded k: 0 - 0
      e = __fmaf_rn(df, df, e);
    }
    if (live) {  // z_q row = codebook row: the row's 4 lanes copy alternating 16-byte pieces
      if constexpr (EXACT) {
#pragma unroll
        for (int u4 = 0; u4 < NU / 4; ++u4) {
          const int u = 4 * u4 + q;
          *reinterpret_cast<float4*>(out + 4 * u) = *reinterpret_cast<const float4*>(wr + 4 * u);
        }
      } else {
        for (int k = q; k < D; k += 4) out[k] = wr[k];
      }
    }
    err += live ? e : 0.f;
  }
  }
  const float tot = vq_block_sum(err, lds_red);
  if (threadIdx.x == 0) err_part[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------ bf16 main
template <int DP, bool EXACT>
__global__ __launch_bounds__(VQ_THREADS, 2) void vq_main_bf16(
    const unsigned short* __restrict__ z, const float* __restrict__ W, const float* __restrict__ nhe,
    const unsigned short* __restrict__ img, int64_t B, int n_e, int D, int n_ep,
    unsigned short* __restrict__ z_q, int64_t* __restrict__ idx_out, float* __restrict__ err_part) {
  constexpr int RT = 4;
  constexpr int ROWS_W = RT * 16;
  constexpr int BM = VQ_WAVES * ROWS_W;  // 256 rows per block
  constexpr int CT = 64;
  constexpr int KS = DP / 32;            // MFMA k-steps
  constexpr int CPR = DP / 8;            // 16-byte chunks per codebook row
  constexpr int SWZ_MASK = vq_min(16, CPR) - 1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CHB = CT * DP * 2 + CT * 4;                                  // one image chunk: rows + (-ee/2)
  int* lds_idx = reinterpret_cast<int*>(smem + 2 * CHB);                     // [BM]
  float* lds_red = reinterpret_cast<float*>(smem + 2 * CHB + BM * 4);        // [VQ_WAVES]
  const char* gimg = reinterpret_cast<const char*>(img);
  vq_stage_chunk<CHB / 16>(gimg, smem);                                      // chunk 0 flies while the z tile is loaded

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * BM + wave * ROWS_W;

  u16x8 a[RT][KS];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    int64_t row = row0 + rt * 16 + r;
    row = row < B ? row : B - 1;
    const unsigned short* zr = z + row * D;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 32 * s + 8 * q;
      u16x8 v;
      if constexpr (EXACT) {
        v = *reinterpret_cast<const u16x8*>(zr + k0);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned short t = zr[k0 + j < D ? k0 + j : D - 1];
          v[j] = k0 + j < D ? t : (unsigned short)0;
        }
      }
      a[rt][s] = v;
    }
  }

  float best[RT][4];
  int bidx[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) { best[rt][i] = -INFINITY; bidx[rt][i] = 0; }

  const int nch = n_ep / CT;
  __syncthreads();  // chunk 0 landed (the barrier waits for the outstanding LDS-DMA)
  for (int ch = 0; ch < nch; ++ch) {
    if (ch + 1 < nch) vq_stage_chunk<CHB / 16>(gimg + (size_t)(ch + 1) * CHB, smem + ((ch + 1) & 1) * CHB);
    const unsigned short* lds_w = reinterpret_cast<const unsigned short*>(smem + (ch & 1) * CHB);
    const float* lds_nhe = reinterpret_cast<const float*>(smem + (ch & 1) * CHB + CT * DP * 2);
#pragma unroll
    for (int ct = 0; ct < CT / 16; ++ct) {
      const int code = ch * CT + ct * 16 + r;
      const float bias = lds_nhe[ct * 16 + r];  // -|w|^2/2 (or -inf for padding)
      f32x4 acc[RT];
      const unsigned short* wrow = lds_w + (ct * 16 + r) * DP;
      const int swz = r & SWZ_MASK;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const u16x8 b = *reinterpret_cast<const u16x8*>(wrow + (((4 * s + q) ^ swz) << 3));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const f32x4 cin = (s == 0) ? (f32x4){bias, bias, bias, bias} : acc[rt];
          acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[rt][s]),
                                                           __builtin_bit_cast(bf16x8_t, b), cin, 0, 0, 0);
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float sc = acc[rt][i];
          const bool better = sc > best[rt][i];  // strict: lowest code wins ties
          best[rt][i] = better ? sc : best[rt][i];
          bidx[rt][i] = better ? code : bidx[rt][i];
        }
    }
    __syncthreads();  // next chunk landed + every wave is done reading this one
  }
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float sc = best[rt][i];
      int bi = bidx[rt][i];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const float os = __shfl_xor(sc, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool take = (os > sc) || (os == sc && oi < bi);
        sc = take ? os : sc;
        bi = take ? oi : bi;
      }
      if (r == 0) lds_idx[wave * ROWS_W + rt * 16 + 4 * q + i] = bi;
    }
  __syncthreads();

  float err = 0.f;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int64_t row = row0 + rt * 16 + r;
    const bool live = row < B;
    const int id = lds_idx[wave * ROWS_W + rt * 16 + r];
    const float* wr = W + (size_t)id * D;
    unsigned short* out = z_q + (live ? row : 0) * D;
    if (live && q == 0) idx_out[row] = id;
    float e = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 32 * s + 8 * q;
      float w[8];
      if constexpr (EXACT) {
        const float4 w0 = *reinterpret_cast<const float4*>(wr + k0);
        const float4 w1 = *reinterpret_cast<const float4*>(wr + k0 + 4);
        w[0] = w0.x; w[1] = w0.y; w[2] = w0.z; w[3] = w0.w;
        w[4] = w1.x; w[5] = w1.y; w[6] = w1.z; w[7] = w1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = wr[k0 + j < D ? k0 + j : D - 1];
          w[j] = k0 + j < D ? t : 0.f;
        }
      }
      bf16x8_t ob;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ob[j] = (__bf16)w[j];  // v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
        const float df = (float)ob[j] - ceeg_bf16_to_f32(a[rt][s][j]);  // against the row z_q returns: loss = (1+beta) mean((z_q - z)^2)
        e = __fmaf_rn(df, df, e);
      }
      const u16x8 o = __builtin_bit_cast(u16x8, ob);
      if constexpr (EXACT) {
        if (live) *reinterpret_cast<u16x8*>(out + k0) = o;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (live && k0 + j < D) out[k0 + j] = o[j];
      }
    }
    err += live ? e : 0.f;
  }
  const float tot = vq_block_sum(err, lds_red);
  if (threadIdx.x == 0) err_part[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------------------ bf16 main, codebook-stationary
// D == DP, n_ep == 128 * NCT (NCT in {1, 2, 4}): the WHOLE codebook lives in the registers of one 512-thread workgroup for
// the lifetime of a persistent kernel - wave w holds codes [16 NCT w, 16 NCT (w+1)) as MFMA B operands (NCT * DP/8 VGPRs)
// - and the latent rows stream through LDS in double-buffered 64-row tiles (LDS-DMA, XOR-swizzled pieces).  Against
// vq_main_bf16 (rows stationary, codebook re-staged through LDS for every 256 rows):
//   * no codebook traffic after the first microseconds, no chunk barriers;
//   * the streamed operand is the SMALL one per MFMA batch (A: 4 VGPRs per k-step, double-buffered by the compiler,
//     where the row-stationary kernel at 256 VGPRs waits on every single-buffered ds_read_b128);
//   * per 16-row tile a wave reduces its NCT*16 codes across the 16 code lanes with DPP row rotations (v_max / v_min with
//     a DPP operand: 10 VALU per row value), the 8 waves' candidates meet in LDS, lowest code wins ties (strict > in
//     ascending wave order = ascending code order);
//   * squared error of a row = |z|^2 - 2 * best score (the accumulator, seeded with -|w|^2/2, IS z.w - |w|^2/2 for the
//     bf16-rounded row that z_q returns); z_q rows are copied from the bf16 image.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
constexpr int VQ_BS_WAVES = 8;
constexpr int VQ_BS_TR = 64;  // rows per LDS tile


template <int CTRL>
__device__ __forceinline__ float vq_dpp_f(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, false)); }
template <int CTRL>
__device__ __forceinline__ int vq_dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false); }

// max without fmaxf's NaN canonicalisation of each operand (v_med3_f32 with +inf: one VALU op)
__device__ __forceinline__ float vq_max(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
// combine a value over the wave's four 16-lane rows (lanes r, r+16, r+32, r+48), one xor step per call, with gfx950's
// v_permlane16_swap / v_permlane32_swap (VALU, no LDS round trip): after the swap the two results hold both partners.
template <int STEP>
__device__ __forceinline__ float vq_xrow_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto p = STEP == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false) : __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return vq_max(__uint_as_float(p[0]), __uint_as_float(p[1]));
}
template <int STEP>
__device__ __forceinline__ int vq_xrow_min(int v) {
  const auto p = STEP == 16 ? __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false)
                            : __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return min((int)p[0], (int)p[1]);
}

// Scores of one 64-row LDS tile against this wave's codes; per row the wave's best (score, lowest code) -> red_s / red_c.
// The LDS tile arrives as a __restrict__ parameter on purpose: after inlining its reads carry alias-scope metadata, which
// keeps the compiler from putting a conservative "wait for every outstanding LDS-DMA" in front of them (the DMA that is
// in flight writes the OTHER buffer; the kernel waits for it explicitly before the barrier that publishes it).
// The codebook is the MFMA's A operand, so lane (r, q) ends up with ONE latent row (r) and 4 NCT codes (16 ct + 4 q + i):
// the row maximum is a per-lane chain, then two cross-row steps (lanes r, r+16, r+32, r+48); the winner's code is the
// lowest one whose score equals the maximum.  hook(k) runs before row tile k (k = 0..3) and after the last one (k = 4):
// the caller hangs the next tile's DMA and the previous tile's output phase there, under this tile's MFMAs.
// Returns |z_row|^2 of row tid / 8 (all 8 threads of the row hold it).
template <int K>
struct VqStep { static constexpr int value = K; };


template <int DP, int NCT, typename Hook>
__device__ __forceinline__ float vq_bs_tile(const char* __restrict__ zb, const u16x8 (&b)[NCT][DP / 32], const f32x4 (&bias)[NCT],
                                            float* __restrict__ red_s, int* __restrict__ red_c, Hook&& hook) {
  constexpr int KS = DP / 32;
  constexpr int CPR = DP / 8;
  constexpr int SWZ_MASK = vq_min(16, CPR) - 1;
  constexpr int TR = VQ_BS_TR;
  static_assert(TR == 64, "four row tiles, five hook points");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // |z_row|^2: 8 threads per row, v_dot2_f32_bf16 over their pieces (any order: the swizzle does not matter), xor butterfly
  float zz = 0.f;
  {
    const int row = tid >> 3, part = tid & 7;
#pragma unroll
    for (int pp = 0; pp < (CPR + 7) / 8; ++pp) {
      const int pos = part + 8 * pp;
      if (CPR % 8 == 0 || pos < CPR) {
        const uint4 u = *reinterpret_cast<const uint4*>(zb + (size_t)row * DP * 2 + (size_t)pos * 16);
        zz = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, u.x), __builtin_bit_cast(bf16x2_t, u.x), zz, false);
        zz = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, u.y), __builtin_bit_cast(bf16x2_t, u.y), zz, false);
        zz = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, u.z), __builtin_bit_cast(bf16x2_t, u.z), zz, false);
        zz = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, u.w), __builtin_bit_cast(bf16x2_t, u.w), zz, false);
      }
    }
    zz += vq_dpp_f<0xB1>(zz);   // quad_perm [1,0,3,2]
    zz += vq_dpp_f<0x4E>(zz);   // quad_perm [2,3,0,1]
    zz += vq_dpp_f<0x141>(zz);  // row_half_mirror: lane l <- lane 7 - l inside each 8 (the other quad's total)
  }

  // operand addresses: row rt * 16 + r has the same swizzle for every rt (16 | SWZ_MASK + 1), and the swizzle only touches
  // the low bits of the piece index, so a lane needs min(KS, 4) addresses per tile; the rest are immediate offsets
  int* wdst = ((q & 1) ? red_c : reinterpret_cast<int*>(red_s)) + wave * TR + r;
  constexpr int NA = KS < 4 ? KS : 4;
  const char* pa[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) pa[j] = zb + (size_t)r * DP * 2 + ((((4 * j + q) & SWZ_MASK) ^ (r & SWZ_MASK)) << 4) + (((4 * j + q) & ~SWZ_MASK) << 4);
  auto scores = [&](int rt, f32x4 (&c)[NCT]) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u16x8 a = *reinterpret_cast<const u16x8*>(pa[s % NA] + rt * 16 * DP * 2 + (s / NA) * NA * 64);
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
        c[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b[ct][s]), __builtin_bit_cast(bf16x8_t, a),
                                                       s == 0 ? bias[ct] : c[ct], 0, 0, 0);
    }
  };
  auto winner = [&](int rt, const f32x4 (&c)[NCT]) {
    float m = c[0][0];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int i = (ct == 0 ? 1 : 0); i < 4; ++i) m = vq_max(m, c[ct][i]);
    m = vq_xrow_max<16>(m);
    m = vq_xrow_max<32>(m);
    int cand = 1 << 20;  // "not mine"
#pragma unroll
    for (int ct = NCT - 1; ct >= 0; --ct)
#pragma unroll
      for (int i = 3; i >= 0; --i) cand = (c[ct][i] == m) ? ct * 16 + i : cand;
    cand += wave * 16 * NCT + 4 * q;
    cand = vq_xrow_min<16>(cand);
    cand = vq_xrow_min<32>(cand);
    // all four lanes of a row hold (m, cand): even q stores the score, odd q the code - one branch-free LDS store
    wdst[rt * 16] = (q & 1) ? cand : __float_as_int(m);
  };
  {
    f32x4 acc[NCT];
    hook(VqStep<0>{});
    scores(0, acc);
    winner(0, acc);
    hook(VqStep<1>{});
    scores(1, acc);
    winner(1, acc);
    hook(VqStep<2>{});
    scores(2, acc);
    winner(2, acc);
    hook(VqStep<3>{});
    scores(3, acc);
    winner(3, acc);
    hook(VqStep<4>{});
  }
  return zz;
}

// the 8 waves' candidates of one row: ascending wave = ascending code, strict > keeps the lowest code on ties
// (__restrict__ parameters for the same reason as vq_bs_tile's: these reads run while the next tile's DMA is in flight)
__device__ __forceinline__ void vq_bs_row_winner(const float* __restrict__ rs, const int* __restrict__ rc, int row, int n_codes,
                                                 float& best, int& code) {
  best = rs[row];
  code = rc[row];
#pragma unroll
  for (int w = 1; w < VQ_BS_WAVES; ++w) {
    const float sc = rs[w * VQ_BS_TR + row];
    const int cc = rc[w * VQ_BS_TR + row];
    const bool better = sc > best;
    best = better ? sc : best;
    code = better ? cc : code;
  }
  code = (unsigned)code < (unsigned)n_codes ? code : 0;  // all-NaN row: no score equals the maximum
}

template <int DP, int NCT>
__global__ __launch_bounds__(VQ_BS_WAVES * 64, 1) void vq_main_bf16_bs(
    const unsigned short* __restrict__ z, const unsigned short* __restrict__ img, int64_t B,
    unsigned short* __restrict__ z_q, int64_t* __restrict__ idx_out, float* __restrict__ err_part) {
  constexpr int KS = DP / 32;
  constexpr int CPR = DP / 8;                       // 16-byte pieces per row
  constexpr int SWZ_MASK = vq_min(16, CPR) - 1;
  constexpr int CT = 64;                            // codes per image chunk (vq_prep_kernel's layout)
  constexpr int CHB = CT * DP * 2 + CT * 4;
  constexpr int TR = VQ_BS_TR;
  constexpr int TILE_BYTES = TR * DP * 2;
  constexpr int THREADS = VQ_BS_WAVES * 64;
  constexpr int PIECES = TR * CPR;                  // 16-byte pieces per tile
  constexpr int NJS = PIECES / THREADS;             // DMA instructions per thread and tile (1, 2 or 4)
  static_assert(PIECES % THREADS == 0 && (NJS == 1 || NJS == 2 || NJS == 4), "D in {64, 128, 256}");
  constexpr int NJ = (CPR + 7) / 8;                 // z_q pieces per thread and row
  constexpr int RED = VQ_BS_WAVES * TR;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* zbuf = smem;                                                   // [2][TILE_BYTES]
  float* red_s = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);      // [2][WAVES][TR]
  int* red_c = reinterpret_cast<int*>(red_s + 2 * RED);                // [2][WAVES][TR]
  float* lds_red = reinterpret_cast<float*>(red_c + 2 * RED);          // [WAVES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const char* gimg = reinterpret_cast<const char*>(img);

  // ---- this wave's codes -> registers (MFMA A operands) + their -|w|^2/2 as the accumulators' initial value
  u16x8 b[NCT][KS];
  f32x4 bias[NCT];  // accumulator rows 4q..4q+3 of tile ct = codes 16 ct + 4 q + i
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    const int c = wave * 16 * NCT + ct * 16 + r;
    const char* row = gimg + (size_t)(c / CT) * CHB + (size_t)(c % CT) * DP * 2;
    const int sw = c & SWZ_MASK;
#pragma unroll
    for (int s = 0; s < KS; ++s) b[ct][s] = *reinterpret_cast<const u16x8*>(row + (((4 * s + q) ^ sw) << 4));
    const int c4 = wave * 16 * NCT + ct * 16 + 4 * q;
    bias[ct] = *reinterpret_cast<const f32x4*>(gimg + (size_t)(c4 / CT) * CHB + (size_t)CT * DP * 2 + (size_t)(c4 % CT) * 4);
  }

  const int64_t ntiles = (B + TR - 1) / TR;
  // DMA instruction j of a tile: LDS slot L = j * THREADS + tid (16 bytes each) holds piece (L % CPR) ^ (row & SWZ_MASK) of
  // row L / CPR: conflict-free operand reads
  auto stage = [&](int64_t tile, int buf, int j) {
    const int L = j * THREADS + tid;
    const int row = L / CPR, pos = L % CPR;
    const int p = pos ^ (row & SWZ_MASK);
    const unsigned short* src;
    if ((tile + 1) * TR <= B) {  // uniform: no per-row clamp on the common path
      src = z + tile * (TR * DP) + (row * DP + p * 8);
    } else {
      int64_t grow = tile * TR + row;
      grow = grow < B ? grow : B - 1;
      src = z + grow * DP + p * 8;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(zbuf + (size_t)buf * TILE_BYTES + (size_t)(L - lane) * 16),
                                     16, 0, 0);
  };

  // output phase of a tile, split so that it can run under the NEXT tile's MFMAs: thread (row = tid / 8, part = tid % 8)
  // finds its row's winner, part 0 writes the index and takes the row's squared error, all 8 fetch pieces part, part + 8,
  // .. of the winner's image row (un-swizzling them) - and, later, store them to z_q
  const int orow = tid >> 3, part = tid & 7;
  float err = 0.f;
  u16x8 ov[NJ];
  auto out_fetch = [&](int64_t tile, int parity, float zz) {
    float best;
    int code;
    vq_bs_row_winner(red_s + parity * RED, red_c + parity * RED, orow, 128 * NCT, best, code);
    const int64_t grow = tile * TR + orow;
    if (part == 0 && grow < B) idx_out[grow] = code;
    const char* wrow = gimg + (size_t)(unsigned)(code * (DP * 2) + (code >> 6) * (CT * 4));  // chunk * CHB + (code % CT) * DP * 2
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int p = (part + 8 * j) % CPR;
      ov[j] = *reinterpret_cast<const u16x8*>(wrow + (size_t)((p ^ (code & SWZ_MASK)) << 4));
    }
    // squared error of the row against the bf16 codebook row z_q returns: |z|^2 - 2 (z.w - |w|^2/2), from numbers the
    // tile already has.  That difference loses its digits when the row sits (almost) on its code; such rows - none on
    // spread-out data, so the branch is wave-uniformly skipped - are summed explicitly from z (re-read) and the fetched row
    float e = __fmaf_rn(-2.0f, best, zz);
    const bool close = grow < B && !(e >= zz * 0.015625f);
    if (__any(close)) {
      float ex = 0.f;
      if (close) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int p = part + 8 * j;
          if (CPR % 8 == 0 || p < CPR) {
            const u16x8 zp = *reinterpret_cast<const u16x8*>(z + grow * DP + p * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const float df = ceeg_bf16_to_f32(ov[j][k]) - ceeg_bf16_to_f32(zp[k]);
              ex = __fmaf_rn(df, df, ex);
            }
          }
        }
      }
      ex += vq_dpp_f<0xB1>(ex);
      ex += vq_dpp_f<0x4E>(ex);
      ex += vq_dpp_f<0x141>(ex);
      e = close ? ex : e;
    }
    if (part == 0 && grow < B) err += e;
  };
  auto out_store = [&](int64_t tile) {
    // the loads are consumed on every path: nothing of them is left in flight behind the conditional stores (a pending
    // load would make the compiler wait for ALL vector memory, the next tile's DMA included, at the next use of a register)
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" ::"v"(ov[j]));
    const int64_t grow = tile * TR + orow;
    if (grow < B) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int p = part + 8 * j;
        if (CPR % 8 == 0 || p < CPR) *reinterpret_cast<u16x8*>(z_q + grow * DP + p * 8) = ov[j];
      }
    }
  };

  int64_t tile = blockIdx.x;
  if (tile < ntiles) {
#pragma unroll
    for (int j = 0; j < NJS; ++j) stage(tile, 0, j);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // first tile landed
#ifdef VQ_BS_TIMING
  unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define VQ_T(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[k] += tn - tprev; tprev = tn; }
#else
#define VQ_T(k)
#endif
  // Per tile: scores -> red[parity]; hung between its row tiles, the next tile's DMA and the PREVIOUS tile's output phase
  // (winner + image fetch, z_q stores two row tiles later); then the wait for the own DMA and ONE barrier.  Safe with one
  // barrier: the buffer a DMA overwrites was last read before the previous barrier; red[parity] is read during the next
  // tile, i.e. before the next barrier, and rewritten only after it.
  float zz_prev = 0.f;
  int64_t prev = -1;
  for (int it = 0; tile < ntiles; ++it, tile += gridDim.x) {
    const int64_t next = tile + gridDim.x;
    const bool has_next = next < ntiles;
    const float zz = vq_bs_tile<DP, NCT>(zbuf + (size_t)(it & 1) * TILE_BYTES, b, bias, red_s + (it & 1) * RED, red_c + (it & 1) * RED,
                                         [&](auto step) {
                                           constexpr int K = decltype(step)::value;
                                           if constexpr (K == 1) {
                                             if (has_next) {
#pragma unroll
                                               for (int j = 0; j < NJS; ++j) stage(next, (it + 1) & 1, j);
                                             }
                                           }
                                           if constexpr (K == 2) {
                                             if (prev >= 0) out_fetch(prev, (it + 1) & 1, zz_prev);
                                           }
                                           if constexpr (K == 4) {
                                             if (prev >= 0) out_store(prev);
                                           }
                                         });
    VQ_T(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next tile has landed
    VQ_T(1)
    __syncthreads();
    VQ_T(2)
    zz_prev = zz;
    prev = tile;
  }
  if (prev >= 0) {  // the last tile's output phase (its red[] was published by the loop's last barrier)
    const int last_parity = (int)(((prev - blockIdx.x) / gridDim.x) & 1);
    out_fetch(prev, last_parity, zz_prev);
    out_store(prev);
  }
  VQ_T(3)
#ifdef VQ_BS_TIMING
  if ((blockIdx.x == 0 || blockIdx.x == 200) && lane == 0 && (wave == 0 || wave == 7))
    printf("blk %d wave %d: tile %llu dmawait %llu barrier %llu drain %llu cycles\n", (int)blockIdx.x, wave, tacc[0], tacc[1], tacc[2], tacc[3]);
#endif
  err = ceeg_wave_sum(err);
  if (lane == 0) lds_red[wave] = err;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < VQ_BS_WAVES; ++w) t += lds_red[w];
    err_part[blockIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------------------ fp32 forward by screening
// The fp32 quantiser's codes, bit for bit, at a fraction of the fp32 matrix time: bf16 MFMA scores of ALL codes with a rigorous
// error bound pick a few candidates per row, and only those get the arithmetic the contract specifies (zz, ee, dot as k-ascending
// fmaf chains, d = fl(fl(zz + ee) - 2 dot), lowest code on ties).
//   s~_c = bf16(z) . bf16(w_c) - ee_c / 2    (v_mfma_f32_16x16x32_bf16, fp32 accumulation, ee_c = the contract's fp32 chain)
//   sigma_c = (zz - d_c) / 2                 (what the contract maximises)
//   |s~_c - sigma_c| <= Delta_row = c1 |z| max_c |w_c| + c2 (|z| + max_c |w_c|)^2 + c3 max_c |w_c|^2 + tiny
//       c1: both operand roundings (2^-9 each, RNE) + the MFMA's product truncation and accumulator rounding + the fp32 chain's own
//           rounding error, all relative to sum_k |z_k w_ck| <= |z| |w_c|;  c2: the two fl() of d;  c3: the accumulator's roundings
//           on the -ee/2 it is seeded with.
//   => the contract's winner c* (and every code tying with it) satisfies  s~_c* >= max_c s~_c - 2 Delta_row.
// Pass 1 finds the row maxima, pass 2 recomputes the same scores (same instructions, same bits) and lists the codes above the
// threshold; a lane then runs the chains of one (row, code) pair.  Whatever the bound cannot handle - more candidates than the
// list holds (near-tie codebooks), a row without a finite maximum (NaN / inf input) - raises a device flag, and the full fp32
// kernel, launched behind this one, runs only if the flag is up (it then rewrites every output): no host decision, and the result
// never depends on how tight the bound is, only the time does.
// One 1024-thread workgroup per CU keeps the whole bf16 codebook in LDS (n_ep * DP * 2 bytes <= 128 KB); a wave owns 32 rows at a time
// (every phase of a unit is a latency chain - operand reads, the pairs' fmaf chains, the output gather - so the CU wants all 16 waves).
constexpr int VQ_SC_WAVES = 16;
constexpr int VQ_SC_RT = 2;                      // 16-row tiles per wave and unit
constexpr int VQ_SC_ROWS = 16 * VQ_SC_RT;        // rows per unit
constexpr int VQ_SC_LIST = 192;                  // candidate entries per wave and unit (expected on spread-out data: ~1.5 per row)

// two fp32 -> packed bf16 with the hardware's round-to-nearest-even conversion (v_cvt_pk_bf16_f32); the screening bound needs
// nothing of it but |relative error| <= 2^-9
typedef __bf16 vq_bf16x2 __attribute__((ext_vector_type(2)));
typedef float vq_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vq_pack_bf16(float lo, float hi) {
  const vq_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, vq_bf16x2));
}

template <int DP>
__global__ __launch_bounds__(VQ_SC_WAVES * 64, 1) void vq_screen_f32(
    const float* __restrict__ z, const float* __restrict__ W, const float* __restrict__ ee, const unsigned short* __restrict__ img16,
    int64_t B, int n_e, int n_ep, float* __restrict__ z_q, int64_t* __restrict__ idx_out, float* __restrict__ err_part,
    int* __restrict__ flag) {
  constexpr int KS = DP / 32;                    // MFMA k-steps
  constexpr int CPR = DP / 8;                    // 16-byte pieces per bf16 row
  constexpr int SWZ_MASK = vq_min(16, CPR) - 1;
  constexpr int RT = VQ_SC_RT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_img = smem;                                                          // [n_ep][DP] bf16, pieces XOR-swizzled per row
  float* lds_bias = reinterpret_cast<float*>(smem + (size_t)n_ep * DP * 2);      // [n_ep]  -ee/2 (padded codes: -inf)
  unsigned* lds_list = reinterpret_cast<unsigned*>(lds_bias + n_ep);             // [WAVES][LIST]  (row in unit) << 16 | code
  unsigned long long* lds_key = reinterpret_cast<unsigned long long*>(lds_list + VQ_SC_WAVES * VQ_SC_LIST);  // [WAVES][ROWS] (ordered d, code)
  int* lds_cnt = reinterpret_cast<int*>(lds_key + VQ_SC_WAVES * VQ_SC_ROWS);      // [WAVES][ROWS]  listed codes per row
  int* lds_total = lds_cnt + VQ_SC_WAVES * VQ_SC_ROWS;                            // [WAVES]  entries in the wave's list
  float* lds_red = reinterpret_cast<float*>(lds_total + VQ_SC_WAVES);             // [WAVES + 1]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;

  // ---- codebook image and biases -> LDS (once per workgroup); max_c |w_c| for the bound
  {
    const uint4* src = reinterpret_cast<const uint4*>(img16);
    uint4* dst = reinterpret_cast<uint4*>(lds_img);
    const int pieces = n_ep * CPR;
    for (int i = tid; i < pieces; i += VQ_SC_WAVES * 64) dst[i] = src[i];
    float wm = 0.f;
    for (int c = tid; c < n_ep; c += VQ_SC_WAVES * 64) {
      const float e = c < n_e ? ee[c] : INFINITY;
      lds_bias[c] = -0.5f * e;
      if (c < n_e) wm = fmaxf(wm, e);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o, 64));
    if (lane == 0) lds_red[wave] = wm;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < VQ_SC_WAVES; ++w) t = fmaxf(t, lds_red[w]);
      lds_red[VQ_SC_WAVES] = t;
    }
    __syncthreads();
  }
  const float wmax = sqrtf(lds_red[VQ_SC_WAVES]) * 1.0001f;   // >= max_c |w_c| (ee is the fp32 chain of |w_c|^2: relative error << 1e-4)
  const bool wfinite = wmax < INFINITY;                      // NaN / inf codebook: the full kernel decides (flag below)

  unsigned* my_list = lds_list + wave * VQ_SC_LIST;
  unsigned long long* my_key = lds_key + wave * VQ_SC_ROWS;
  int* my_cnt = lds_cnt + wave * VQ_SC_ROWS;
  int* my_total = lds_total + wave;
  const int nct = n_ep / 16;
  float err = 0.f;
  bool raise = !wfinite;

  const int64_t nunits = (B + VQ_SC_ROWS - 1) / VQ_SC_ROWS;
  const int64_t stride = (int64_t)gridDim.x * VQ_SC_WAVES;
  for (int64_t unit = (int64_t)blockIdx.x * VQ_SC_WAVES + wave; unit < nunits; unit += stride) {
    const int64_t row0 = unit * VQ_SC_ROWS;
    // somebody already handed the call over to the full kernel (it rewrites every output): no point in finishing the screening
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
    // ---- z -> bf16 MFMA operands: lane (n, q) keeps z~[row n of tile rt][32 s + 8 q .. + 8]; |z|^2 per row for the bound
    u16x8 zb[RT][KS];
    float thr[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      int64_t row = row0 + rt * 16 + n;
      row = row < B ? row : B - 1;
      const float* zr = z + row * DP;
      float n2 = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4 v0 = *reinterpret_cast<const float4*>(zr + 32 * s + 8 * q);
        const float4 v1 = *reinterpret_cast<const float4*>(zr + 32 * s + 8 * q + 4);
        n2 += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        const uint4 pk = make_uint4(vq_pack_bf16(v0.x, v0.y), vq_pack_bf16(v0.z, v0.w), vq_pack_bf16(v1.x, v1.y), vq_pack_bf16(v1.z, v1.w));
        zb[rt][s] = __builtin_bit_cast(u16x8, pk);
      }
      n2 += __shfl_xor(n2, 16, 64);
      n2 += __shfl_xor(n2, 32, 64);
      const float zn = sqrtf(n2) * 1.0001f;                  // >= |z_row|
      // Delta_row (see the header): c1 = 2^-8 * 1.05 (K <= 256); the two fl() of d cost 2^-24 (|z| + |w|)^2 (2^-22 taken), the
      // accumulator's roundings on the -ee/2 it starts from (K/8) 2^-25 |w|^2 (2^-19 taken); an absolute grain covers flushed subnormals
      const float sum = zn + wmax;
      thr[rt] = 2.0f * (0.0041015625f * zn * wmax + 2.384185791015625e-7f * sum * sum + 1.9073486328125e-6f * wmax * wmax + 1e-30f);  // 2 Delta_row for now
    }
    // ---- pass 1: row maxima of the bf16 scores
    float m[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) m[rt] = -INFINITY;
    // two code tiles (ct, ct + 1) at once: 2 RT independent accumulator chains, all operand reads issued up front
    auto scores = [&](int ct, f32x4 (&acc)[2][RT]) {
      u16x8 a[2][KS];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int c = (ct + t) * 16 + n;                       // the A operand row this lane reads: code (ct + t)*16 + (lane & 15)
        const char* wrow = lds_img + (size_t)c * DP * 2;
        const int sw = c & SWZ_MASK;
        const f32x4 bias = *reinterpret_cast<const f32x4*>(lds_bias + (ct + t) * 16 + 4 * q);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[t][rt] = bias;
#pragma unroll
        for (int s = 0; s < KS; ++s) a[t][s] = *reinterpret_cast<const u16x8*>(wrow + (((4 * s + q) ^ sw) << 4));
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[t][s]), __builtin_bit_cast(bf16x8_t, zb[rt][s]),
                                                                acc[t][rt], 0, 0, 0);
    };
    for (int ct = 0; ct < nct; ct += 2) {   // n_ep is a multiple of 32
      f32x4 acc[2][RT];
      scores(ct, acc);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          m[rt] = fmaxf(fmaxf(fmaxf(m[rt], acc[t][rt][0]), fmaxf(acc[t][rt][1], acc[t][rt][2])), acc[t][rt][3]);
    }
    bool bad = false;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      m[rt] = fmaxf(m[rt], __shfl_xor(m[rt], 16, 64));
      m[rt] = fmaxf(m[rt], __shfl_xor(m[rt], 32, 64));
      bad |= !(fabsf(m[rt]) < INFINITY) || !(thr[rt] < INFINITY);   // NaN / inf rows: no finite maximum or no finite bound
      thr[rt] = m[rt] - thr[rt];
    }
    // ---- pass 2: the same scores again; every (row, code) at or above the row's threshold goes to the wave's list
    if (lane < VQ_SC_ROWS) {
      my_key[lane] = ~0ull;
      my_cnt[lane] = 0;
    }
    if (lane == 0) *my_total = 0;
    // The listing of tile pair p runs while the MFMAs of pair p + 1 are in flight (two accumulator sets): a data-dependent branch
    // right behind its own MFMAs would drain the pipeline once per trip.
    bool over = false;
    for (int ct = 0; ct < nct; ct += 2) {
      f32x4 acc[2][RT];
      scores(ct, acc);
      bool hit = false;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          hit |= fmaxf(fmaxf(acc[t][rt][0], acc[t][rt][1]), fmaxf(acc[t][rt][2], acc[t][rt][3])) >= thr[rt];
      if (hit) {   // rare per lane: the slot comes from the wave's LDS counter (the order of the list does not matter: what is
                   // taken from it - a count per row and an integer minimum per row - does not depend on it)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (acc[t][rt][i] >= thr[rt]) {
                const int slot = atomicAdd(my_total, 1);
                if (slot < VQ_SC_LIST) my_list[slot] = ((unsigned)(rt * 16 + n) << 16) | (unsigned)((ct + t) * 16 + 4 * q + i);
                over |= slot >= VQ_SC_LIST;
              }
      }
      if (__any(over)) break;   // the list is full: the unit is handed over below (cnt > VQ_SC_LIST), no point in listing on
    }
    const int cnt = __builtin_amdgcn_readfirstlane(*my_total);
    bad |= cnt > VQ_SC_LIST;
    if (__any(bad)) {   // this unit's rows are left to the full kernel (it rewrites every output once the flag is up)
      raise = true;
      if (lane == 0) atomicOr(flag, 1);   // at once, so that the other waves stop early (integer OR: order-independent)
      continue;
    }
    // ---- a row with ONE listed code is decided (the winner is in the list); the others get the contract's arithmetic, a lane per pair
    for (int b0 = 0; b0 < cnt; b0 += 64)
      if (b0 + lane < cnt) atomicAdd(&my_cnt[my_list[b0 + lane] >> 16], 1);
    for (int b0 = 0; b0 < cnt; b0 += 64) {
      const int e = b0 + lane;
      if (e < cnt && my_cnt[my_list[e] >> 16] == 1) {
        const unsigned ent = my_list[e];
        my_key[ent >> 16] = (unsigned long long)(ent & 0xffffu);
      } else if (e < cnt) {
        const unsigned ent = my_list[e];
        const int code = (int)(ent & 0xffffu);
        int64_t row = row0 + (int)(ent >> 16);
        row = row < B ? row : B - 1;
        const float* zr = z + row * DP;
        const float* wr = W + (size_t)code * DP;
        float zz = 0.f, dot = 0.f;
#pragma unroll 4
        for (int k = 0; k < DP; k += 4) {
          const float4 a = *reinterpret_cast<const float4*>(zr + k);
          const float4 w = *reinterpret_cast<const float4*>(wr + k);
          zz = __fmaf_rn(a.x, a.x, zz); zz = __fmaf_rn(a.y, a.y, zz); zz = __fmaf_rn(a.z, a.z, zz); zz = __fmaf_rn(a.w, a.w, zz);
          dot = __fmaf_rn(a.x, w.x, dot); dot = __fmaf_rn(a.y, w.y, dot); dot = __fmaf_rn(a.z, w.z, dot); dot = __fmaf_rn(a.w, w.w, dot);
        }
        const float d = __fmaf_rn(-2.0f, dot, zz + ee[code]);  // fl(fl(zz + ee) - 2 dot), as vq_main_f32
        if (d == d) {   // a NaN never wins (vq_main_f32's strict <)
          // lowest d, lowest code on ties = the minimum of (d mapped to an order-preserving integer, code); an integer minimum
          // does not depend on the order of the updates
          const unsigned u = __float_as_uint(d);
          const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
          atomicMin(&my_key[ent >> 16], ((unsigned long long)ord << 32) | (unsigned long long)code);
        }
      }
    }
    // (same wave wrote the keys: LDS operations of a wave complete in order)
    {
      const unsigned long long got = lane < VQ_SC_ROWS ? my_key[lane] : 0ull;
      if (__any(got == ~0ull && row0 + lane < B)) {   // a row whose every listed d is NaN: the full kernel's rule applies
        raise = true;
        continue;
      }
      if (lane < VQ_SC_ROWS && row0 + lane < B) idx_out[row0 + lane] = (int64_t)(got & 0xffffffffull);
    }
    // ---- z_q rows (the codebook rows), squared error: lane-consecutive 16-byte pieces of the unit's contiguous block
    constexpr int PPR = DP / 4;
    constexpr int NP = VQ_SC_ROWS * PPR / 64;
#pragma unroll 4
    for (int j = 0; j < NP; ++j) {
      const int p = j * 64 + lane;
      const int lr = p / PPR, c4 = 4 * (p % PPR);
      const int64_t row = row0 + lr;
      if (row < B) {
        const int id = (int)(my_key[lr] & 0xffffffffull);
        const float4 wv = *reinterpret_cast<const float4*>(W + (size_t)id * DP + c4);
        const float4 zv = *reinterpret_cast<const float4*>(z + row * DP + c4);
        *reinterpret_cast<float4*>(z_q + row * DP + c4) = wv;
        const float dx = wv.x - zv.x, dy = wv.y - zv.y, dz = wv.z - zv.z, dw = wv.w - zv.w;
        err = __fmaf_rn(dx, dx, err);
        err = __fmaf_rn(dy, dy, err);
        err = __fmaf_rn(dz, dz, err);
        err = __fmaf_rn(dw, dw, err);
      }
    }
  }
  if (__any(raise) && lane == 0) atomicOr(flag, 1);   // integer OR: order-independent
  err = ceeg_wave_sum(err);
  __syncthreads();
  if (lane == 0) lds_red[wave] = err;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < VQ_SC_WAVES; ++w) t += lds_red[w];
    err_part[blockIdx.x] = t;
  }
}

// bf16 codebook rows for vq_screen_f32: [n_ep][DP], 16-byte pieces XOR-swizzled inside a row (piece p of row c at p ^ (c & mask));
// rows >= n_e and columns >= D are zero
__global__ void vq_screen_prep_kernel(const float* __restrict__ W, int n_e, int D, int n_ep, int DP, unsigned short* __restrict__ img16) {
  const int64_t total = (int64_t)n_ep * DP;
  const int cpr = DP / 8;
  const int mask = vq_min(16, cpr) - 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / DP), k = (int)(i % DP);
    const float w = (c < n_e && k < D) ? W[(size_t)c * D + k] : 0.f;
    img16[(size_t)c * DP + ((((k >> 3) ^ (c & mask)) << 3) | (k & 7))] = ceeg_f32_to_bf16(w);
  }
}

// ------------------------------------------------------------------------------------------ histogram
// slice g covers rows [g*rows_per, (g+1)*rows_per); counts in LDS (integer adds commute: deterministic)
__global__ void vq_hist_kernel(const int64_t* __restrict__ idx, int64_t B, int n_e, int64_t rows_per,
                               uint32_t* __restrict__ part /* [gridDim.x][n_e] */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* h = reinterpret_cast<uint32_t*>(smem);
  for (int c = threadIdx.x; c < n_e; c += blockDim.x) h[c] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * rows_per;
  const int64_t hi = lo + rows_per < B ? lo + rows_per : B;
  for (int64_t b0 = lo + threadIdx.x; b0 < hi; b0 += 8 * (int64_t)blockDim.x) {  // 8 loads in flight
    int64_t k[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + u * (int64_t)blockDim.x;
      k[u] = b < hi ? idx[b] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k[u] >= 0 && k[u] < n_e) atomicAdd(&h[k[u]], 1u);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < n_e; c += blockDim.x) part[(size_t)blockIdx.x * n_e + c] = h[c];
}

// one block: loss, perplexity, hist totals.  Integer counts are added with LDS atomics (commutative: exact
// and order-independent); the float sums go thread -> fixed tree in fp64.
__global__ __launch_bounds__(1024) void vq_finalize_kernel(const float* __restrict__ err_part, int nblk,
                                                           const uint32_t* __restrict__ hpart, int nslice, int n_e, int64_t B,
                                                           int D, float beta, float* __restrict__ loss, float* __restrict__ ppl,
                                                           uint32_t* __restrict__ hist, const float* __restrict__ err_alt, int nblk_alt,
                                                           const int* __restrict__ alt_flag) {
  if (alt_flag && *alt_flag != 0) {  // the screening kernel handed over to the full kernel: its partial sums count
    err_part = err_alt;
    nblk = nblk_alt;
  }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* tot = reinterpret_cast<uint32_t*>(smem);  // [n_e]
  __shared__ double red[1024], red2[1024];
  for (int c = threadIdx.x; c < n_e; c += 1024) tot[c] = 0;
  __syncthreads();
  const int64_t pairs = (int64_t)nslice * n_e;
  for (int64_t i0 = threadIdx.x; i0 < pairs; i0 += 8 * 1024) {  // 8 loads in flight
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = i0 + u * 1024 < pairs ? hpart[i0 + u * 1024] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (v[u]) atomicAdd(&tot[(i0 + u * 1024) % n_e], v[u]);
  }
  double e = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 1024) e += (double)err_part[i];
  __syncthreads();
  double ent = 0.0;
  for (int c = threadIdx.x; c < n_e; c += 1024) {
    const uint32_t t = tot[c];
    if (hist) hist[c] = t;
    const float p = (float)t / (float)B;
    ent += (double)(p * logf(p + 1e-10f));
  }
  red[threadIdx.x] = e;
  red2[threadIdx.x] = ent;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {  // both sums down the same fixed tree
    if ((int)threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      red2[threadIdx.x] += red2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double m = red[0] / ((double)B * (double)D);
    *loss = (float)(m + (double)beta * m);
    *ppl = (float)exp(-red2[0]);
  }
}

// ------------------------------------------------------------------------------------------ backward
template <bool BF16>
__device__ __forceinline__ float vq_ld(const void* p, size_t i) {
  if (BF16) return ceeg_bf16_to_f32(reinterpret_cast<const unsigned short*>(p)[i]);
  return reinterpret_cast<const float*>(p)[i];
}

// dz = g + s*(2/M)*(z - W[idx])
template <bool BF16>
__global__ void vq_dz_kernel(const void* __restrict__ z, const float* __restrict__ W, const int64_t* __restrict__ idx,
                             const void* __restrict__ g, const float* __restrict__ s_dev, int64_t B, int D,
                             void* __restrict__ dz) {
  const float coef = s_dev[0] * (2.0f / ((float)B * (float)D));
  const int64_t total = B * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / D;
    const int k = (int)(i - b * D);
    const float zv = vq_ld<BF16>(z, i);
    const float w = W[(size_t)idx[b] * D + k];
    float v = coef * (zv - w);
    if (g) v += vq_ld<BF16>(g, i);
    if (BF16) reinterpret_cast<unsigned short*>(dz)[i] = ceeg_f32_to_bf16(v);
    else reinterpret_cast<float*>(dz)[i] = v;
  }
}

// fp32, D % 4 == 0, 16-byte aligned operands: one float4 per lane (same arithmetic per element as above)
__global__ __launch_bounds__(256) void vq_dz4_kernel(const float* __restrict__ z, const float* __restrict__ W, const int64_t* __restrict__ idx,
                                                     const float* __restrict__ g, const float* __restrict__ s_dev, int64_t B, int D,
                                                     float* __restrict__ dz) {
  const float coef = s_dev[0] * (2.0f / ((float)B * (float)D));
  const int dq = D / 4;
  const int64_t total = B * dq;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / dq;
    const int k = (int)(i - b * dq) * 4;
    const float4 zv = *reinterpret_cast<const float4*>(z + b * D + k);
    const float4 w = *reinterpret_cast<const float4*>(W + (size_t)idx[b] * D + k);
    float4 v;
    v.x = coef * (zv.x - w.x); v.y = coef * (zv.y - w.y); v.z = coef * (zv.z - w.z); v.w = coef * (zv.w - w.w);
    if (g) {
      const float4 gv = *reinterpret_cast<const float4*>(g + b * D + k);
      v.x += gv.x; v.y += gv.y; v.z += gv.z; v.w += gv.w;
    }
    *reinterpret_cast<float4*>(dz + b * D + k) = v;
  }
}

// scan of the per-slice histograms down the slices, 64 codes per workgroup and 16 slice groups per code (the one-block form of
// this kernel took as long as the scatter it feeds): off[g][c] = rows with code c in slices < g, tot[c] = rows with code c.
constexpr int VQ_SCAN_GROUPS = 16;
__global__ __launch_bounds__(64 * VQ_SCAN_GROUPS) void vq_scan_kernel(const uint32_t* __restrict__ hpart, int nslice, int n_e,
                                                                      uint32_t* __restrict__ off, uint32_t* __restrict__ tot) {
  __shared__ uint32_t part[VQ_SCAN_GROUPS][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int per = (nslice + VQ_SCAN_GROUPS - 1) / VQ_SCAN_GROUPS;
  const int g_lo = q * per, g_hi = g_lo + per < nslice ? g_lo + per : nslice;
  uint32_t t = 0;
  if (c < n_e) {
#pragma unroll 8
    for (int g = g_lo; g < g_hi; ++g) t += hpart[(size_t)g * n_e + c];
  }
  part[q][cl] = t;
  __syncthreads();
  uint32_t run = 0;
  for (int p = 0; p < q; ++p) run += part[p][cl];
  if (c >= n_e) return;
  if (q == VQ_SCAN_GROUPS - 1) tot[c] = run + t;
  for (int g0 = g_lo; g0 < g_hi; g0 += 8) {  // 8 loads in flight
    uint32_t h[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) h[u] = g0 + u < g_hi ? hpart[(size_t)(g0 + u) * n_e + c] : 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (g0 + u < g_hi) off[(size_t)(g0 + u) * n_e + c] = run;
      run += h[u];
    }
  }
}

// stable scatter: one wave per slice walks its rows in order, 64 at a time.  The lanes holding the same code find
// each other with one ballot per code bit (no loop over the distinct codes of the batch); equal codes are ranked
// by lane order -> sorted[] lists the rows of each code in ascending row order (deterministic).  Every wave first scans tot[] over
// the codes (first row of each code = cursor base); the wave of slice 0 also writes that scan and the per-code chunk table of the
// segmented sum for the kernels behind it.
__global__ __launch_bounds__(64) void vq_scatter_kernel(const int64_t* __restrict__ idx, int64_t B, int n_e, int nbits, int64_t rows_per,
                                                        const uint32_t* __restrict__ off, const uint32_t* __restrict__ tot,
                                                        uint32_t* __restrict__ sorted, uint32_t* __restrict__ code_base /* [n_e+1] */,
                                                        uint32_t* __restrict__ chunk_base /* [n_e+1] */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* cur = reinterpret_cast<uint32_t*>(smem);  // [n_e]
  const int lane = threadIdx.x;
  const bool first = blockIdx.x == 0;
  uint32_t run = 0, crun = 0;
  for (int c0 = 0; c0 < n_e; c0 += 64) {
    const int c = c0 + lane;
    const uint32_t t = c < n_e ? tot[c] : 0u;
    const uint32_t ch = (t + VQ_SEG - 1) / VQ_SEG;
    uint32_t inc = t, cinc = ch;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t a = __shfl_up(inc, o, 64), b = __shfl_up(cinc, o, 64);
      if (lane >= o) { inc += a; cinc += b; }
    }
    if (c < n_e) {
      cur[c] = run + inc - t + off[(size_t)blockIdx.x * n_e + c];
      if (first) {
        code_base[c] = run + inc - t;
        chunk_base[c] = crun + cinc - ch;
      }
    }
    run += __shfl(inc, 63, 64);
    crun += __shfl(cinc, 63, 64);
  }
  if (first && lane == 0) {
    code_base[n_e] = run;
    chunk_base[n_e] = crun;
  }
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * rows_per;
  const int64_t hi = lo + rows_per < B ? lo + rows_per : B;
  const unsigned long long below = (1ull << lane) - 1ull;
  int64_t k_next = lo + lane < hi ? idx[lo + lane] : -1;
  for (int64_t b0 = lo; b0 < hi; b0 += 64) {
    const int64_t b = b0 + lane;
    int k = (int)k_next;
    if (k_next < 0 || k_next >= n_e) k = -1;
    k_next = b + 64 < hi ? idx[b + 64] : -1;  // next batch in flight
    unsigned long long peers = __ballot(k >= 0);
    for (int bit = 0; bit < nbits; ++bit) {
      const bool one = (k >> bit) & 1;
      const unsigned long long m = __ballot(one);
      peers &= one ? m : ~m;
    }
    if (k >= 0) {
      const uint32_t base = cur[k];
      sorted[base + (uint32_t)__popcll(peers & below)] = (uint32_t)b;
    }
    __syncthreads();  // single-wave block: every lane has read cur[] before the leaders advance it
    if (k >= 0 && (peers & below) == 0) cur[k] += (uint32_t)__popcll(peers);
    __syncthreads();
  }
}

// segmented sum of W[c] - z[row]: one wave per (code, chunk of <= VQ_SEG sorted rows); rows are added in list order.  The
// DIFFERENCES are summed (as autograd does for the reference), not count * W[c] - sum z: on a trained codebook the rows sit close to
// their code and that form cancels to a few per cent of relative error in fp32.
// Lane l owns elements l, l+64, ... of the row (D <= 256: up to 4); the chunk's row ids arrive 64 at a time in one
// lane-parallel load and are broadcast from there, 8 rows' loads in flight.
template <bool BF16>
__global__ __launch_bounds__(256) void vq_segsum_kernel(const void* __restrict__ z, const float* __restrict__ W,
                                                        const uint32_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ code_base, const uint32_t* __restrict__ chunk_base,
                                                        int n_e, int D, float* __restrict__ partial /* [nchunks][D] */) {
  const int lane = threadIdx.x & 63;
  const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= chunk_base[n_e]) return;
  int lo = 0, hi = n_e;  // last code c with chunk_base[c] <= w
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_base[mid] <= w) lo = mid; else hi = mid;
  }
  const int c = lo;
  const uint32_t p0 = code_base[c] + (w - chunk_base[c]) * VQ_SEG;
  const uint32_t p1 = p0 + VQ_SEG < code_base[c + 1] ? p0 + VQ_SEG : code_base[c + 1];
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, wc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) wc[t] = (lane + 64 * t < D) ? W[(size_t)c * D + lane + 64 * t] : 0.f;
  for (uint32_t pb = p0; pb < p1; pb += 64) {
    const int n = (int)(p1 - pb < 64u ? p1 - pb : 64u);
    const int my = lane < n ? (int)sorted[pb + lane] : 0;
    for (int j0 = 0; j0 < n; j0 += 8) {
      float v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u < n ? j0 + u : n - 1;  // wave-uniform; the tail re-reads the last row and adds nothing
        const size_t row = (size_t)(uint32_t)__builtin_amdgcn_readlane(my, j) * D;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[u][t] = (lane + 64 * t < D) ? vq_ld<BF16>(z, row + lane + 64 * t) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (j0 + u < n) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] += wc[t] - v[u][t];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (lane + 64 * t < D) partial[(size_t)w * D + lane + 64 * t] = acc[t];
}

// dW[c] = s*beta*(2/M) * (sum of the code's chunk partials of W[c] - z[row], in chunk order)
__global__ void vq_dw_kernel(const float* __restrict__ partial, const uint32_t* __restrict__ chunk_base, const float* __restrict__ s_dev,
                             int64_t B, int n_e, int D, float beta, float* __restrict__ dW) {
  const int c = blockIdx.x;
  const float coef = s_dev[0] * beta * (2.0f / ((float)B * (float)D));
  const uint32_t w0 = chunk_base[c], w1 = chunk_base[c + 1];
  for (int k = threadIdx.x; k < D; k += blockDim.x) {
    float sum = 0.f;
    for (uint32_t w = w0; w < w1; w += 16) {  // 16 partials in flight, added in chunk order (a code that took many rows has many chunks)
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = w + u < w1 ? partial[(size_t)(w + u) * D + k] : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (w + u < w1) sum += v[u];
    }
    dW[(size_t)c * D + k] = coef * sum;
  }
}

int vq_dp(int D, int dtype) {
  const int lo = dtype == CEEG_BF16 ? 32 : 16;
  int dp = lo;
  while (dp < D) dp <<= 1;
  return dp;
}
int vq_ct(int dtype) { return dtype == CEEG_BF16 ? 64 : 32; }
int vq_bm(int dtype) { return dtype == CEEG_BF16 ? 256 : 128; }
int64_t vq_nslice(int64_t B) { return (B + VQ_HIST_ROWS - 1) / VQ_HIST_ROWS; }

struct VqFwdWs {
  float* ee;
  void* img;
  float* err_part;
  uint32_t* hpart;
  unsigned short* img16 = nullptr;  // fp32 forward by screening (vq_screen_f32): bf16 codebook rows, its partial sums, its flag
  float* err_sc = nullptr;
  int* flag = nullptr;
};
// shapes vq_screen_f32 takes: fp32, D a power of two >= 32 (= DP), the bf16 codebook within 128 KB of LDS
bool vq_screen_shape(int n_e, int D, int dtype) {
  if (dtype != CEEG_F32 || D < 32 || (D & (D - 1)) != 0 || D > 256) return false;
  const int n_ep = (n_e + 31) / 32 * 32;
  return (size_t)n_ep * D * 2 <= (size_t)128 * 1024 && n_e < 65536;
}
size_t vq_fwd_layout(int64_t B, int n_e, int D, int dtype, void* ws, size_t ws_bytes, VqFwdWs* out) {
  const int DP = vq_dp(D, dtype), CT = vq_ct(dtype);
  const int n_ep = (n_e + CT - 1) / CT * CT;
  const int64_t nblk = (B + 127) / 128 + 256;  // upper bound over the kernels' partial-sum slots
  CeegWs a(ws, ws_bytes);
  size_t need = ceeg_ws_part(n_ep, 4) + ceeg_ws_part((size_t)n_ep * DP * (dtype == CEEG_BF16 ? 2 : 4) + (size_t)n_ep * 4, 1) +
                ceeg_ws_part(nblk, 4) + ceeg_ws_part((size_t)VQ_FWD_SLICES * n_e, 4);
  const bool sc = vq_screen_shape(n_e, D, dtype);
  if (sc) need += ceeg_ws_part((size_t)n_ep * DP, 2) + ceeg_ws_part(256, 4) + ceeg_ws_part(1, 4);
  if (out) {
    out->ee = a.take<float>(n_ep);
    out->img = a.take<char>((size_t)n_ep * DP * (dtype == CEEG_BF16 ? 2 : 4) + (size_t)n_ep * 4);
    out->err_part = a.take<float>(nblk);
    out->hpart = a.take<uint32_t>((size_t)VQ_FWD_SLICES * n_e);
    if (sc) {
      out->img16 = a.take<unsigned short>((size_t)n_ep * DP);
      out->err_sc = a.take<float>(256);
      out->flag = a.take<int>(1);
    }
    if (!a.ok) return 0;
  }
  return need;
}

bool vq_screen_on() {  // read per call: CEEG_VQ_SCREEN=0 keeps the full fp32 kernel alone (the checker of the screened form)
  const char* e = getenv("CEEG_VQ_SCREEN");
  return !(e && e[0] == '0');
}
// returns the number of workgroups (= partial sums) of the screening launch, 0 when the shape is not taken
template <int DP>
int launch_screen(const float* z, const float* W, const VqFwdWs& w, int64_t B, int n_e, int D, int n_ep, float* z_q, int64_t* idx,
                  hipStream_t st) {
  if constexpr (DP < 32) {
    return 0;
  } else {
    if (D != DP || !w.img16 || !vq_screen_on()) return 0;
    const int64_t nunits = (B + VQ_SC_ROWS - 1) / VQ_SC_ROWS;
    const int64_t want = (nunits + VQ_SC_WAVES - 1) / VQ_SC_WAVES;
    const int grid = (int)(want < 256 ? want : 256);   // one persistent workgroup per CU
    const size_t lds = (size_t)n_ep * DP * 2 + (size_t)n_ep * 4 + (size_t)VQ_SC_WAVES * VQ_SC_LIST * 4 + (size_t)VQ_SC_WAVES * VQ_SC_ROWS * 12 +
                       (2 * VQ_SC_WAVES + 1) * 4;
    const int64_t total = (int64_t)n_ep * DP;
    if (hipMemsetAsync(w.flag, 0, sizeof(int), st) != hipSuccess) return -1;
    hipLaunchKernelGGL(vq_screen_prep_kernel, dim3((unsigned)((total + 255) / 256 < 512 ? (total + 255) / 256 : 512)), dim3(256), 0, st, W, n_e, D,
                       n_ep, DP, w.img16);
    hipLaunchKernelGGL((vq_screen_f32<DP>), dim3((unsigned)grid), dim3(VQ_SC_WAVES * 64), lds, st, z, W, w.ee, w.img16, B, n_e, n_ep, z_q, idx,
                       w.err_sc, w.flag);
    return grid;
  }
}
template <int DP>
void launch_f32(const float* z, const float* W, const VqFwdWs& w, int64_t B, int n_e, int D, int n_ep, float* z_q,
                int64_t* idx, int64_t nblk, hipStream_t st, const int* run_flag) {
  size_t lds = 2 * (32 * DP * 4 + 32 * 4) + 128 * 4 + 16;
  if (D == DP && DP <= 128) {  // per-wave z strips (VIA_LDS) overlay everything after chunk buffer 0
    const size_t with_strips = (32 * DP * 4 + 32 * 4) + (size_t)VQ_WAVES * 16 * (DP + 4) * 4;
    lds = lds > with_strips ? lds : with_strips;
  }
  if (D == DP)
    hipLaunchKernelGGL((vq_main_f32<DP, true>), dim3((unsigned)nblk), dim3(VQ_THREADS), lds, st, z, W, w.ee,
                       reinterpret_cast<const float*>(w.img), B, n_e, D, n_ep, z_q, idx, w.err_part, run_flag);
  else
    hipLaunchKernelGGL((vq_main_f32<DP, false>), dim3((unsigned)nblk), dim3(VQ_THREADS), lds, st, z, W, w.ee,
                       reinterpret_cast<const float*>(w.img), B, n_e, D, n_ep, z_q, idx, w.err_part, run_flag);
}
bool vq_bs_off() {  // checker's switch (read per call): CEEG_VQ_BF16_BS=0 selects the row-stationary kernel
  const char* e = getenv("CEEG_VQ_BF16_BS");
  return e && e[0] == '0';
}
template <int DP, int NCT>
int64_t launch_bf16_bs(const void* z, const VqFwdWs& w, int64_t B, void* z_q, int64_t* idx, hipStream_t st) {
  const int64_t ntiles = (B + VQ_BS_TR - 1) / VQ_BS_TR;
  const unsigned grid = (unsigned)(ntiles < 256 ? ntiles : 256);  // one persistent workgroup per CU
  const size_t lds = 2 * (size_t)VQ_BS_TR * DP * 2 + 2 * (size_t)VQ_BS_WAVES * VQ_BS_TR * 8 + VQ_BS_WAVES * 4;
  hipLaunchKernelGGL((vq_main_bf16_bs<DP, NCT>), dim3(grid), dim3(VQ_BS_WAVES * 64), lds, st, reinterpret_cast<const unsigned short*>(z),
                     reinterpret_cast<const unsigned short*>(w.img), B, reinterpret_cast<unsigned short*>(z_q), idx, w.err_part);
  return (int64_t)grid;
}
// returns the number of err_part slots the launched kernel fills
template <int DP>
int64_t launch_bf16(const void* z, const float* W, const VqFwdWs& w, int64_t B, int n_e, int D, int n_ep, void* z_q,
                    int64_t* idx, int64_t nblk, hipStream_t st) {
  if constexpr (DP >= 64) {
    if (D == DP && !vq_bs_off()) {  // whole codebook in one workgroup's registers
      if (n_ep == 128) return launch_bf16_bs<DP, 1>(z, w, B, z_q, idx, st);
      if (n_ep == 256) return launch_bf16_bs<DP, 2>(z, w, B, z_q, idx, st);
      if (n_ep == 512) return launch_bf16_bs<DP, 4>(z, w, B, z_q, idx, st);
    }
  }
  const size_t lds = 2 * (64 * DP * 2 + 64 * 4) + 256 * 4 + VQ_WAVES * 4;
  if (D == DP)
    hipLaunchKernelGGL((vq_main_bf16<DP, true>), dim3((unsigned)nblk), dim3(VQ_THREADS), lds, st,
                       reinterpret_cast<const unsigned short*>(z), W, w.ee, reinterpret_cast<const unsigned short*>(w.img), B,
                       n_e, D, n_ep, reinterpret_cast<unsigned short*>(z_q), idx, w.err_part);
  else
    hipLaunchKernelGGL((vq_main_bf16<DP, false>), dim3((unsigned)nblk), dim3(VQ_THREADS), lds, st,
                       reinterpret_cast<const unsigned short*>(z), W, w.ee, reinterpret_cast<const unsigned short*>(w.img), B,
                       n_e, D, n_ep, reinterpret_cast<unsigned short*>(z_q), idx, w.err_part);
  return nblk;
}

}  // namespace

extern "C" size_t ceeg_vq_forward_ws_bytes(int64_t B, int n_e, int D, int dtype) {
  if (B <= 0 || n_e <= 0 || D <= 0 || D > 256) return 0;
  return vq_fwd_layout(B, n_e, D, dtype, nullptr, 0, nullptr);
}

namespace {
// argument checks shared by the two forward entry points (before any launch)
int vq_forward_check(const void* z, const float* W, int64_t B, int n_e, int D, int dtype, const void* z_q, const int64_t* idx,
                     const float* loss, const float* perplexity) {
  if (!z || !W || !z_q || !idx || !loss || !perplexity || B <= 0 || n_e <= 0 || D <= 0) return CEEG_EINVAL;
  if (dtype != CEEG_F32 && dtype != CEEG_BF16) return CEEG_EUNSUPPORTED;
  if (D > 256 || n_e > 4096 || B > ((int64_t)1 << 31) * 128) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(z, 16) || !ceeg_aligned(z_q, 16) || !ceeg_aligned(W, 16)) return CEEG_EINVAL;
  return CEEG_OK;
}

// the forward's launches over an already carved workspace `w` (every word of it that a kernel reads is written by an
// earlier kernel of the same call: nothing has to be zero on entry)
int vq_forward_run(const void* z, const float* W, int64_t B, int n_e, int D, int dtype, float beta, void* z_q, int64_t* idx,
                   float* loss, float* perplexity, uint32_t* hist, const VqFwdWs& w, hipStream_t st) {
  const int DP = vq_dp(D, dtype), CT = vq_ct(dtype);
  const int n_ep = (n_e + CT - 1) / CT * CT;
  int64_t nblk = (B + vq_bm(dtype) - 1) / vq_bm(dtype);
  int n_sc = 0;  // partial sums of the screening launch (fp32 only)

  const int prep_blocks = (int)(((int64_t)n_ep * DP + 255) / 256 < 1024 ? ((int64_t)n_ep * DP + 255) / 256 : 1024);
  if (dtype == CEEG_BF16)
    hipLaunchKernelGGL(vq_prep_kernel<true>, dim3(prep_blocks > (n_ep + 255) / 256 ? prep_blocks : (n_ep + 255) / 256), dim3(256), 0, st, W, n_e, D, n_ep, DP, CT, w.ee, static_cast<char*>(w.img));
  else
    hipLaunchKernelGGL(vq_prep_kernel<false>, dim3(prep_blocks > (n_ep + 255) / 256 ? prep_blocks : (n_ep + 255) / 256), dim3(256), 0, st, W, n_e, D, n_ep, DP, CT, w.ee, static_cast<char*>(w.img));
  CEEG_CHECK_LAUNCH();

  if (dtype == CEEG_F32) {
    const float* zf = static_cast<const float*>(z);
    float* zq = static_cast<float*>(z_q);
    // screened form first where the shape allows (vq_screen_f32); the full kernel behind it then runs only if the flag went up
#define VQ_F32_CASE(P)                                                                                                  \
  case P:                                                                                                               \
    n_sc = launch_screen<P>(zf, W, w, B, n_e, D, n_ep, zq, idx, st);                                                    \
    if (n_sc < 0) return CEEG_ELAUNCH;                                                                                  \
    launch_f32<P>(zf, W, w, B, n_e, D, n_ep, zq, idx, nblk, st, n_sc > 0 ? w.flag : nullptr);                            \
    break;
    switch (DP) {
      VQ_F32_CASE(16)
      VQ_F32_CASE(32)
      VQ_F32_CASE(64)
      VQ_F32_CASE(128)
      VQ_F32_CASE(256)
      default: return CEEG_EUNSUPPORTED;
    }
#undef VQ_F32_CASE
  } else {
    switch (DP) {
      case 32: nblk = launch_bf16<32>(z, W, w, B, n_e, D, n_ep, z_q, idx, nblk, st); break;
      case 64: nblk = launch_bf16<64>(z, W, w, B, n_e, D, n_ep, z_q, idx, nblk, st); break;
      case 128: nblk = launch_bf16<128>(z, W, w, B, n_e, D, n_ep, z_q, idx, nblk, st); break;
      case 256: nblk = launch_bf16<256>(z, W, w, B, n_e, D, n_ep, z_q, idx, nblk, st); break;
      default: return CEEG_EUNSUPPORTED;
    }
  }
  CEEG_CHECK_LAUNCH();

  // forward histogram: at most VQ_FWD_SLICES slices (the finalize kernel reads nslice*n_e counts)
  int64_t rows_per = (B + VQ_FWD_SLICES - 1) / VQ_FWD_SLICES;
  rows_per = rows_per < VQ_HIST_ROWS ? VQ_HIST_ROWS : (rows_per + 255) / 256 * 256;
  const int nslice = (int)((B + rows_per - 1) / rows_per);
  hipLaunchKernelGGL(vq_hist_kernel, dim3(nslice), dim3(256), (size_t)n_e * 4, st, idx, B, n_e, rows_per, w.hpart);
  CEEG_CHECK_LAUNCH();
  if (n_sc > 0)  // the screening kernel's sums, unless its flag says the full kernel took over
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(1024), (size_t)n_e * 4, st, w.err_sc, n_sc, w.hpart, nslice, n_e, B, D, beta, loss,
                       perplexity, hist, w.err_part, (int)nblk, w.flag);
  else
    hipLaunchKernelGGL(vq_finalize_kernel, dim3(1), dim3(1024), (size_t)n_e * 4, st, w.err_part, (int)nblk, w.hpart, nslice, n_e, B, D, beta,
                       loss, perplexity, hist, static_cast<const float*>(nullptr), 0, static_cast<const int*>(nullptr));
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}
}  // namespace

extern "C" int ceeg_vq_forward(const void* z, const float* W, int64_t B, int n_e, int D, int dtype, float beta,
                               void* z_q, int64_t* idx, float* loss, float* perplexity, uint32_t* hist,
                               void* ws, size_t ws_bytes, void* stream) {
  const int rc = vq_forward_check(z, W, B, n_e, D, dtype, z_q, idx, loss, perplexity);
  if (rc != CEEG_OK) return rc;
  VqFwdWs w;
  if (ws_bytes < vq_fwd_layout(B, n_e, D, dtype, nullptr, 0, nullptr)) return CEEG_EWORKSPACE;
  if (!ws || !ceeg_aligned(ws, 256) || vq_fwd_layout(B, n_e, D, dtype, ws, ws_bytes, &w) == 0) return CEEG_EWORKSPACE;
  return vq_forward_run(z, W, B, n_e, D, dtype, beta, z_q, idx, loss, perplexity, hist, w, static_cast<hipStream_t>(stream));
}

// Test / measurement helper: did the screened form (vq_screen_f32) decide the last fp32 forward that used this workspace, or did it
// hand over to the full kernel?  *out = 1 screened, 0 handed over, -1 the shape (or CEEG_VQ_SCREEN=0) does not screen.  Synchronises.
extern "C" int ceeg_vq_forward_screened(const void* ws, size_t ws_bytes, int64_t B, int n_e, int D, int dtype, int* out, void* stream) {
  if (!ws || !out || B <= 0 || n_e <= 0 || D <= 0 || D > 256) return CEEG_EINVAL;
  *out = -1;
  if (!vq_screen_shape(n_e, D, dtype) || D != vq_dp(D, dtype) || !vq_screen_on()) return CEEG_OK;
  VqFwdWs w;
  if (vq_fwd_layout(B, n_e, D, dtype, const_cast<void*>(ws), ws_bytes, &w) == 0) return CEEG_EWORKSPACE;
  int flag = 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(&flag, w.flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return CEEG_ELAUNCH;
  *out = flag ? 0 : 1;
  return CEEG_OK;
}

// ---------------------------------------------------------------------------------------------- forward without a workspace
// For callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc): the forward's
// scratch (codebook image, code norms, per-block error partials, per-slice histograms) lives in a file-scope device array.
// Every GPU the library runs on holds its own copy at its own address; host code never uses the array's NAME as a pointer
// (it is a host-side stand-in) - the device address comes from hipGetSymbolAddress, once per device, and only that pointer
// goes to kernels.
namespace {
constexpr size_t VQ_STATIC_SCRATCH_BYTES = (size_t)4 << 20;  // covers config 3 (4 Mi x 512 x 256: 0.63 MiB in bf16, 0.88 MiB in fp32)
static __device__ __align__(256) unsigned char vq_static_scratch[VQ_STATIC_SCRATCH_BYTES];
constexpr int VQ_MAX_DEVICES = 64;
void* g_scratch_ptr[VQ_MAX_DEVICES];  // per device; written once under g_scratch_mu
std::mutex g_scratch_mu;

// device address of THIS device's copy of the array (looked up on the first call on the device, with that device current)
int vq_static_scratch_ptr(void** out) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= VQ_MAX_DEVICES) return CEEG_ENODEVICE;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(vq_static_scratch)) != hipSuccess || p == nullptr) return CEEG_ENODEVICE;
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return CEEG_OK;
}
}  // namespace

extern "C" size_t ceeg_vq_forward_static_max_bytes(void) { return VQ_STATIC_SCRATCH_BYTES; }

extern "C" int ceeg_vq_forward_static_prepare(void) {  // the per-device set-up of ceeg_vq_forward_static, callable on its own
  void* scratch = nullptr;
  return vq_static_scratch_ptr(&scratch);
}

extern "C" int ceeg_vq_forward_static(const void* z, const float* W, int64_t B, int n_e, int D, int dtype, float beta,
                                      void* z_q, int64_t* idx, float* loss, float* perplexity, uint32_t* hist, void* stream) {
  const int rc = vq_forward_check(z, W, B, n_e, D, dtype, z_q, idx, loss, perplexity);
  if (rc != CEEG_OK) return rc;
  if (vq_fwd_layout(B, n_e, D, dtype, nullptr, 0, nullptr) > VQ_STATIC_SCRATCH_BYTES) return CEEG_ETOOBIG;
  void* scratch = nullptr;
  const int rp = vq_static_scratch_ptr(&scratch);
  if (rp != CEEG_OK) return rp;
  VqFwdWs w;
  if (vq_fwd_layout(B, n_e, D, dtype, scratch, VQ_STATIC_SCRATCH_BYTES, &w) == 0) return CEEG_ETOOBIG;
  return vq_forward_run(z, W, B, n_e, D, dtype, beta, z_q, idx, loss, perplexity, hist, w, static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------------- backward API
namespace {
struct VqBwdWs {
  uint32_t *hpart, *off, *tot, *code_base, *chunk_base, *sorted;
  float* partial;
};
size_t vq_bwd_layout(int64_t B, int n_e, int D, void* ws, size_t ws_bytes, VqBwdWs* out) {
  const int64_t nslice = vq_nslice(B);
  const int64_t nchunk_max = B / VQ_SEG + n_e + 1;
  size_t need = 2 * ceeg_ws_part((size_t)nslice * n_e, 4) + 3 * ceeg_ws_part(n_e + 1, 4) + ceeg_ws_part(B, 4) +
                ceeg_ws_part((size_t)nchunk_max * D, 4);
  if (out) {
    CeegWs a(ws, ws_bytes);
    out->hpart = a.take<uint32_t>((size_t)nslice * n_e);
    out->off = a.take<uint32_t>((size_t)nslice * n_e);
    out->tot = a.take<uint32_t>(n_e + 1);
    out->code_base = a.take<uint32_t>(n_e + 1);
    out->chunk_base = a.take<uint32_t>(n_e + 1);
    out->sorted = a.take<uint32_t>(B);
    out->partial = a.take<float>((size_t)nchunk_max * D);
    if (!a.ok) return 0;
  }
  return need;
}
}  // namespace

extern "C" size_t ceeg_vq_backward_ws_bytes(int64_t B, int n_e, int D, int dtype) {
  (void)dtype;
  if (B <= 0 || n_e <= 0 || D <= 0) return 0;
  return vq_bwd_layout(B, n_e, D, nullptr, 0, nullptr);
}

extern "C" int ceeg_vq_backward(const void* z, const float* W, const int64_t* idx, const void* g, const float* s_dev,
                                int64_t B, int n_e, int D, int dtype, float beta, void* dz, float* dW,
                                void* ws, size_t ws_bytes, void* stream) {
  if (!z || !W || !idx || !s_dev || B <= 0 || n_e <= 0 || D <= 0) return CEEG_EINVAL;
  if (dtype != CEEG_F32 && dtype != CEEG_BF16) return CEEG_EUNSUPPORTED;
  if (D > 256 || n_e > 4096 || B >= ((int64_t)1 << 32)) return CEEG_EUNSUPPORTED;
  hipStream_t st = static_cast<hipStream_t>(stream);
  VqBwdWs w;
  if (dW) {  // every argument is validated before the first launch
    if (ws_bytes < vq_bwd_layout(B, n_e, D, nullptr, 0, nullptr)) return CEEG_EWORKSPACE;
    if (!ws || !ceeg_aligned(ws, 256) || vq_bwd_layout(B, n_e, D, ws, ws_bytes, &w) == 0) return CEEG_EWORKSPACE;
  }
  if (dz) {
    const int64_t total = B * D;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == CEEG_BF16)
      hipLaunchKernelGGL(vq_dz_kernel<true>, dim3(blocks), dim3(256), 0, st, z, W, idx, g, s_dev, B, D, dz);
    else if ((D % 4) == 0 && ceeg_aligned(z, 16) && ceeg_aligned(W, 16) && ceeg_aligned(dz, 16) && (!g || ceeg_aligned(g, 16)))
      hipLaunchKernelGGL(vq_dz4_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(z), W, idx, static_cast<const float*>(g),
                         s_dev, B, D, static_cast<float*>(dz));
    else
      hipLaunchKernelGGL(vq_dz_kernel<false>, dim3(blocks), dim3(256), 0, st, z, W, idx, g, s_dev, B, D, dz);
    CEEG_CHECK_LAUNCH();
  }
  if (dW) {
    const int nslice = (int)vq_nslice(B);
    hipLaunchKernelGGL(vq_hist_kernel, dim3(nslice), dim3(256), (size_t)n_e * 4, st, idx, B, n_e, (int64_t)VQ_HIST_ROWS, w.hpart);
    CEEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(vq_scan_kernel, dim3((unsigned)((n_e + 63) / 64)), dim3(64 * VQ_SCAN_GROUPS), 0, st, w.hpart, nslice, n_e, w.off, w.tot);
    CEEG_CHECK_LAUNCH();
    int nbits = 0;
    while ((1 << nbits) < n_e) ++nbits;
    hipLaunchKernelGGL(vq_scatter_kernel, dim3(nslice), dim3(64), (size_t)n_e * 4, st, idx, B, n_e, nbits, (int64_t)VQ_HIST_ROWS, w.off, w.tot, w.sorted,
                       w.code_base, w.chunk_base);
    CEEG_CHECK_LAUNCH();
    const int64_t nchunk_max = B / VQ_SEG + n_e + 1;
    const unsigned sblocks = (unsigned)((nchunk_max + 3) / 4);
    if (dtype == CEEG_BF16)
      hipLaunchKernelGGL(vq_segsum_kernel<true>, dim3(sblocks), dim3(256), 0, st, z, W, w.sorted, w.code_base, w.chunk_base, n_e, D, w.partial);
    else
      hipLaunchKernelGGL(vq_segsum_kernel<false>, dim3(sblocks), dim3(256), 0, st, z, W, w.sorted, w.code_base, w.chunk_base, n_e, D, w.partial);
    CEEG_CHECK_LAUNCH();
    hipLaunchKernelGGL(vq_dw_kernel, dim3(n_e), dim3(256), 0, st, w.partial, w.chunk_base, s_dev, B, n_e, D, beta, dW);
    CEEG_CHECK_LAUNCH();
  }
  return CEEG_OK;
}
