// Small streaming kernels around the dense transforms of a GCN layer (HBM-bound, one pass each):
//   ceeg_bias_relu      : y = relu(y + b) in place (epilogue of (A_hat x) W^T)
//   ceeg_relu_bwd_colsum: gm = g * (y > 0) (gm may alias g) AND colsum[f] = sum_n g[n,f] (the bias gradient) in one pass
//   ceeg_colsum         : colsum[f] = sum_n g[n,f]
// Column sums are deterministic: per-block partial rows in a fixed order, then one fp64 pass over the blocks.
// They stand behind torch.relu / its backward / bias.grad of the reference-style host code (no reference
// counterpart of their own: PyG GCNConv adds its bias and autograd reduces it).
#include "ceeg_internal.h"

namespace {

constexpr int EW_THREADS = 256;
constexpr int EW_MAX_BLOCKS = 1024;

__global__ __launch_bounds__(EW_THREADS) void bias_relu_kernel(float* __restrict__ y, int64_t ld, int64_t N, int F,
                                                               const float* __restrict__ bias, int relu) {
  const int fq = F / 4;
  const int64_t total = N * fq;
  for (int64_t g = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; g < total; g += (int64_t)gridDim.x * EW_THREADS) {
    const int64_t n = g / fq;
    const int f = (int)(g - n * fq) * 4;
    float4 v = *reinterpret_cast<float4*>(y + n * ld + f);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + f);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    *reinterpret_cast<float4*>(y + n * ld + f) = v;
  }
}

// block b owns rows [b*rows_per, ...); thread t owns column quad (t % fq) and walks rows (t / fq), +rpb, ...
template <bool MASK>
__global__ __launch_bounds__(EW_THREADS) void colsum_kernel(const float* g, int64_t ldg, const float* __restrict__ y, int64_t ldy,
                                                            float* gm, int64_t ldm, int64_t N, int F, int64_t rows_per,
                                                            float* __restrict__ part /* [grid][F] */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* red = reinterpret_cast<float4*>(smem);  // [EW_THREADS]
  const int fq = F / 4;              // <= EW_THREADS
  const int rpb = EW_THREADS / fq;   // rows walked concurrently by the block
  const int cq = threadIdx.x % fq, rr = threadIdx.x / fq;
  const int64_t lo = (int64_t)blockIdx.x * rows_per;
  const int64_t hi = lo + rows_per < N ? lo + rows_per : N;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rr < rpb) {
    for (int64_t n = lo + rr; n < hi; n += rpb) {
      float4 v = *reinterpret_cast<const float4*>(g + n * ldg + 4 * cq);
      if (MASK) {
        const float4 o = *reinterpret_cast<const float4*>(y + n * ldy + 4 * cq);
        v.x = o.x > 0.f ? v.x : 0.f; v.y = o.y > 0.f ? v.y : 0.f; v.z = o.z > 0.f ? v.z : 0.f; v.w = o.w > 0.f ? v.w : 0.f;
        *reinterpret_cast<float4*>(gm + n * ldm + 4 * cq) = v;
      }
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rr == 0) {  // fixed order over the block's row groups
    for (int k = 1; k < rpb; ++k) {
      const float4 t = red[k * fq + cq];
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * F + 4 * cq) = acc;
  }
}

// 16 columns per block; 16 groups of threads walk the block partials in a fixed interleave, fp64, then a fixed-order
// LDS sum over the 16 groups.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ part, int nblk, int F, float* __restrict__ out) {
  __shared__ double red[256];
  const int c = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int f = blockIdx.x * 16 + c;
  double s = 0.0;
  if (f < F) {
    int b = grp;
    for (; b + 16 * 7 < nblk; b += 16 * 8) {  // eight loads in flight, added in the same ascending order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 16 * u) * F + f];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; b < nblk; b += 16) s += (double)part[(size_t)b * F + f];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (grp == 0 && f < F) {
    for (int k = 1; k < 16; ++k) s += red[k * 16 + c];
    out[f] = (float)s;
  }
}

int ew_blocks(int64_t work) {
  int64_t b = (work + EW_THREADS - 1) / EW_THREADS;
  if (b < 1) b = 1;
  return (int)(b < 4 * EW_MAX_BLOCKS ? b : 4 * EW_MAX_BLOCKS);
}

int colsum_launch(const float* g, int64_t ldg, const float* y, int64_t ldy, float* gm, int64_t ldm, int64_t N, int F, float* colsum,
                  void* ws, size_t ws_bytes, hipStream_t st) {
  if (!g || !colsum || N <= 0 || F < 4 || (F % 4) != 0 || F > 4 * EW_THREADS || ldg < F || (ldg % 4) != 0) return CEEG_EINVAL;
  if (y && (!gm || ldy < F || (ldy % 4) != 0 || ldm < F || (ldm % 4) != 0)) return CEEG_EINVAL;
  if (!ceeg_aligned(g, 16) || (y && (!ceeg_aligned(y, 16) || !ceeg_aligned(gm, 16)))) return CEEG_EINVAL;
  if (ws_bytes < ceeg_colsum_ws_bytes(N, F) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  int64_t nblk = (N + 255) / 256;
  if (nblk > EW_MAX_BLOCKS) nblk = EW_MAX_BLOCKS;
  const int64_t rows_per = (N + nblk - 1) / nblk;
  nblk = (N + rows_per - 1) / rows_per;
  float* part = static_cast<float*>(ws);
  if (y)
    hipLaunchKernelGGL(colsum_kernel<true>, dim3((unsigned)nblk), dim3(EW_THREADS), EW_THREADS * 16, st, g, ldg, y, ldy, gm, ldm, N, F, rows_per, part);
  else
    hipLaunchKernelGGL(colsum_kernel<false>, dim3((unsigned)nblk), dim3(EW_THREADS), EW_THREADS * 16, st, g, ldg, y, ldy, gm, ldm, N, F, rows_per, part);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((F + 15) / 16), dim3(256), 0, st, part, (int)nblk, F, colsum);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

}  // namespace

extern "C" int ceeg_bias_relu(float* y, int64_t ld, int64_t N, int F, const float* bias, int relu, void* stream) {
  if (!y || N <= 0 || F < 4 || (F % 4) != 0 || ld < F || (ld % 4) != 0) return CEEG_EINVAL;
  if (!ceeg_aligned(y, 16) || (bias && !ceeg_aligned(bias, 16))) return CEEG_EINVAL;
  hipLaunchKernelGGL(bias_relu_kernel, dim3(ew_blocks(N * (F / 4))), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), y, ld, N, F,
                     bias, relu);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

namespace {
// out[n] = k-ascending fmaf chain from 0 of x[n,k]^2: a thread per row, the row's 16-byte pieces 8 at a time
__global__ __launch_bounds__(EW_THREADS) void row_sqnorm_kernel(const float* __restrict__ x, int64_t ld, int64_t N, int F, float* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (n >= N) return;
  const float* r = x + n * ld;
  float acc = 0.f;
  int k = 0;
  for (; k + 32 <= F; k += 32) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(r + k + 4 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc = __fmaf_rn(v[u].x, v[u].x, acc); acc = __fmaf_rn(v[u].y, v[u].y, acc);
      acc = __fmaf_rn(v[u].z, v[u].z, acc); acc = __fmaf_rn(v[u].w, v[u].w, acc);
    }
  }
  for (; k < F; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(r + k);
    acc = __fmaf_rn(v.x, v.x, acc); acc = __fmaf_rn(v.y, v.y, acc); acc = __fmaf_rn(v.z, v.z, acc); acc = __fmaf_rn(v.w, v.w, acc);
  }
  out[n] = acc;
}
}  // namespace

extern "C" int ceeg_row_sqnorm(const float* x, int64_t ld, int64_t N, int F, float* out, void* stream) {
  if (!x || !out || N <= 0 || F < 4 || (F % 4) != 0 || ld < F || (ld % 4) != 0 || !ceeg_aligned(x, 16)) return CEEG_EINVAL;
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((N + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), x,
                     ld, N, F, out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

namespace {
// out[n, :F] = x[n, :], out[n, F:F+C] = cond[batch[n], :]: one 16-byte piece per thread (cond is a few KB: cache-resident)
__global__ __launch_bounds__(EW_THREADS) void concat_condition_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int F,
                                                                      const float* __restrict__ cond, int C, int64_t G,
                                                                      const int64_t* __restrict__ batch, float* __restrict__ out) {
  const int pw = (F + C) / 4;
  const int64_t total = N * pw;
  for (int64_t i = (int64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * EW_THREADS) {
    const int64_t n = i / pw;
    const int k = (int)(i - n * pw) * 4;
    float4 v;
    if (k < F) {
      v = *reinterpret_cast<const float4*>(x + n * ldx + k);
    } else {
      int64_t g = batch[n];
      g = g < 0 ? 0 : (g >= G ? G - 1 : g);   // checked on the host side of the collation; clamped here so that nothing faults
      v = *reinterpret_cast<const float4*>(cond + g * C + (k - F));
    }
    *reinterpret_cast<float4*>(out + n * (int64_t)(F + C) + k) = v;
  }
}
}  // namespace

extern "C" int ceeg_concat_condition(const float* x, int64_t ldx, int64_t N, int F, const float* cond, int C, int64_t G, const int64_t* batch,
                                     float* out, void* stream) {
  if (!x || !cond || !batch || !out || N <= 0 || G <= 0 || F < 4 || C < 4 || (F % 4) != 0 || (C % 4) != 0 || ldx < F || (ldx % 4) != 0)
    return CEEG_EINVAL;
  if (!ceeg_aligned(x, 16) || !ceeg_aligned(cond, 16) || !ceeg_aligned(out, 16)) return CEEG_EINVAL;
  hipLaunchKernelGGL(concat_condition_kernel, dim3(ew_blocks(N * ((F + C) / 4))), dim3(EW_THREADS), 0, static_cast<hipStream_t>(stream), x, ldx, N, F,
                     cond, C, G, batch, out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" size_t ceeg_colsum_ws_bytes(int64_t N, int F) {
  if (N <= 0 || F <= 0) return 0;
  return ceeg_ws_part((size_t)EW_MAX_BLOCKS * F, 4);
}

extern "C" int ceeg_relu_bwd_colsum(const float* g, int64_t ldg, const float* y, int64_t ldy, float* gm, int64_t ldm, int64_t N, int F,
                                    float* colsum, void* ws, size_t ws_bytes, void* stream) {
  if (!y) return CEEG_EINVAL;
  return colsum_launch(g, ldg, y, ldy, gm, ldm, N, F, colsum, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

extern "C" int ceeg_colsum(const float* g, int64_t ldg, int64_t N, int F, float* colsum, void* ws, size_t ws_bytes, void* stream) {
  return colsum_launch(g, ldg, nullptr, 0, nullptr, 0, N, F, colsum, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

// recon_loss's last step (upstream PyG GAE.recon_loss: pos_loss.mean() + neg_loss.mean()) as ONE tiny launch instead of a chain
// of torch scalar kernels: out = { pos_sum / pos_count + neg_sum / max(neg_count, 1), 1 / pos_count, 1 / max(neg_count, 1),
// pos_sum, neg_sum, neg_count }.  The two reciprocals are what the decoders' backward scales its coefficient sums with.
namespace {
__global__ void recon_finish_kernel(const float* __restrict__ pos_sum, const float* __restrict__ neg_sum, const float* __restrict__ neg_cnt,
                                    float pos_count, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float ps = pos_sum[0], ns = neg_sum[0], nc = neg_cnt[0];
  const float ip = 1.0f / pos_count, in = 1.0f / fmaxf(nc, 1.0f);
  out[0] = ps * ip + ns * in;
  out[1] = ip;
  out[2] = in;
  out[3] = ps;
  out[4] = ns;
  out[5] = nc;
}
}  // namespace

extern "C" int ceeg_recon_finish(const float* pos_sum, const float* neg_sum, const float* neg_count, float pos_count, float* out, void* stream) {
  if (!pos_sum || !neg_sum || !neg_count || !out || !(pos_count > 0.f)) return CEEG_EINVAL;
  hipLaunchKernelGGL(recon_finish_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), pos_sum, neg_sum, neg_count, pos_count, out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

// loss = rec + kl * kl_scale + vq  (the loss sum of scripts/run.py:95 with the VGAE example's (1/N) kl_loss) in ONE launch; the
// backward's three factors are (g, g * kl_scale, g): grads = { g, g * kl_scale } written by the same kernel when g is given.
namespace {
__global__ void loss_sum_kernel(const float* __restrict__ rec, const float* __restrict__ kl, const float* __restrict__ vq, float kl_scale,
                                const float* __restrict__ g, float* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (g) {
    out[0] = g[0];
    out[1] = g[0] * kl_scale;
  } else {
    out[0] = rec[0] + kl[0] * kl_scale + vq[0];
  }
}
}  // namespace

extern "C" int ceeg_loss_sum(const float* rec, const float* kl, const float* vq, float kl_scale, const float* g, float* out, void* stream) {
  if (!out || (!g && (!rec || !kl || !vq))) return CEEG_EINVAL;
  hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), rec, kl, vq, kl_scale, g, out);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}
