// Graph-mode LayerNorm over a collated batch (the reference instantiates
// torch_geometric.nn.LayerNorm(mode='graph') after every GATConv: scripts/models.py:107): statistics over ALL nodes and
// channels of each graph, out = (x - mean_g) / sqrt(var_g + eps) * weight + bias [, ReLU: scripts/models.py:108].
//
// HBM-bound streaming work; two reads of x and one write per forward (the minimum of a two-stage normalisation), no atomics:
//   1. gln_rows_kernel   one wave per node row: fp64 row sums of (x, x^2) [forward] or (g w, g w xhat) [backward]; in the
//                        backward the same pass also leaves per-block column partials of (g, g xhat) for dweight / dbias
//   2. gln_graph_kernel  one workgroup per graph: fixed-order fp64 sum of its rows' pairs -> (mean, rstd) or (m1, m2)
//   3. gln_apply_kernel  one wave per row: normalise + affine (+ ReLU), resp. dx = rstd (g w - m1 - xhat m2)
//   4. gln_cols_kernel   (backward) column partials of the blocks summed in block order in fp64
// The graph of a row is found by a binary search in ptr[G+1] (the collated batch is sorted by graph).
#include "ceeg_internal.h"

namespace {

constexpr int GLN_WAVES = 4;
constexpr int GLN_ROWS_PER_BLOCK = 64;   // rows per workgroup of the row passes (16 per wave)
constexpr int GLN_MAXF = 2048;           // columns (the reference's layers are 2048 wide)
constexpr int GLN_CPL = GLN_MAXF / 256;  // float4 pieces per lane at the widest row

__device__ __forceinline__ int gln_graph_of(const int32_t* __restrict__ ptr, int G, int row) {
  int lo = 0, hi = G;  // last g with ptr[g] <= row
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ptr[mid] <= row) lo = mid; else hi = mid;
  }
  return lo;
}

// BWD = false: pair[n] = (sum_f x, sum_f x^2).
// BWD = true : gm = g * [out > 0 if relu]; pair[n] = (sum_f gm w, sum_f gm w xhat); column partials of (gm xhat, gm) per block.
template <bool BWD>
__global__ __launch_bounds__(GLN_WAVES * 64) void gln_rows_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ g, int64_t ldg, const float* __restrict__ y, int64_t ldy,
    const int32_t* __restrict__ ptr, int G, int64_t N, int F, const float* __restrict__ weight, const double* __restrict__ mean,
    const double* __restrict__ rstd, int relu, double2* __restrict__ pair, float* __restrict__ colpart /* [nblk][2][F] */) {
  __shared__ float s_col[BWD ? 2 * GLN_MAXF : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * GLN_ROWS_PER_BLOCK;
  float4 cw[GLN_CPL], cb[GLN_CPL];  // running column sums of this lane's columns: (gm xhat), (gm)
#pragma unroll
  for (int k = 0; k < GLN_CPL; ++k) cw[k] = cb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = wave; i < GLN_ROWS_PER_BLOCK; i += GLN_WAVES) {  // rows in a fixed order per wave
    const int64_t n = r0 + i;
    if (n >= N) break;
    double mu = 0.0, rs = 0.0;  // fp64: x - mean must not lose the digits a rounded mean would (|mean| >> std)
    if (BWD) {
      const int gi = gln_graph_of(ptr, G, (int)n);
      mu = mean[gi];
      rs = rstd[gi];
    }
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < GLN_CPL; ++k) {
      const int c = 4 * lane + 256 * k;
      if (c >= F) break;
      const float4 xv = *reinterpret_cast<const float4*>(x + n * ldx + c);
      if (!BWD) {
        a += (double)xv.x + (double)xv.y + (double)xv.z + (double)xv.w;
        b += (double)xv.x * xv.x + (double)xv.y * xv.y + (double)xv.z * xv.z + (double)xv.w * xv.w;
      } else {
        float4 gv = *reinterpret_cast<const float4*>(g + n * ldg + c);
        if (relu) {
          const float4 yv = *reinterpret_cast<const float4*>(y + n * ldy + c);
          gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f;
          gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
        }
        const float4 xh = make_float4((float)(((double)xv.x - mu) * rs), (float)(((double)xv.y - mu) * rs),
                                      (float)(((double)xv.z - mu) * rs), (float)(((double)xv.w - mu) * rs));
        float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (weight) w4 = *reinterpret_cast<const float4*>(weight + c);
        const float4 gw = make_float4(gv.x * w4.x, gv.y * w4.y, gv.z * w4.z, gv.w * w4.w);
        a += (double)gw.x + (double)gw.y + (double)gw.z + (double)gw.w;
        b += (double)gw.x * xh.x + (double)gw.y * xh.y + (double)gw.z * xh.z + (double)gw.w * xh.w;
        cw[k].x = __fmaf_rn(gv.x, xh.x, cw[k].x); cw[k].y = __fmaf_rn(gv.y, xh.y, cw[k].y);
        cw[k].z = __fmaf_rn(gv.z, xh.z, cw[k].z); cw[k].w = __fmaf_rn(gv.w, xh.w, cw[k].w);
        cb[k].x += gv.x; cb[k].y += gv.y; cb[k].z += gv.z; cb[k].w += gv.w;
      }
    }
    a = ceeg_wave_sum_d(a);
    b = ceeg_wave_sum_d(b);
    if (lane == 0) pair[n] = make_double2(a, b);
  }
  if (BWD) {  // the four waves' column sums, added in wave order (wave 0 stores, waves 1..3 add in turn)
    for (int w = 0; w < GLN_WAVES; ++w) {
      if (wave == w) {
#pragma unroll
        for (int k = 0; k < GLN_CPL; ++k) {
          const int c = 4 * lane + 256 * k;
          if (c >= F) break;
          float4 tw = cw[k], tb = cb[k];
          if (w > 0) {
            const float4 pw = *reinterpret_cast<const float4*>(&s_col[c]);
            const float4 pb = *reinterpret_cast<const float4*>(&s_col[GLN_MAXF + c]);
            tw.x += pw.x; tw.y += pw.y; tw.z += pw.z; tw.w += pw.w;
            tb.x += pb.x; tb.y += pb.y; tb.z += pb.z; tb.w += pb.w;
          }
          *reinterpret_cast<float4*>(&s_col[c]) = tw;
          *reinterpret_cast<float4*>(&s_col[GLN_MAXF + c]) = tb;
        }
      }
      __syncthreads();
    }
    for (int c = threadIdx.x; c < 2 * F; c += GLN_WAVES * 64) {
      const int which = c / F, col = c % F;
      colpart[((size_t)blockIdx.x * 2 + which) * F + col] = s_col[which * GLN_MAXF + col];
    }
  }
}

// one workgroup per graph: (sum a, sum b) over its rows in a fixed order -> stats
// mode 0: out0 = mean, out1 = rstd = 1/sqrt(var + eps)   (count = rows * F, clamp(min=1) rows as the reference)
// mode 1: out0 = sum a / count, out1 = sum b / count
__global__ __launch_bounds__(256) void gln_graph_kernel(const double2* __restrict__ pair, const int32_t* __restrict__ ptr, int F, float eps,
                                                       int mode, double* __restrict__ out0, double* __restrict__ out1) {
  __shared__ double ra[256], rb[256];
  const int gi = blockIdx.x;
  const int lo = ptr[gi], hi = ptr[gi + 1];
  double a = 0.0, b = 0.0;
  for (int n = lo + (int)threadIdx.x; n < hi; n += 256) {
    const double2 p = pair[n];
    a += p.x;
    b += p.y;
  }
  ra[threadIdx.x] = a;
  rb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ra[threadIdx.x] += ra[threadIdx.x + o]; rb[threadIdx.x] += rb[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int rows = hi - lo;
    const double cnt = (double)(rows < 1 ? 1 : rows) * (double)F;
    if (mode == 0) {
      const double m = ra[0] / cnt;
      double var = rb[0] / cnt - m * m;
      if (var < 0.0) var = 0.0;
      out0[gi] = m;
      out1[gi] = 1.0 / sqrt(var + (double)eps);
    } else {
      out0[gi] = ra[0] / cnt;
      out1[gi] = rb[0] / cnt;
    }
  }
}

// BWD = false: out = xhat * weight + bias (+ReLU);  BWD = true: dx = rstd * (gm w - m1 - xhat m2)
template <bool BWD>
__global__ __launch_bounds__(GLN_WAVES * 64) void gln_apply_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ g, int64_t ldg, const float* __restrict__ y, int64_t ldy,
    const int32_t* __restrict__ ptr, int G, int64_t N, int F, const float* __restrict__ weight, const float* __restrict__ bias,
    const double* __restrict__ mean, const double* __restrict__ rstd, const double* __restrict__ m1, const double* __restrict__ m2,
    int relu, float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * GLN_ROWS_PER_BLOCK;
  for (int i = wave; i < GLN_ROWS_PER_BLOCK; i += GLN_WAVES) {
    const int64_t n = r0 + i;
    if (n >= N) break;
    const int gi = gln_graph_of(ptr, G, (int)n);
    const double mu = mean[gi], rs = rstd[gi];
    float a1 = 0.f, a2 = 0.f;
    const float rsf = (float)rs;
    if (BWD) { a1 = (float)m1[gi]; a2 = (float)m2[gi]; }
#pragma unroll
    for (int k = 0; k < GLN_CPL; ++k) {
      const int c = 4 * lane + 256 * k;
      if (c >= F) break;
      const float4 xv = *reinterpret_cast<const float4*>(x + n * ldx + c);
      const float4 xh = make_float4((float)(((double)xv.x - mu) * rs), (float)(((double)xv.y - mu) * rs),
                                    (float)(((double)xv.z - mu) * rs), (float)(((double)xv.w - mu) * rs));
      float4 w4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (weight) w4 = *reinterpret_cast<const float4*>(weight + c);
      float4 o;
      if (!BWD) {
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) b4 = *reinterpret_cast<const float4*>(bias + c);
        o = make_float4(__fmaf_rn(xh.x, w4.x, b4.x), __fmaf_rn(xh.y, w4.y, b4.y), __fmaf_rn(xh.z, w4.z, b4.z), __fmaf_rn(xh.w, w4.w, b4.w));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      } else {
        float4 gv = *reinterpret_cast<const float4*>(g + n * ldg + c);
        if (relu) {
          const float4 yv = *reinterpret_cast<const float4*>(y + n * ldy + c);
          gv.x = yv.x > 0.f ? gv.x : 0.f; gv.y = yv.y > 0.f ? gv.y : 0.f;
          gv.z = yv.z > 0.f ? gv.z : 0.f; gv.w = yv.w > 0.f ? gv.w : 0.f;
        }
        o = make_float4(rsf * (gv.x * w4.x - a1 - xh.x * a2), rsf * (gv.y * w4.y - a1 - xh.y * a2),
                        rsf * (gv.z * w4.z - a1 - xh.z * a2), rsf * (gv.w * w4.w - a1 - xh.w * a2));
      }
      *reinterpret_cast<float4*>(out + n * ldo + c) = o;
    }
  }
}

// dweight[f] = sum over blocks of colpart[blk][0][f], dbias likewise from [1]: block order, fp64
__global__ __launch_bounds__(256) void gln_cols_kernel(const float* __restrict__ colpart, int64_t nblk, int F, float* __restrict__ dweight,
                                                      float* __restrict__ dbias) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * F) return;
  const int which = c / F, col = c % F;
  double t = 0.0;
  for (int64_t b = 0; b < nblk; ++b) t += (double)colpart[((size_t)b * 2 + which) * F + col];
  float* dst = which == 0 ? dweight : dbias;
  if (dst) dst[col] = (float)t;
}

struct GlnWs {
  double2* pair;
  double *s0, *s1;
  float* colpart;
};
size_t gln_layout(int64_t N, int F, int64_t G, void* ws, size_t ws_bytes, GlnWs* out) {
  const int64_t nblk = (N + GLN_ROWS_PER_BLOCK - 1) / GLN_ROWS_PER_BLOCK;
  const size_t need = ceeg_ws_part((size_t)N, 16) + 2 * ceeg_ws_part((size_t)G, 8) + ceeg_ws_part((size_t)nblk * 2 * F, 4);
  if (out) {
    CeegWs a(ws, ws_bytes);
    out->pair = a.take<double2>((size_t)N);
    out->s0 = a.take<double>((size_t)G);
    out->s1 = a.take<double>((size_t)G);
    out->colpart = a.take<float>((size_t)nblk * 2 * F);
    if (!a.ok) return 0;
  }
  return need;
}

int gln_check(const void* x, int64_t ldx, const int32_t* ptr, int64_t N, int F, int64_t G) {
  if (!x || !ptr || N <= 0 || G <= 0 || F <= 0) return CEEG_EINVAL;
  if ((F % 4) != 0 || F > GLN_MAXF || (ldx % 4) != 0 || ldx < F || N >= ((int64_t)1 << 31) || G >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(x, 16)) return CEEG_EINVAL;
  return CEEG_OK;
}

}  // namespace

extern "C" size_t ceeg_graph_layernorm_ws_bytes(int64_t N, int F, int64_t G) {
  if (N <= 0 || F <= 0 || G <= 0) return 0;
  return gln_layout(N, F, G, nullptr, 0, nullptr);
}

extern "C" int ceeg_graph_layernorm_fwd(const float* x, int64_t ldx, const int32_t* ptr, int64_t N, int F, int64_t G, const float* weight,
                                        const float* bias, float eps, int relu, float* out, int64_t ldo, double* mean, double* rstd,
                                        void* ws, size_t ws_bytes, void* stream) {
  int rc = gln_check(x, ldx, ptr, N, F, G);
  if (rc != CEEG_OK) return rc;
  if (!out || !mean || !rstd) return CEEG_EINVAL;
  if ((ldo % 4) != 0 || ldo < F) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(out, 16) || (weight && !ceeg_aligned(weight, 16)) || (bias && !ceeg_aligned(bias, 16))) return CEEG_EINVAL;
  GlnWs w;
  if (ws_bytes < gln_layout(N, F, G, nullptr, 0, nullptr) || !ws || !ceeg_aligned(ws, 256) || gln_layout(N, F, G, ws, ws_bytes, &w) == 0)
    return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nblk = (unsigned)((N + GLN_ROWS_PER_BLOCK - 1) / GLN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(gln_rows_kernel<false>, dim3(nblk), dim3(GLN_WAVES * 64), 0, st, x, ldx, nullptr, 0, nullptr, 0, ptr, (int)G, N, F,
                     nullptr, nullptr, nullptr, 0, w.pair, nullptr);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(gln_graph_kernel, dim3((unsigned)G), dim3(256), 0, st, w.pair, ptr, F, eps, 0, mean, rstd);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(gln_apply_kernel<false>, dim3(nblk), dim3(GLN_WAVES * 64), 0, st, x, ldx, nullptr, 0, nullptr, 0, ptr, (int)G, N, F, weight,
                     bias, mean, rstd, nullptr, nullptr, relu, out, ldo);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_graph_layernorm_bwd(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* y, int64_t ldy,
                                        const int32_t* ptr, int64_t N, int F, int64_t G, const float* weight, const double* mean,
                                        const double* rstd, int relu, float* dx, int64_t lddx, float* dweight, float* dbias,
                                        void* ws, size_t ws_bytes, void* stream) {
  int rc = gln_check(x, ldx, ptr, N, F, G);
  if (rc != CEEG_OK) return rc;
  if (!g || !mean || !rstd || !dx || (relu && !y)) return CEEG_EINVAL;
  if ((ldg % 4) != 0 || ldg < F || (lddx % 4) != 0 || lddx < F || (relu && ((ldy % 4) != 0 || ldy < F))) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(g, 16) || !ceeg_aligned(dx, 16) || (y && !ceeg_aligned(y, 16)) || (weight && !ceeg_aligned(weight, 16))) return CEEG_EINVAL;
  GlnWs w;
  if (ws_bytes < gln_layout(N, F, G, nullptr, 0, nullptr) || !ws || !ceeg_aligned(ws, 256) || gln_layout(N, F, G, ws, ws_bytes, &w) == 0)
    return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const unsigned nblk = (unsigned)((N + GLN_ROWS_PER_BLOCK - 1) / GLN_ROWS_PER_BLOCK);
  hipLaunchKernelGGL(gln_rows_kernel<true>, dim3(nblk), dim3(GLN_WAVES * 64), 0, st, x, ldx, g, ldg, y, ldy, ptr, (int)G, N, F, weight, mean, rstd,
                     relu, w.pair, w.colpart);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(gln_graph_kernel, dim3((unsigned)G), dim3(256), 0, st, w.pair, ptr, F, 0.f, 1, w.s0, w.s1);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(gln_apply_kernel<true>, dim3(nblk), dim3(GLN_WAVES * 64), 0, st, x, ldx, g, ldg, y, ldy, ptr, (int)G, N, F, weight, nullptr,
                     mean, rstd, w.s0, w.s1, relu, dx, lddx);
  CEEG_CHECK_LAUNCH();
  if (dweight || dbias) {
    hipLaunchKernelGGL(gln_cols_kernel, dim3((unsigned)((2 * F + 255) / 256)), dim3(256), 0, st, w.colpart, (int64_t)nblk, F, dweight, dbias);
    CEEG_CHECK_LAUNCH();
  }
  return CEEG_OK;
}
