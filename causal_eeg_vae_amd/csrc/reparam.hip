// Fused VGAE reparametrisation + KL (forward and backward) for gfx950.  Pure HBM streaming: every
// input element is read once and every output written once; the KL sum is a fixed-order tree
// (thread -> wave butterfly -> block -> one fp64 pass over the block partials), no atomics.
//
// Stands behind upstream PyG VGAE.encode (MAX_LOGSTD clamp, /root/reference/scripts/run.py:13),
// VGAE.reparametrize, VGAE.kl_loss, and the in-repo log-variance variant scripts/models.py:38-43.
#include "ceeg_internal.h"

namespace {

constexpr int RP_THREADS = 256;
constexpr int RP_MAX_BLOCKS = 2048;

struct RpElem {
  float z, klterm, dmu, dls;
};

// s2 = log variance (= 2*logstd in the PyG form, = the input itself in the log_var form)
__device__ __forceinline__ void rp_fwd_elem(float mu, float raw, float eps, float max_ls, int log_var, int training,
                                            float& z, float& klterm) {
  const float ls = fminf(raw, max_ls);
  const float s2 = log_var ? ls : 2.0f * ls;
  const float sd = expf(0.5f * s2);
  z = training ? __fmaf_rn(eps, sd, mu) : mu;
  klterm = 1.0f + s2 - mu * mu - expf(s2);
}

__global__ __launch_bounds__(RP_THREADS) void reparam_kl_fwd_kernel(
    const float* __restrict__ mu, int64_t ld_mu, const float* __restrict__ logstd, int64_t ld_ls,
    const float* __restrict__ eps, uint64_t seed, int64_t N, int D, float max_ls, int log_var, int training, int vec,
    float* __restrict__ z, float* __restrict__ part) {
  __shared__ float red[RP_THREADS / 64];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * RP_THREADS;
  if (vec) {
    const int dq = D / 4;
    const int64_t total = N * dq;
    for (int64_t g = (int64_t)blockIdx.x * RP_THREADS + threadIdx.x; g < total; g += stride) {
      const int64_t n = g / dq;
      const int d = (int)(g - n * dq) * 4;
      const float4 m = *reinterpret_cast<const float4*>(mu + n * ld_mu + d);
      const float4 l = *reinterpret_cast<const float4*>(logstd + n * ld_ls + d);
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (training) {
        if (eps) {
          e = *reinterpret_cast<const float4*>(eps + n * D + d);
        } else {
          const uint4 x = ceeg_philox(seed, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), CEEG_TAG_EPS, 0u);
          const float2 a = ceeg_box_muller(x.x, x.y), b = ceeg_box_muller(x.z, x.w);
          e = make_float4(a.x, a.y, b.x, b.y);
        }
      }
      float4 o;
      float t0, t1, t2, t3;
      rp_fwd_elem(m.x, l.x, e.x, max_ls, log_var, training, o.x, t0);
      rp_fwd_elem(m.y, l.y, e.y, max_ls, log_var, training, o.y, t1);
      rp_fwd_elem(m.z, l.z, e.z, max_ls, log_var, training, o.z, t2);
      rp_fwd_elem(m.w, l.w, e.w, max_ls, log_var, training, o.w, t3);
      *reinterpret_cast<float4*>(z + n * D + d) = o;
      acc += (t0 + t1) + (t2 + t3);
    }
  } else {
    const int64_t total = N * D;
    for (int64_t i = (int64_t)blockIdx.x * RP_THREADS + threadIdx.x; i < total; i += stride) {
      const int64_t n = i / D;
      const int d = (int)(i - n * D);
      float e = 0.f;
      if (training) {
        if (eps) {
          e = eps[i];
        } else {
          const int64_t g = i >> 2;
          const uint4 x = ceeg_philox(seed, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), CEEG_TAG_EPS, 0u);
          const int k = (int)(i & 3);
          const float2 a = k < 2 ? ceeg_box_muller(x.x, x.y) : ceeg_box_muller(x.z, x.w);
          e = (k & 1) ? a.y : a.x;
        }
      }
      float o, t;
      rp_fwd_elem(mu[n * ld_mu + d], logstd[n * ld_ls + d], e, max_ls, log_var, training, o, t);
      z[i] = o;
      acc += t;
    }
  }
  acc = ceeg_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void reparam_kl_finalize_kernel(const float* __restrict__ part, int nblk, int64_t N, float* __restrict__ kl) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *kl = (float)(-0.5 * red[0] / (double)N);
}

__device__ __forceinline__ void rp_bwd_elem(float mu, float raw, float eps, float gz, float gkl_n, float max_ls, int log_var,
                                            int training, float& dmu, float& dls) {
  const float ls = fminf(raw, max_ls);
  const float s2 = log_var ? ls : 2.0f * ls;
  const float chain = log_var ? 1.0f : 2.0f;  // d s2 / d input
  const float sd = expf(0.5f * s2);
  dmu = __fmaf_rn(gkl_n, mu, gz);                           // gz + gkl*mu/N
  float g = -0.5f * gkl_n * (1.0f - expf(s2)) * chain;      // KL part
  if (training) g = __fmaf_rn(gz * eps, sd * (0.5f * chain), g);  // z part: eps*sd*d(0.5*s2)/d input
  dls = raw <= max_ls ? g : 0.f;                            // clamp(max=) passes the gradient where raw <= max
}

__global__ __launch_bounds__(RP_THREADS) void reparam_kl_bwd_kernel(
    const float* __restrict__ mu, int64_t ld_mu, const float* __restrict__ logstd, int64_t ld_ls,
    const float* __restrict__ eps, uint64_t seed, const float* __restrict__ gz, const float* __restrict__ gkl_dev,
    int64_t N, int D, float max_ls, int log_var, int training, int vec,
    float* __restrict__ dmu, int64_t ld_dmu, float* __restrict__ dls, int64_t ld_dls) {
  const float gkl_n = gkl_dev ? gkl_dev[0] / (float)N : 0.f;
  const int64_t stride = (int64_t)gridDim.x * RP_THREADS;
  if (vec) {
    const int dq = D / 4;
    const int64_t total = N * dq;
    for (int64_t g = (int64_t)blockIdx.x * RP_THREADS + threadIdx.x; g < total; g += stride) {
      const int64_t n = g / dq;
      const int d = (int)(g - n * dq) * 4;
      const float4 m = *reinterpret_cast<const float4*>(mu + n * ld_mu + d);
      const float4 l = *reinterpret_cast<const float4*>(logstd + n * ld_ls + d);
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (training) {
        if (eps) {
          e = *reinterpret_cast<const float4*>(eps + n * D + d);
        } else {
          const uint4 x = ceeg_philox(seed, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), CEEG_TAG_EPS, 0u);
          const float2 a = ceeg_box_muller(x.x, x.y), b = ceeg_box_muller(x.z, x.w);
          e = make_float4(a.x, a.y, b.x, b.y);
        }
      }
      float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gz) go = *reinterpret_cast<const float4*>(gz + n * D + d);
      float4 om, ol;
      rp_bwd_elem(m.x, l.x, e.x, go.x, gkl_n, max_ls, log_var, training, om.x, ol.x);
      rp_bwd_elem(m.y, l.y, e.y, go.y, gkl_n, max_ls, log_var, training, om.y, ol.y);
      rp_bwd_elem(m.z, l.z, e.z, go.z, gkl_n, max_ls, log_var, training, om.z, ol.z);
      rp_bwd_elem(m.w, l.w, e.w, go.w, gkl_n, max_ls, log_var, training, om.w, ol.w);
      *reinterpret_cast<float4*>(dmu + n * ld_dmu + d) = om;
      *reinterpret_cast<float4*>(dls + n * ld_dls + d) = ol;
    }
  } else {
    const int64_t total = N * D;
    for (int64_t i = (int64_t)blockIdx.x * RP_THREADS + threadIdx.x; i < total; i += stride) {
      const int64_t n = i / D;
      const int d = (int)(i - n * D);
      float e = 0.f;
      if (training) {
        if (eps) {
          e = eps[i];
        } else {
          const int64_t g = i >> 2;
          const uint4 x = ceeg_philox(seed, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), CEEG_TAG_EPS, 0u);
          const int k = (int)(i & 3);
          const float2 a = k < 2 ? ceeg_box_muller(x.x, x.y) : ceeg_box_muller(x.z, x.w);
          e = (k & 1) ? a.y : a.x;
        }
      }
      float om, ol;
      rp_bwd_elem(mu[n * ld_mu + d], logstd[n * ld_ls + d], e, gz ? gz[i] : 0.f, gkl_n, max_ls, log_var, training, om, ol);
      dmu[n * ld_dmu + d] = om;
      dls[n * ld_dls + d] = ol;
    }
  }
}

// ------------------------------------------------------------------ fused: VQ dz + reparam backward + column sums
// The backward of  z = reparametrize(mu, logstd);  (loss_vq, z_q) = VectorQuantizer(z)  in one pass over the rows:
//   gz  = g_zq + g_z + s * (2 / (N D)) * (z - W[idx])      (ceeg_vq_backward's dz; z is RECOMPUTED from mu, logstd, eps)
//   dmu, dlogstd = reparam backward of gz and gkl          (rp_bwd_elem, same arithmetic as reparam_kl_bwd_kernel)
//   part[block][0:D | D:2D] = column sums of dmu | dlogstd over the block's rows  (the head GCNConv's bias gradient)
// Block b owns rows [b*rows_per, ...); thread t owns column quad t % (D/4) and walks rows t / (D/4), + RP_THREADS/(D/4), ...
__global__ __launch_bounds__(RP_THREADS) void reparam_vq_bwd_kernel(
    const float* __restrict__ mu, int64_t ld_mu, const float* __restrict__ logstd, int64_t ld_ls, const float* __restrict__ eps,
    uint64_t seed, const float* __restrict__ g_zq, const float* __restrict__ g_z, const float* __restrict__ W,
    const int64_t* __restrict__ idx, int n_e, const float* __restrict__ s_dev, const float* __restrict__ gkl_dev, int64_t N, int D,
    float max_ls, int log_var, int training, float* __restrict__ dmu, int64_t ld_dmu, float* __restrict__ dls, int64_t ld_dls,
    int64_t rows_per, float* __restrict__ part /* [grid][2D] or NULL */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* red = reinterpret_cast<float4*>(smem);  // [2][RP_THREADS]
  const int dq = D / 4;
  const int rpb = RP_THREADS / dq;
  const int cq = threadIdx.x % dq, rr = threadIdx.x / dq;
  const int d = 4 * cq;
  const float gkl_n = gkl_dev ? gkl_dev[0] / (float)N : 0.f;
  const float coef = s_dev ? s_dev[0] * (2.0f / ((float)N * (float)D)) : 0.f;
  const int64_t lo = (int64_t)blockIdx.x * rows_per;
  const int64_t hi = lo + rows_per < N ? lo + rows_per : N;
  float4 am = make_float4(0.f, 0.f, 0.f, 0.f), al = am;
  if (rr < rpb) {
    for (int64_t n = lo + rr; n < hi; n += rpb) {
      const float4 m = *reinterpret_cast<const float4*>(mu + n * ld_mu + d);
      const float4 l = *reinterpret_cast<const float4*>(logstd + n * ld_ls + d);
      int64_t k = idx[n];
      k = k < 0 ? 0 : (k >= n_e ? n_e - 1 : k);
      const float4 w = *reinterpret_cast<const float4*>(W + k * D + d);
      float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g_zq) go = *reinterpret_cast<const float4*>(g_zq + n * D + d);
      float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
      if (training) {
        if (eps) {
          e = *reinterpret_cast<const float4*>(eps + n * D + d);
        } else {
          const int64_t g = n * dq + cq;
          const uint4 x = ceeg_philox(seed, (uint32_t)g, (uint32_t)((uint64_t)g >> 32), CEEG_TAG_EPS, 0u);
          const float2 a = ceeg_box_muller(x.x, x.y), b = ceeg_box_muller(x.z, x.w);
          e = make_float4(a.x, a.y, b.x, b.y);
        }
      }
      float4 z;
      float t;
      rp_fwd_elem(m.x, l.x, e.x, max_ls, log_var, training, z.x, t);
      rp_fwd_elem(m.y, l.y, e.y, max_ls, log_var, training, z.y, t);
      rp_fwd_elem(m.z, l.z, e.z, max_ls, log_var, training, z.z, t);
      rp_fwd_elem(m.w, l.w, e.w, max_ls, log_var, training, z.w, t);
      float4 gz;  // as vq_dz4_kernel: coef * (z - w), then + g
      gz.x = coef * (z.x - w.x) + go.x; gz.y = coef * (z.y - w.y) + go.y;
      gz.z = coef * (z.z - w.z) + go.z; gz.w = coef * (z.w - w.w) + go.w;
      if (g_z) {
        const float4 h = *reinterpret_cast<const float4*>(g_z + n * D + d);
        gz.x += h.x; gz.y += h.y; gz.z += h.z; gz.w += h.w;
      }
      float4 om, ol;
      rp_bwd_elem(m.x, l.x, e.x, gz.x, gkl_n, max_ls, log_var, training, om.x, ol.x);
      rp_bwd_elem(m.y, l.y, e.y, gz.y, gkl_n, max_ls, log_var, training, om.y, ol.y);
      rp_bwd_elem(m.z, l.z, e.z, gz.z, gkl_n, max_ls, log_var, training, om.z, ol.z);
      rp_bwd_elem(m.w, l.w, e.w, gz.w, gkl_n, max_ls, log_var, training, om.w, ol.w);
      *reinterpret_cast<float4*>(dmu + n * ld_dmu + d) = om;
      *reinterpret_cast<float4*>(dls + n * ld_dls + d) = ol;
      am.x += om.x; am.y += om.y; am.z += om.z; am.w += om.w;
      al.x += ol.x; al.y += ol.y; al.z += ol.z; al.w += ol.w;
    }
  }
  if (part == nullptr) return;
  red[threadIdx.x] = am;
  red[RP_THREADS + threadIdx.x] = al;
  __syncthreads();
  if (rr == 0) {  // fixed order over the block's row groups
    for (int k = 1; k < rpb; ++k) {
      const float4 t1 = red[k * dq + cq], t2 = red[RP_THREADS + k * dq + cq];
      am.x += t1.x; am.y += t1.y; am.z += t1.z; am.w += t1.w;
      al.x += t2.x; al.y += t2.y; al.z += t2.z; al.w += t2.w;
    }
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * D + d) = am;
    *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 2 * D + D + d) = al;
  }
}

// column f of out = fp64 sum over the block partials in a fixed interleave (16 thread groups), as elementwise.hip
__global__ __launch_bounds__(256) void reparam_colsum_finalize_kernel(const float* __restrict__ part, int nblk, int F, float* __restrict__ out) {
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

constexpr int RP_FUSED_MAX_BLOCKS = 2048;

int rp_blocks(int64_t work) {
  int64_t b = (work + RP_THREADS - 1) / RP_THREADS;
  if (b < 1) b = 1;
  return (int)(b < RP_MAX_BLOCKS ? b : RP_MAX_BLOCKS);
}
bool rp_vec_ok(int D, int64_t a, int64_t b, const void* p0, const void* p1, const void* p2, const void* p3) {
  return (D % 4) == 0 && (a % 4) == 0 && (b % 4) == 0 && ceeg_aligned(p0, 16) && ceeg_aligned(p1, 16) &&
         (p2 == nullptr || ceeg_aligned(p2, 16)) && (p3 == nullptr || ceeg_aligned(p3, 16));
}

}  // namespace

extern "C" size_t ceeg_reparam_kl_ws_bytes(int64_t N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return ceeg_ws_part(RP_MAX_BLOCKS, 4);
}

extern "C" int ceeg_reparam_kl_fwd(const float* mu, int64_t ld_mu, const float* logstd, int64_t ld_ls, const float* eps,
                                   uint64_t seed, int64_t N, int D, float max_logstd, int log_var, int training,
                                   float* z, float* kl, void* ws, size_t ws_bytes, void* stream) {
  if (!mu || !logstd || !z || !kl || N <= 0 || D <= 0 || ld_mu < D || ld_ls < D) return CEEG_EINVAL;
  if (ws_bytes < ceeg_reparam_kl_ws_bytes(N, D) || !ws || !ceeg_aligned(ws, 256)) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vec = rp_vec_ok(D, ld_mu, ld_ls, mu, logstd, eps, z) ? 1 : 0;
  const int nblk = rp_blocks(vec ? N * (D / 4) : N * D);
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(nblk), dim3(RP_THREADS), 0, st, mu, ld_mu, logstd, ld_ls, eps, seed, N, D, max_logstd,
                     log_var, training, vec, z, part);
  CEEG_CHECK_LAUNCH();
  hipLaunchKernelGGL(reparam_kl_finalize_kernel, dim3(1), dim3(256), 0, st, part, nblk, N, kl);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_reparam_kl_bwd(const float* mu, int64_t ld_mu, const float* logstd, int64_t ld_ls, const float* eps,
                                   uint64_t seed, const float* gz, const float* gkl_dev, int64_t N, int D,
                                   float max_logstd, int log_var, int training,
                                   float* dmu, int64_t ld_dmu, float* dlogstd, int64_t ld_dls, void* stream) {
  if (!mu || !logstd || !dmu || !dlogstd || N <= 0 || D <= 0 || ld_mu < D || ld_ls < D || ld_dmu < D || ld_dls < D) return CEEG_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int vec = (rp_vec_ok(D, ld_mu, ld_ls, mu, logstd, eps, gz) && rp_vec_ok(D, ld_dmu, ld_dls, dmu, dlogstd, nullptr, nullptr)) ? 1 : 0;
  const int nblk = rp_blocks(vec ? N * (D / 4) : N * D);
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(nblk), dim3(RP_THREADS), 0, st, mu, ld_mu, logstd, ld_ls, eps, seed, gz, gkl_dev, N, D,
                     max_logstd, log_var, training, vec, dmu, ld_dmu, dlogstd, ld_dls);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" size_t ceeg_reparam_vq_bwd_ws_bytes(int64_t N, int D) {
  if (N <= 0 || D <= 0) return 0;
  return ceeg_ws_part((size_t)RP_FUSED_MAX_BLOCKS * 2 * D, 4);
}

extern "C" int ceeg_reparam_vq_bwd(const float* mu, int64_t ld_mu, const float* logstd, int64_t ld_ls, const float* eps, uint64_t seed,
                                   const float* g_zq, const float* g_z, const float* W, const int64_t* idx, int n_e,
                                   const float* s_dev, const float* gkl_dev, int64_t N, int D, float max_logstd, int log_var,
                                   int training, float* dmu, int64_t ld_dmu, float* dlogstd, int64_t ld_dls, float* colsum,
                                   void* ws, size_t ws_bytes, void* stream) {
  if (!mu || !logstd || !W || !idx || !dmu || !dlogstd || N <= 0 || D <= 0 || n_e <= 0) return CEEG_EINVAL;
  if (ld_mu < D || ld_ls < D || ld_dmu < D || ld_dls < D) return CEEG_EINVAL;
  if ((D % 4) != 0 || D > 4 * RP_THREADS) return CEEG_EUNSUPPORTED;
  if (!(rp_vec_ok(D, ld_mu, ld_ls, mu, logstd, eps, g_zq) && rp_vec_ok(D, ld_dmu, ld_dls, dmu, dlogstd, g_z, W))) return CEEG_EUNSUPPORTED;
  if (colsum && (ws_bytes < ceeg_reparam_vq_bwd_ws_bytes(N, D) || !ws || !ceeg_aligned(ws, 256))) return CEEG_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  int64_t nblk = (N + 127) / 128;
  if (nblk > RP_FUSED_MAX_BLOCKS) nblk = RP_FUSED_MAX_BLOCKS;
  const int64_t rows_per = (N + nblk - 1) / nblk;
  nblk = (N + rows_per - 1) / rows_per;
  float* part = colsum ? static_cast<float*>(ws) : nullptr;
  hipLaunchKernelGGL(reparam_vq_bwd_kernel, dim3((unsigned)nblk), dim3(RP_THREADS), 2 * RP_THREADS * 16, st, mu, ld_mu, logstd, ld_ls, eps,
                     seed, g_zq, g_z, W, idx, n_e, s_dev, gkl_dev, N, D, max_logstd, log_var, training, dmu, ld_dmu, dlogstd, ld_dls,
                     rows_per, part);
  CEEG_CHECK_LAUNCH();
  if (colsum) {
    hipLaunchKernelGGL(reparam_colsum_finalize_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, st, part, (int)nblk, 2 * D, colsum);
    CEEG_CHECK_LAUNCH();
  }
  return CEEG_OK;
}
