// Learned-attention token pooling of the reference's conditioners (scripts/models.py:118-119,
// also :138-139,:170-171):   a[n,t] = <hidden[n,t,:], w> + b;   x[n,:] = sum_t a[n,t] hidden[n,t,:]
// (attn = nn.Linear(H, 1); x = bmm(hidden^T, attn(hidden))).  One workgroup per node reads its [T, H] tile ONCE from HBM
// (the second sweep over the tile is served by L2) instead of once for the linear and once for the bmm.
// Backward:  da[t] = <g, hidden[t,:]>;  dhidden[t,:] = a[t] g + da[t] w;  dw = sum_{n,t} da[t] hidden[n,t,:]  (per-node partials,
// summed over nodes by ceeg_colsum in a fixed order);  db = sum_{n,t} da[t].
#include "ceeg_internal.h"

namespace {
constexpr int TP_THREADS = 256;
constexpr int TP_MAXT = 512;

// dots of every token row with a vector v (H floats): wave w takes tokens w, w+4, ...; result in s_out[t]
__device__ __forceinline__ void tp_token_dots(const float* __restrict__ tile, int T, int H, const float* __restrict__ v, float add, float* s_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = wave; t < T; t += TP_THREADS / 64) {
    const float* row = tile + (size_t)t * H;
    float acc = 0.f;
    for (int h = 4 * lane; h < H; h += 256) {
      const float4 x = *reinterpret_cast<const float4*>(row + h);
      const float4 y = *reinterpret_cast<const float4*>(v + h);
      acc = __fmaf_rn(x.x, y.x, acc); acc = __fmaf_rn(x.y, y.y, acc); acc = __fmaf_rn(x.z, y.z, acc); acc = __fmaf_rn(x.w, y.w, acc);
    }
    acc = ceeg_wave_sum(acc);
    if (lane == 0) s_out[t] = acc + add;
  }
}

__global__ __launch_bounds__(TP_THREADS) void token_pool_fwd_kernel(const float* __restrict__ hidden, int T, int H, const float* __restrict__ w,
                                                                    const float* __restrict__ b, float* __restrict__ a_out, float* __restrict__ x) {
  __shared__ float s_a[TP_MAXT];
  const int64_t n = blockIdx.x;
  const float* tile = hidden + (size_t)n * T * H;
  tp_token_dots(tile, T, H, w, b ? b[0] : 0.f, s_a);
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += TP_THREADS) a_out[n * T + t] = s_a[t];
  for (int h = 4 * threadIdx.x; h < H; h += 4 * TP_THREADS) {  // tokens ascending: fixed order
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
      const float4 v = *reinterpret_cast<const float4*>(tile + (size_t)t * H + h);
      const float at = s_a[t];
      acc.x = __fmaf_rn(at, v.x, acc.x); acc.y = __fmaf_rn(at, v.y, acc.y); acc.z = __fmaf_rn(at, v.z, acc.z); acc.w = __fmaf_rn(at, v.w, acc.w);
    }
    *reinterpret_cast<float4*>(x + n * H + h) = acc;
  }
}

__global__ __launch_bounds__(TP_THREADS) void token_pool_bwd_kernel(const float* __restrict__ hidden, int T, int H, const float* __restrict__ w,
                                                                    const float* __restrict__ a, const float* __restrict__ g,
                                                                    float* __restrict__ dhidden, float* __restrict__ dw_part /* [N][H] */,
                                                                    float* __restrict__ db_part /* [N][4] */) {
  __shared__ float s_da[TP_MAXT];
  __shared__ float s_red[TP_THREADS / 64];
  const int64_t n = blockIdx.x;
  const float* tile = hidden + (size_t)n * T * H;
  tp_token_dots(tile, T, H, g + n * H, 0.f, s_da);
  __syncthreads();
  for (int h = 4 * threadIdx.x; h < H; h += 4 * TP_THREADS) {
    const float4 gv = *reinterpret_cast<const float4*>(g + n * H + h);
    const float4 wv = *reinterpret_cast<const float4*>(w + h);
    float4 dw = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {
      const float at = a[n * T + t], dt = s_da[t];
      if (dhidden) {
        float4 o = make_float4(__fmaf_rn(dt, wv.x, at * gv.x), __fmaf_rn(dt, wv.y, at * gv.y), __fmaf_rn(dt, wv.z, at * gv.z), __fmaf_rn(dt, wv.w, at * gv.w));
        *reinterpret_cast<float4*>(dhidden + ((size_t)n * T + t) * H + h) = o;
      }
      const float4 v = *reinterpret_cast<const float4*>(tile + (size_t)t * H + h);
      dw.x = __fmaf_rn(dt, v.x, dw.x); dw.y = __fmaf_rn(dt, v.y, dw.y); dw.z = __fmaf_rn(dt, v.z, dw.z); dw.w = __fmaf_rn(dt, v.w, dw.w);
    }
    *reinterpret_cast<float4*>(dw_part + n * H + h) = dw;
  }
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += TP_THREADS) s += s_da[t];
  s = ceeg_wave_sum(s);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < TP_THREADS / 64; ++i) tot += s_red[i];
    *reinterpret_cast<float4*>(db_part + n * 4) = make_float4(tot, 0.f, 0.f, 0.f);
  }
}

int tp_check(const float* hidden, int64_t N, int T, int H, const float* w) {
  if (!hidden || !w || N <= 0 || T <= 0 || H <= 0) return CEEG_EINVAL;
  if ((H % 4) != 0 || T > TP_MAXT || N >= ((int64_t)1 << 31)) return CEEG_EUNSUPPORTED;
  if (!ceeg_aligned(hidden, 16) || !ceeg_aligned(w, 16)) return CEEG_EINVAL;
  return CEEG_OK;
}
}  // namespace

extern "C" int ceeg_token_pool_fwd(const float* hidden, int64_t N, int T, int H, const float* w, const float* b, float* a, float* x, void* stream) {
  const int rc = tp_check(hidden, N, T, H, w);
  if (rc != CEEG_OK) return rc;
  if (!a || !x || !ceeg_aligned(x, 16)) return CEEG_EINVAL;
  hipLaunchKernelGGL(token_pool_fwd_kernel, dim3((unsigned)N), dim3(TP_THREADS), 0, static_cast<hipStream_t>(stream), hidden, T, H, w, b, a, x);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}

extern "C" int ceeg_token_pool_bwd(const float* hidden, int64_t N, int T, int H, const float* w, const float* a, const float* g, float* dhidden,
                                   float* dw_part, float* db_part, void* stream) {
  const int rc = tp_check(hidden, N, T, H, w);
  if (rc != CEEG_OK) return rc;
  if (!a || !g || !dw_part || !db_part) return CEEG_EINVAL;
  if (!ceeg_aligned(g, 16) || !ceeg_aligned(dw_part, 16) || !ceeg_aligned(db_part, 16) || (dhidden && !ceeg_aligned(dhidden, 16))) return CEEG_EINVAL;
  hipLaunchKernelGGL(token_pool_bwd_kernel, dim3((unsigned)N), dim3(TP_THREADS), 0, static_cast<hipStream_t>(stream), hidden, T, H, w, a, g, dhidden,
                     dw_part, db_part);
  CEEG_CHECK_LAUNCH();
  return CEEG_OK;
}
