// Micro-benchmark (GPU box): LDS read rate of the decoder gradient's access shape - every 16-lane group of a wave reads one 256-byte
// row segment at a random row of a 128 KB slab - as one ds_read_b128 per lane, two ds_read_b64 or four ds_read_b32, with and without
// a per-group rotation of the pieces.  hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 512, ROWB = 256, ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(1024, 1) void k(const int* __restrict__ rows, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < ROWS * ROWB / 4; i += 1024) reinterpret_cast<float*>(lds)[i] = (float)(i & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int r = rows[(blockIdx.x * 1024 + threadIdx.x) & 65535];
  for (int it = 0; it < ITERS; ++it) {
    const int row = __shfl(r, (grp * 16 + (it & 15)), 64) & (ROWS - 1);   // one row per 16-lane group and step
    const char* p = lds + row * ROWB;
    if (MODE == 0) {          // one 16-byte piece per lane
      const float4 v = *reinterpret_cast<const float4*>(p + gl * 16);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    } else if (MODE == 1) {   // two 8-byte pieces per lane: [0,128) and [128,256)
      const float2 a = *reinterpret_cast<const float2*>(p + gl * 8);
      const float2 b = *reinterpret_cast<const float2*>(p + 128 + gl * 8);
      acc.x += a.x; acc.y += a.y; acc.z += b.x; acc.w += b.y;
    } else if (MODE == 2) {   // the same, odd groups take the halves in the other order
      const int h = (grp & 1) * 128;
      const float2 a = *reinterpret_cast<const float2*>(p + h + gl * 8);
      const float2 b = *reinterpret_cast<const float2*>(p + (128 - h) + gl * 8);
      acc.x += a.x; acc.y += a.y; acc.z += b.x; acc.w += b.y;
    } else if (MODE == 3) {   // four 4-byte pieces per lane, quarter q of group g = (q + g) & 3
      float t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) t[q] = *reinterpret_cast<const float*>(p + (((q + grp) & 3) * 64) + gl * 4);
      acc.x += t[0]; acc.y += t[1]; acc.z += t[2]; acc.w += t[3];
    } else if (MODE == 4) {   // b128, rows of the four groups forced to the SAME row (broadcast-like: no bank spread)
      const int row0 = __shfl(r, (it & 15), 64) & (ROWS - 1);
      const float4 v = *reinterpret_cast<const float4*>(lds + row0 * ROWB + gl * 16);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    } else if (MODE == 5) {   // b128, 64 lanes read 1 KB contiguous (4 consecutive rows)
      const float4 v = *reinterpret_cast<const float4*>(lds + ((row & ~3) * ROWB) + lane * 16);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if ((it & 15) == 15) r = r * 1103515245 + 12345 + lane;
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
void run(const char* name, const int* rows, float* out, double clk_ghz) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), ROWS * ROWB, 0, rows, out);
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), ROWS * ROWB, 0, rows, out);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= 5;
  const double bytes_per_cu = 16.0 * ITERS * 1024.0;   // 16 waves x ITERS steps x 1 KB
  printf("%-52s %.3f ms  %.1f bytes / clock / CU at %.1f GHz\n", name, ms, bytes_per_cu / (ms * 1e-3 * clk_ghz * 1e9), clk_ghz);
}

int main() {
  int* rows; float* out;
  CK(hipMalloc(&rows, 65536 * 4)); CK(hipMalloc(&out, 256 * 1024 * 4));
  int* h = (int*)malloc(65536 * 4);
  for (int i = 0; i < 65536; ++i) h[i] = rand();
  CK(hipMemcpy(rows, h, 65536 * 4, hipMemcpyHostToDevice));
  const double g = 2.4;
  run<0>("b128: 16-byte piece per lane, 4 random rows per wave", rows, out, g);
  run<1>("2 x b64: halves in the same order in every group", rows, out, g);
  run<2>("2 x b64: odd groups take the halves in the other order", rows, out, g);
  run<3>("4 x b32: quarters rotated by the group number", rows, out, g);
  run<4>("b128: the four groups on the SAME row", rows, out, g);
  run<5>("b128: 64 lanes on 1 KB contiguous", rows, out, g);
  return 0;
}
