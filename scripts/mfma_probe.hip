// Measurement helper (not product): what does v_mfma_f32_16x16x32_bf16 compute, bit for bit?
// Each test t is one MFMA: A [16 x 32] bf16, B [32 x 16] bf16, C [16 x 16] fp32 -> D [16 x 16] fp32.
// Inputs are read from a file the host generates (raw bits), outputs written raw; the analysis (which summation model
// reproduces D) runs on the CPU: scripts/mfma_probe_fit.py.
//   build: hipcc --offload-arch=gfx950 -O2 -o scripts/mfma_probe scripts/mfma_probe.hip
//   run  : scripts/mfma_probe <in.bin> <out.bin> <T>
// in.bin  per test: A row-major [16][32] u16, B as [16 cols][32 k] u16 (column c's 32 k-values), C row-major [16][16] f32
// out.bin per test: D row-major [16][16] f32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;

struct Test {
  uint16_t A[16][32];
  uint16_t B[16][32];
  float C[16][16];
};

__global__ __launch_bounds__(64) void probe(const Test* __restrict__ in, float* __restrict__ out, int T) {
  const int t = blockIdx.x;
  if (t >= T) return;
  const int lane = threadIdx.x;
  const int r = lane & 15, q = lane >> 4;
  u16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = in[t].A[r][8 * q + i];  // A operand: lane (r, q) holds A[r][8q .. 8q+7]
    b[i] = in[t].B[r][8 * q + i];  // B operand: lane (c, q) holds B[8q .. 8q+7][c]
  }
  f32x4 c;
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = in[t].C[4 * q + j][r];  // C/D: lane (col = r, q) holds rows 4q .. 4q+3
  const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) out[(size_t)t * 256 + (4 * q + j) * 16 + r] = d[j];
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int T = atoi(argv[3]);
  std::vector<Test> h(T);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(h.data(), sizeof(Test), T, f) != (size_t)T) { fprintf(stderr, "bad input\n"); return 1; }
  fclose(f);
  Test* din;
  float* dout;
  if (hipMalloc(&din, sizeof(Test) * T) != hipSuccess || hipMalloc(&dout, sizeof(float) * 256 * T) != hipSuccess) return 1;
  hipMemcpy(din, h.data(), sizeof(Test) * T, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(T), dim3(64), 0, 0, din, dout, T);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
  std::vector<float> o((size_t)T * 256);
  hipMemcpy(o.data(), dout, sizeof(float) * 256 * T, hipMemcpyDeviceToHost);
  f = fopen(argv[2], "wb");
  fwrite(o.data(), sizeof(float), o.size(), f);
  fclose(f);
  printf("wrote %d tests\n", T);
  return 0;
}
