// Micro-benchmark (GPU box): rocPRIM keys-only radix sort of 2E incidence keys on bits [10, 10+20) with the library's default
// onesweep configuration against wider digits (fewer passes).  hipcc --offload-arch=gfx950 -O3 sort_cfg.hip -o sort_cfg
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class Cfg>
float run(const char* name, const uint32_t* in, uint32_t* out, size_t M, unsigned b0, unsigned b1, const uint32_t* ref) {
  size_t tb = 0;
  CK(rocprim::radix_sort_keys<Cfg>(nullptr, tb, in, out, M, b0, b1, (hipStream_t)0, false));
  void* temp;
  CK(hipMalloc(&temp, tb));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(rocprim::radix_sort_keys<Cfg>(temp, tb, in, out, M, b0, b1, (hipStream_t)0, false));
  CK(hipEventRecord(a));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) CK(rocprim::radix_sort_keys<Cfg>(temp, tb, in, out, M, b0, b1, (hipStream_t)0, false));
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  int same = -1;
  if (ref) {
    std::vector<uint32_t> h(M), r(M);
    CK(hipMemcpy(h.data(), out, M * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(r.data(), ref, M * 4, hipMemcpyDeviceToHost));
    same = 1;
    for (size_t i = 0; i < M; ++i) if (h[i] != r[i]) { same = 0; break; }
  }
  printf("%-34s %.3f ms  temp %.1f MB  equals default: %d\n", name, ms / reps, tb / 1e6, same);
  CK(hipFree(temp));
  return ms / reps;
}

template <unsigned BITS, unsigned BS, unsigned IPT, rocprim::block_radix_rank_algorithm ALG = rocprim::block_radix_rank_algorithm::default_algorithm>
using Cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                       rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;

int main(int argc, char** argv) {
  const size_t M = argc > 1 ? strtoull(argv[1], nullptr, 10) : 32000000;
  const unsigned N = argc > 2 ? strtoul(argv[2], nullptr, 10) : 1000000;
  unsigned nb = 1;
  while ((1ull << nb) <= N) ++nb;
  std::vector<uint32_t> h(M);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < M; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    h[i] = (uint32_t)((s >> 20) % N) << 10 | (uint32_t)(s & 511);
  }
  uint32_t *in, *ref, *out;
  CK(hipMalloc(&in, M * 4)); CK(hipMalloc(&ref, M * 4)); CK(hipMalloc(&out, M * 4));
  CK(hipMemcpy(in, h.data(), M * 4, hipMemcpyHostToDevice));
  printf("M = %zu keys, node bits %u\n", M, nb);
  run<rocprim::default_config>("default", in, ref, M, 10, 10 + nb, nullptr);
  constexpr auto MATCH = rocprim::block_radix_rank_algorithm::match;
  run<Cfg<8, 1024, 8, MATCH>>("8 bits, 1024 x 8", in, out, M, 10, 10 + nb, ref);
  run<Cfg<8, 512, 16, MATCH>>("8 bits, 512 x 16", in, out, M, 10, 10 + nb, ref);
  run<Cfg<8, 1024, 16, MATCH>>("8 bits, 1024 x 16", in, out, M, 10, 10 + nb, ref);
  run<Cfg<10, 1024, 8, MATCH>>("10 bits, 1024 x 8", in, out, M, 10, 10 + nb, ref);
  run<Cfg<10, 1024, 12, MATCH>>("10 bits, 1024 x 12", in, out, M, 10, 10 + nb, ref);
  run<Cfg<10, 1024, 16, MATCH>>("10 bits, 1024 x 16", in, out, M, 10, 10 + nb, ref);
  run<Cfg<10, 1024, 22, MATCH>>("10 bits, 1024 x 22", in, out, M, 10, 10 + nb, ref);
  run<Cfg<10, 512, 16, MATCH>>("10 bits, 512 x 16", in, out, M, 10, 10 + nb, ref);
  run<Cfg<10, 512, 24, MATCH>>("10 bits, 512 x 24", in, out, M, 10, 10 + nb, ref);
  run<Cfg<10, 256, 24, MATCH>>("10 bits, 256 x 24", in, out, M, 10, 10 + nb, ref);
  run<Cfg<11, 1024, 12, MATCH>>("11 bits, 1024 x 12", in, out, M, 10, 10 + nb, ref);
  run<Cfg<11, 1024, 16, MATCH>>("11 bits, 1024 x 16", in, out, M, 10, 10 + nb, ref);
  run<Cfg<11, 512, 24, MATCH>>("11 bits, 512 x 24", in, out, M, 10, 10 + nb, ref);
  return 0;
}
