// ubench_p256.hip -- latency (one wave, a chain of dependent products) and throughput (full occupancy) of the Fp256Base
// Montgomery product in csrc/fp256.h (and of a second variant, if the header defines FP256_HAVE_MUL2 / fp256_mul2).
// Measured on MI355X: 820 ns per dependent product on one wave (~350 instructions at one issue per ~5.6 cycles: a lone wave's
// issue rate, not the dependency chain, sets it), 111.5 G products/s at full occupancy.  A variant with two independent
// operand-scanning chains (more instruction-level parallelism, ~15 % more instructions) measured 972 ns / 98.7 G/s and was
// dropped: only fewer instructions would shorten the sumcheck's chains.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I. tools/ubench_p256.hip -o tools/ubench_p256
#include <cstdio>
#include <hip/hip_runtime.h>

#include "../longfellow-zk_amd/csrc/fp256.h"

#define CHK(x)                                                                      \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

template <int V>
__device__ __forceinline__ elt32_t mulv(const elt32_t& a, const elt32_t& b) {
  if (V == 0) return fp256_mul(a, b);
#ifdef FP256_HAVE_MUL2
  return fp256_mul2(a, b);
#else
  return fp256_mul(a, b);
#endif
}
template <int V>
__global__ void chain_kernel(elt32_t* d, int iters) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  elt32_t x = d[2 * i], y = d[2 * i + 1];
  for (int k = 0; k < iters; ++k) {
    x = mulv<V>(x, y);
    y = mulv<V>(y, x);
  }
  d[2 * i] = x;
  d[2 * i + 1] = y;
}
template <int V>
int run(const char* name) {
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    const int blocks = mode ? 256 * 8 : 1, threads = mode ? 256 : 64;
    elt32_t* d;
    const size_t n = (size_t)blocks * threads * 2;
    CHK(hipMalloc(&d, n * 32));
    CHK(hipMemset(d, 0x5a, n * 32));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(chain_kernel<V>, dim3(blocks), dim3(threads), 0, 0, d, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(chain_kernel<V>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (mode == 0) printf("%-26s latency   %8.1f ns per dependent product (one wave)\n", name, ms * 1e6 / (2.0 * iters));
    else printf("%-26s throughput %7.1f G products/s (%d x %d threads)\n", name, 2.0 * iters * blocks * threads / (ms * 1e-3) / 1e9, blocks, threads);
    CHK(hipFree(d));
  }
  return 0;
}
int main() {
  if (run<0>("fp256_mul")) return 1;
#ifdef FP256_HAVE_MUL2
  if (run<1>("fp256_mul2")) return 1;
#endif
  return 0;
}
