// ubench.hip -- instruction-rate microbenchmarks on gfx950 that price the field arithmetic
// (SURVEY 8d "first microbenchmarks to run on the device").  Prints G lane-ops/s per op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../longfellow-zk_amd/csrc/fields.h"

#define ITERS 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
  u32 a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a1 * 3u;
  u32 b = seed | 1u;
  u64 c0 = a0, c1 = a1, c2 = a2, c3 = a3;
  for (int i = 0; i < ITERS; ++i) {
    if (OP == 0) { a0 = a0 * b; a1 = a1 * b; a2 = a2 * b; a3 = a3 * b; }                       // v_mul_lo_u32
    if (OP == 1) { a0 = __umulhi(a0, b); a1 = __umulhi(a1, b); a2 = __umulhi(a2, b); a3 = __umulhi(a3, b); a0 |= 0x80000001u; a1 |= 0x80000001u; a2 |= 0x80000001u; a3 |= 0x80000001u; }
    if (OP == 2) { c0 = (u64)(u32)c0 * b + c0; c1 = (u64)(u32)c1 * b + c1; c2 = (u64)(u32)c2 * b + c2; c3 = (u64)(u32)c3 * b + c3; }  // v_mad_u64_u32
    if (OP == 3) { a0 = __umul24(a0, b); a1 = __umul24(a1, b); a2 = __umul24(a2, b); a3 = __umul24(a3, b); }  // v_mul_u32_u24
    if (OP == 4) { a0 ^= a1; a1 ^= a2; a2 ^= a3; a3 ^= a0; }                                      // v_xor
    if (OP == 5) { a0 = (a0 << 3) ^ a1; a1 = (a1 << 5) ^ a2; a2 = (a2 << 7) ^ a3; a3 = (a3 << 9) ^ a0; }  // shift+xor
    if (OP == 6) { a0 = a0 + a1 + a2; a1 = a1 + a2 + a3; a2 = a2 + a3 + a0; a3 = a3 + a0 + a1; }   // v_add3
    if (OP == 7) { a0 = __builtin_amdgcn_alignbit(a0, a1, 7); a1 = __builtin_amdgcn_alignbit(a1, a2, 9); a2 = __builtin_amdgcn_alignbit(a2, a3, 11); a3 = __builtin_amdgcn_alignbit(a3, a0, 13); }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (u32)c0 ^ (u32)c1 ^ (u32)c2 ^ (u32)c3 ^ (u32)(c0 >> 32);
}

template <int F>
__global__ __launch_bounds__(256) void kf(elt_t* out, u64 seed) {
  elt_t x{threadIdx.x * 0x9E3779B97F4A7C15ull + seed, seed ^ 0x1234567ull}, y{seed * 3 + 1, 0x0FFFFFFFFFFFFFFFull & (seed * 7)};
  elt_t z{x.hi, x.lo & 0x0FFFFFFFFFFFFFFFull};
  x.hi &= 0x0FFFFFFFFFFFFFFFull;
  for (int i = 0; i < ITERS / 8; ++i) {
    x = Fld<F>::mul(x, y);
    z = Fld<F>::mul(z, y);
  }
  out[blockIdx.x * 256 + threadIdx.x] = Fld<F>::add(x, z);
}

template <int OP>
double run(const char* name, int ops_per_iter) {
  u32* d;
  int blocks = 256 * 16;
  CHK(hipMalloc(&d, blocks * 256 * 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 12345u + r);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  double ops = 5.0 * blocks * 256.0 * ITERS * ops_per_iter;
  double g = ops / (ms * 1e-3) / 1e9;
  printf("%-28s %10.1f G lane-ops/s  (%.3f ms)\n", name, g, ms / 5);
  CHK(hipFree(d));
  return g;
}
template <int F>
void runf(const char* name) {
  elt_t* d;
  int blocks = 256 * 16;
  CHK(hipMalloc(&d, blocks * 256 * 16));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kf<F>, dim3(blocks), dim3(256), 0, 0, d, 99ull);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kf<F>, dim3(blocks), dim3(256), 0, 0, d, 99ull + r);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  double muls = 5.0 * blocks * 256.0 * (ITERS / 8) * 2;
  printf("%-28s %10.2f G field-mul/s (%.3f ms)\n", name, muls / (ms * 1e-3) / 1e9, ms / 5);
  CHK(hipFree(d));
}

int main() {
  run<4>("v_xor_b32", 4);
  run<5>("shl+xor (2 ops)", 8);
  run<6>("v_add3_u32", 4);
  run<7>("v_alignbit_b32", 4);
  run<3>("v_mul_u32_u24", 4);
  run<0>("v_mul_lo_u32", 4);
  run<1>("v_mul_hi_u32 (+or)", 4);
  run<2>("v_mad_u64_u32", 4);
  runf<FIELD_FP128>("fp128 montgomery mul");
  runf<FIELD_GF2_128>("gf2_128 mul (kronecker)");
  return 0;
}
