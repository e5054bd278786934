// ubench.hip -- instruction-rate microbenchmarks on gfx950 that price the field arithmetic
// (SURVEY 8d "first microbenchmarks to run on the device").  Prints G lane-ops/s per op.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../longfellow-zk_amd/csrc/fields.h"

#define ITERS 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// Every op is pinned with `asm volatile` on four independent dependence chains per lane: the round-1 version expressed
// OP 0/3/4 in C and the compiler folded the loops (impossible 140-650 T lane-ops/s lines in profiles/r01).
#define A4(INS, SRC)                                             \
  asm volatile(INS " %0, %0, " SRC "\n\t" INS " %1, %1, " SRC "\n\t" INS " %2, %2, " SRC "\n\t" INS " %3, %3, " SRC \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)          \
               : "v"(b))
template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, u32 seed) {
  u32 a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77u, a3 = a1 * 3u;
  u32 b = seed | 1u;
  u64 c0 = a0, c1 = a1, c2 = a2, c3 = a3;
  double f0 = 1.0 + a0 * 1e-10, f1 = 1.0 + a1 * 1e-10, f2 = 1.0 + a2 * 1e-10, f3 = 1.0 + a3 * 1e-10, fb = 1.0 + seed * 1e-12, fc = 1e-9;
  float g0 = 1.0f + a0 * 1e-10f, g1 = 1.0f + a1 * 1e-10f, g2 = 1.0f + a2 * 1e-10f, g3 = 1.0f + a3 * 1e-10f, gb = 1.0f + seed * 1e-9f, gc = 1e-6f;
#pragma unroll 8
  for (int i = 0; i < ITERS; ++i) {
    if (OP == 0) A4("v_mul_lo_u32", "%4");
    if (OP == 1) A4("v_mul_hi_u32", "%4");
    if (OP == 2)
      asm volatile("v_mad_u64_u32 %0, vcc, %4, %4, %0\n\tv_mad_u64_u32 %1, vcc, %4, %4, %1\n\tv_mad_u64_u32 %2, vcc, %4, %4, %2\n\tv_mad_u64_u32 %3, vcc, %4, %4, %3"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b) : "vcc");
    if (OP == 3) A4("v_mul_u32_u24", "%4");
    if (OP == 4) A4("v_xor_b32", "%4");
    if (OP == 5) A4("v_mul_hi_u32_u24", "%4");
    if (OP == 6)
      asm volatile("v_add3_u32 %0, %0, %4, %1\n\tv_add3_u32 %1, %1, %4, %2\n\tv_add3_u32 %2, %2, %4, %3\n\tv_add3_u32 %3, %3, %4, %0"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    if (OP == 7)
      asm volatile("v_alignbit_b32 %0, %0, %1, 7\n\tv_alignbit_b32 %1, %1, %2, 9\n\tv_alignbit_b32 %2, %2, %3, 11\n\tv_alignbit_b32 %3, %3, %0, 13"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    if (OP == 8)  // 64-bit add as the carry pair: 2 instructions per add
      asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc\n\tv_add_co_u32 %2, vcc, %2, %0\n\tv_addc_co_u32 %3, vcc, %3, %1, vcc"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");
    if (OP == 9)  // gfx940+: 64-bit shift-add in one instruction (no carry out)
      asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %4\n\tv_lshl_add_u64 %2, %2, 0, %4\n\tv_lshl_add_u64 %3, %3, 0, %4"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(c0 | 1));
    if (OP == 10)
      asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fb), "v"(fc));
    if (OP == 11)
      asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                   : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(gb), "v"(gc));
    if (OP == 12)
      asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n\tv_mad_u32_u24 %1, %1, %4, %2\n\tv_mad_u32_u24 %2, %2, %4, %3\n\tv_mad_u32_u24 %3, %3, %4, %0"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    if (OP == 13)
      asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4"
                   : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fc));
    if (OP == 15)  // a ^= (x & m): three-input boolean op (truth table 0x78 = A ^ (B & C)), mask in a VGPR
      asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x78\n\tv_bitop3_b32 %1, %1, %4, %5 bitop3:0x78\n\tv_bitop3_b32 %2, %2, %4, %5 bitop3:0x78\n\tv_bitop3_b32 %3, %3, %4, %5 bitop3:0x78"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(seed ^ 0x55u));
    if (OP == 16)  // ... mask in an SGPR (wave-uniform)
      asm volatile("v_bitop3_b32 %0, %0, %4, %5 bitop3:0x78\n\tv_bitop3_b32 %1, %1, %4, %5 bitop3:0x78\n\tv_bitop3_b32 %2, %2, %4, %5 bitop3:0x78\n\tv_bitop3_b32 %3, %3, %4, %5 bitop3:0x78"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "s"(seed ^ 0x55u));
    if (OP == 17)  // the two-instruction form: v_and + v_xor
      asm volatile("v_and_b32 %4, %0, %5\n\tv_xor_b32 %1, %1, %4\n\tv_and_b32 %4, %2, %5\n\tv_xor_b32 %3, %3, %4"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b) : "v"(seed ^ 0x55u));
    if (OP == 14)  // v_mad_u64_u32 with the accumulate chain only through the 64-bit addend (the schoolbook-row shape)
      asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %6, %1\n\tv_mad_u64_u32 %2, vcc, %4, %7, %2\n\tv_mad_u64_u32 %3, vcc, %4, %8, %3"
                   : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(b), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (u32)c0 ^ (u32)c1 ^ (u32)c2 ^ (u32)c3 ^ (u32)(c0 >> 32) ^ (u32)(c1 >> 32) ^
                                        (u32)__double_as_longlong(f0 + f1 + f2 + f3) ^ __float_as_uint(g0 + g1 + g2 + g3);
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

// the same loop under a given occupancy: `lds` bytes of dynamic LDS per 256-thread workgroup bound the workgroups per CU
// (160 KiB per CU), i.e. waves per SIMD = workgroups per CU
template <int OP>
double run_occ(const char* name, int ops_per_iter, size_t lds) {
  u32* d;
  int blocks = 256 * 16;
  CHK(hipMalloc(&d, blocks * 256 * 4));
  CHK(hipFuncSetAttribute((const void*)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, 12345u);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), lds, 0, d, 12345u + r);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  double g = 5.0 * blocks * 256.0 * ITERS * ops_per_iter / (ms * 1e-3) / 1e9;
  printf("%-22s %6zu KiB LDS/WG (%zu waves/SIMD) %10.1f G lane-ops/s\n", name, lds >> 10, lds ? (160 * 1024) / lds : 8, g);
  CHK(hipFree(d));
  return g;
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
  run<6>("v_add3_u32", 4);
  run<7>("v_alignbit_b32", 4);
  run<8>("v_add_co+v_addc_co (2 instr)", 4);
  run<9>("v_lshl_add_u64", 4);
  run<3>("v_mul_u32_u24", 4);
  run<5>("v_mul_hi_u32_u24", 4);
  run<12>("v_mad_u32_u24", 4);
  run<0>("v_mul_lo_u32", 4);
  run<1>("v_mul_hi_u32", 4);
  run<2>("v_mad_u64_u32 (b*b+acc)", 4);
  run<14>("v_mad_u64_u32 (a*b+acc)", 4);
  run<11>("v_fma_f32", 4);
  run<10>("v_fma_f64", 4);
  run<13>("v_add_f64", 4);
  for (size_t lds : {(size_t)160 * 1024, (size_t)80 * 1024, (size_t)53 * 1024, (size_t)40 * 1024, (size_t)20 * 1024}) {
    run_occ<4>("v_xor_b32", 4, lds);
    run_occ<8>("v_add_co+v_addc_co", 4, lds);
    run_occ<14>("v_mad_u64_u32", 4, lds);
    run_occ<15>("v_bitop3_b32 (vgpr mask)", 4, lds);
    run_occ<16>("v_bitop3_b32 (sgpr mask)", 4, lds);
    run_occ<17>("v_and_b32 + v_xor_b32", 4, lds);
  }
  runf<FIELD_FP128>("fp128 montgomery mul");
  runf<FIELD_GF2_128>("gf2_128 mul (kronecker)");
  return 0;
}
