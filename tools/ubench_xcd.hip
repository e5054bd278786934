// ubench_xcd.hip -- what a barrier among G co-resident workgroups costs when they all sit on ONE XCD (one L2) instead of
// being dealt round-robin over the eight, and which fences it then still needs.  Every iteration is a data hand-off: each
// workgroup stores 4 KB (plain stores), barrier, loads its neighbour's 4 KB (plain loads) and checks them -- so a variant
// that is fast but hands over stale bytes shows up in the error count.
//   V0  spread   (G workgroups, blockIdx -> XCD round-robin), agent-scope release (buffer_wbl2) + acquire (buffer_inv): the
//                barrier of sc_grid_layer_kernel today
//   V1  confined (8 G workgroups launched, those with blockIdx % 8 == 0 take part; HW_REG_XCC_ID recorded), same fences
//   V2  confined, NO release fence: s_waitcnt vmcnt(0) (stores are in the shared L2), relaxed agent atomics, agent acquire
//   V3  confined, V2 with workgroup-scope atomics (executed in the XCD's L2) and sc1 polls
//   V4  spread, V2's fences (expected: stale data -- the control that shows the check can see staleness)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_xcd tools/ubench_xcd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Sync {
  u32 count, pad0[15];
  u32 gen, pad1[15];
  u32 xcc[64];
  u32 errors;
  u64 ticks;
};

// false: a wait ran out (200 ms): every workgroup leaves, the host sees the flag -- no wait is unbounded
template <int V>
__device__ __forceinline__ bool barrier(Sync* gs, u32 G, u32& gen, u64 limit) {
  __shared__ u32 dead;
  if (threadIdx.x == 0) dead = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its stores have reached the L2 before lane 0 signals for them
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 t;
    if (V == 0 || V == 1) {
      t = __hip_atomic_fetch_add(&gs->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    } else if (V == 2 || V == 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t = __hip_atomic_fetch_add(&gs->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      t = __hip_atomic_fetch_add(&gs->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (t == G - 1) {
      if (V == 3) {
        __hip_atomic_store(&gs->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&gs->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        __hip_atomic_store(&gs->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (V == 0 || V == 1) __hip_atomic_fetch_add(&gs->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(&gs->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    } else {
      const u64 t0 = wall_clock64();
      while (__hip_atomic_load(&gs->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        if (wall_clock64() - t0 > limit) {
          dead = 1;
          atomicOr(&gs->errors, 0x80000000u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  ++gen;
  __syncthreads();
  return dead == 0;
}

template <int V>
__global__ __launch_bounds__(1024) void bench(u32 n, u32 G, u32 confined, Sync* gs, u32* data /* 2 x G x 1024 words */, u64 limit) {
  extern __shared__ u32 big[];  // sized by the launch so that one workgroup fills a CU, as the real kernel's does
  u32 g = blockIdx.x;
  if (confined) {
    if (blockIdx.x & 7) return;
    g = blockIdx.x >> 3;
  }
  if (threadIdx.x == 0) {
    u32 x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    gs->xcc[g] = x & 0xf;
    big[0] = x;
  }
  u32 gen = 0, bad = 0;
  const u64 t0 = wall_clock64();
  for (u32 k = 0; k < n; ++k) {
    u32* mine = data + ((size_t)(k & 1) * G + g) * 1024;
    mine[threadIdx.x] = k * 2654435761u + g * 1024 + threadIdx.x;
    if (!barrier<V>(gs, G, gen, limit)) return;
    const u32 nb = (g + 1) % G;
    const u32* theirs = data + ((size_t)(k & 1) * G + nb) * 1024;
    if (theirs[threadIdx.x] != k * 2654435761u + nb * 1024 + threadIdx.x) ++bad;
  }
  const u64 t1 = wall_clock64();
  if (bad) atomicAdd(&gs->errors, bad);
  if (g == 0 && threadIdx.x == 0) gs->ticks = t1 - t0;
}

template <int V>
static void run(const char* name, u32 G, bool confined, u32 n, Sync* gs, u32* data, double khz) {
  CK(hipMemset(gs, 0, sizeof(Sync)));
  CK(hipMemset(data, 0xff, 2 * (size_t)G * 1024 * 4));
  const size_t lds = 112 * 1024;
  CK(hipFuncSetAttribute((const void*)bench<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(bench<V>, dim3(confined ? 8 * G : G), dim3(1024), lds, 0, n, G, confined ? 1u : 0u, gs, data, (u64)(200.0 * khz));
  CK(hipDeviceSynchronize());
  Sync h;
  CK(hipMemcpy(&h, gs, sizeof(Sync), hipMemcpyDeviceToHost));
  u32 distinct = 0, seen = 0;
  for (u32 g = 0; g < G; ++g)
    if (!(seen & (1u << h.xcc[g]))) { seen |= 1u << h.xcc[g]; ++distinct; }
  printf("%-44s G %2u: %6.2f us per hand-off, %u stale words of %llu, %u XCD(s)%s\n", name, G, (double)h.ticks / khz * 1e3 / n,
         h.errors & 0x7fffffffu, (unsigned long long)n * G * 1024, distinct, (h.errors >> 31) ? "  ** a wait TIMED OUT **" : "");
}

int main() {
  int khz_i = 0;
  CK(hipDeviceGetAttribute(&khz_i, hipDeviceAttributeWallClockRate, 0));
  const double khz = khz_i;
  Sync* gs;
  u32* data;
  CK(hipMalloc((void**)&gs, sizeof(Sync)));
  CK(hipMalloc((void**)&data, 2 * 64 * 1024 * 4));
  const u32 n = 20000;
  for (u32 G : {2u, 8u, 16u, 32u}) {
    run<0>("V0 spread, agent release + acquire (today)", G, false, n, gs, data, khz);
    run<1>("V1 one XCD, agent release + acquire", G, true, n, gs, data, khz);
    run<2>("V2 one XCD, vmcnt(0) + agent acquire", G, true, n, gs, data, khz);
    run<3>("V3 one XCD, V2 + atomics in the L2", G, true, n, gs, data, khz);
    run<4>("V4 spread, V2's fences (control)", G, false, n, gs, data, khz);
  }
  return 0;
}
