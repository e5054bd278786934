// ubench_sync.hip -- latency of the two synchronisation primitives of the resident sumcheck kernel on this system:
//   (1) device <-> host handshake through coherent pinned memory (kernel posts, host answers)
//   (2) device-wide barrier among G resident workgroups (atomic counter + generation)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_sync tools/ubench_sync.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void pingpong(u32 n, volatile u64* post, const volatile u64* cmd, u64 timeout) {
  for (u32 k = 1; k <= n; ++k) {
    post[0] = k * 3;
    __threadfence_system();
    __hip_atomic_store((u64*)&post[5], (u64)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const u64 t0 = wall_clock64();
    while (__hip_atomic_load((const u64*)&cmd[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != k) {
      if (wall_clock64() - t0 > timeout) return;
      __builtin_amdgcn_s_sleep(4);
    }
  }
}
struct Sync { u32 count, gen; };
__global__ void barriers(u32 n, Sync* gs, u32 work) {
  u32 gen = 0;
  u32 acc = threadIdx.x;
  for (u32 k = 0; k < n; ++k) {
    for (u32 w = 0; w < work; ++w) acc = acc * 1664525u + 1013904223u;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const u32 t = __hip_atomic_fetch_add(&gs->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gridDim.x - 1) {
        __hip_atomic_store(&gs->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&gs->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(&gs->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(2);
      }
      __threadfence();
    }
    ++gen;
    __syncthreads();
  }
  if (acc == 0x12345678u) gs->count = 99;
}
int main() {
  u64* h = nullptr;
  CK(hipHostMalloc((void**)&h, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  for (int i = 0; i < 512; ++i) h[i] = 0;
  volatile u64* post = h;
  volatile u64* cmd = h + 64;
  const u32 n = 2000;
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, 0u, post, (const volatile u64*)cmd, 100000000ull);  // load the code object
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, n, post, (const volatile u64*)cmd, 100000000ull);
  for (u32 k = 1; k <= n; ++k) {
    while (__atomic_load_n((u64*)&post[5], __ATOMIC_ACQUIRE) != k) {}
    __atomic_store_n((u64*)&cmd[2], (u64)k, __ATOMIC_RELEASE);
  }
  CK(hipDeviceSynchronize());
  auto t1 = std::chrono::steady_clock::now();
  printf("host<->device handshake: %.2f us per round trip\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
  Sync* gs;
  CK(hipMalloc((void**)&gs, 64));
  for (u32 G : {2u, 8u, 32u, 128u, 256u}) {
    CK(hipMemset(gs, 0, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const u32 nb = 2000;
    u32 work = 0;
    void* args[] = {(void*)&nb, (void*)&gs, (void*)&work};
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel((const void*)barriers, dim3(G), dim3(1024), args, 0, 0));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("device barrier, %3u workgroups x 1024 threads: %.2f us\n", G, ms * 1000 / nb);
  }
  return 0;
}
