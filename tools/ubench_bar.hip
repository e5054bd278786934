// ubench_bar.hip -- does the host <-> device handshake of the resident sumcheck kernel get shorter when the word the DEVICE
// polls lives in device memory and the host writes it through the PCIe BAR (a posted write) instead of the device reading
// pinned host memory across PCIe (a non-posted read per poll)?  Prints the round trip for both placements.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_bar tools/ubench_bar.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void pingpong(u32 n, volatile u64* post, const volatile u64* cmd, u64 timeout) {
  for (u32 k = 1; k <= n; ++k) {
    __hip_atomic_store((u64*)&post[0], (u64)k * 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_store((u64*)&post[5], (u64)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const u64 t0 = wall_clock64();
    while (__hip_atomic_load((const u64*)&cmd[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != k) {
      if (wall_clock64() - t0 > timeout) return;
      __builtin_amdgcn_s_sleep(1);
    }
  }
}
static double run(volatile u64* post, volatile u64* cmd_dev, volatile u64* cmd_host, u32 n) {
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, 0u, post, (const volatile u64*)cmd_dev, 100000000ull);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(pingpong, dim3(1), dim3(64), 0, 0, n, post, (const volatile u64*)cmd_dev, 100000000ull);
  for (u32 k = 1; k <= n; ++k) {
    while (__atomic_load_n((u64*)&post[5], __ATOMIC_ACQUIRE) != k) {}
    __atomic_store_n((u64*)&cmd_host[2], (u64)k, __ATOMIC_RELEASE);
  }
  CK(hipDeviceSynchronize());
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  int large_bar = -1;
  hipError_t e = hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
  printf("hipDeviceAttributeIsLargeBar: %d (%s)\n", large_bar, hipGetErrorString(e));
  u64* h = nullptr;
  CK(hipHostMalloc((void**)&h, 4096, hipHostMallocCoherent | hipHostMallocMapped));
  for (int i = 0; i < 512; ++i) h[i] = 0;
  const u32 n = 3000;
  printf("cmd in pinned host memory (device polls across PCIe): %.2f us per round trip\n", run(h, h + 64, h + 64, n));
  fflush(stdout);
  if (large_bar != 1) {
    printf("no large BAR: host cannot write device memory directly\n");
    return 0;
  }
  u64* d = nullptr;
  e = hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 0;
  CK(hipMemset(d, 0, 4096));
  CK(hipDeviceSynchronize());
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, d) == hipSuccess) printf("pointer: type %d hostPointer %p devicePointer %p\n", (int)at.type, at.hostPointer, at.devicePointer);
  fflush(stdout);
  for (int i = 0; i < 512; ++i) h[i] = 0;
  printf("cmd in fine-grained device memory (host writes through the BAR): %.2f us per round trip\n", run(h, d + 64, d + 64, n));
  fflush(stdout);
  // and the post too in device memory (host polls across PCIe: expected slower)
  for (int i = 0; i < 512; ++i) h[i] = 0;
  CK(hipMemset(d, 0, 4096));
  CK(hipDeviceSynchronize());
  printf("post AND cmd in device memory (host polls the BAR): %.2f us per round trip\n", run(d, d + 64, d + 64, n));
  return 0;
}
