// coop_exit_repro.hip -- smallest program that separates the candidate causes of the SIGSEGV seen at process exit
// under `rocprofv3 --kernel-trace` (round-1 logs gpurun_out/prof_zk{1,3,4}.log: stack exit -> atexit handler -> profiler
// library).  One variant per run; each writes /proc/self/maps to argv[2] just before returning from main so the
// crashing frames can be resolved.
//   plain       one ordinary launch
//   coop        one hipLaunchCooperativeKernel
//   coop_reset  the same, then hipDeviceReset() before exit
//   hostpoll    an ordinary launch that handshakes with the host through coherent mapped memory while it runs
//   hostpoll_nofree  the same, without hipHostFree of the mapped buffer
// build: hipcc --offload-arch=gfx950 -O2 -o tools/coop_exit_repro tools/coop_exit_repro.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <fstream>

__global__ void k_plain(unsigned* out) { out[threadIdx.x] = threadIdx.x; }

__global__ void k_poll(volatile unsigned long long* post, const volatile unsigned long long* cmd, unsigned long long ticks) {
  if (threadIdx.x) return;
  __hip_atomic_store((unsigned long long*)&post[0], 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
    if (__hip_atomic_load((const unsigned long long*)&cmd[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 1ull) break;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_store((unsigned long long*)&post[0], 2ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                   \
      return 2;                                                                \
    }                                                                          \
  } while (0)

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "plain";
  unsigned* d = nullptr;
  CK(hipMalloc(&d, 4096));
  hipStream_t s = nullptr;
  if (!strcmp(mode, "plain")) {
    hipLaunchKernelGGL(k_plain, dim3(1), dim3(64), 0, s, d);
  } else if (!strncmp(mode, "coop", 4)) {
    void* args[] = {&d};
    CK(hipLaunchCooperativeKernel((const void*)k_plain, dim3(1), dim3(64), args, 0, s));
  } else if (!strncmp(mode, "hostpoll", 8)) {
    void* ph = nullptr;
    CK(hipHostMalloc(&ph, 4096, hipHostMallocCoherent | hipHostMallocMapped));
    memset(ph, 0, 4096);
    volatile unsigned long long* post = (volatile unsigned long long*)ph;
    volatile unsigned long long* cmd = post + 64;
    int khz = 100000;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    hipLaunchKernelGGL(k_poll, dim3(1), dim3(64), 0, s, post, cmd, 200ull * (unsigned long long)khz);
    unsigned long long spins = 0;
    while (__atomic_load_n((const unsigned long long*)&post[0], __ATOMIC_ACQUIRE) == 0 && ++spins < (1ull << 32)) {
    }
    const bool seen_while_running = __atomic_load_n((const unsigned long long*)&post[0], __ATOMIC_ACQUIRE) == 1;
    __atomic_store_n((unsigned long long*)&cmd[0], 1ull, __ATOMIC_RELEASE);
    CK(hipStreamSynchronize(s));
    printf("hostpoll: first post seen while the kernel ran: %d\n", (int)seen_while_running);
    if (strcmp(mode, "hostpoll_nofree")) CK(hipHostFree(ph));
  }
  CK(hipStreamSynchronize(s));
  CK(hipFree(d));
  if (!strcmp(mode, "coop_reset")) CK(hipDeviceReset());
  if (argc > 2) {
    std::ifstream in("/proc/self/maps");
    std::ofstream out(argv[2]);
    out << in.rdbuf();
  }
  printf("%s: main returns\n", mode);
  return 0;
}
