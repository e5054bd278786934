// ubench_dispatch.hip -- what K streams of SMALL kernels cost each other on one MI355X (throughput mode).
// K host threads x own non-blocking stream; every thread runs `n` tiny kernels back to back (optionally a stream
// synchronisation every `sync_every` launches, optionally a hipMemsetAsync / small D2H copy in the mix), with R further
// streams each holding a long-lived spinning kernel of `rg` workgroups (the shape of a resident sumcheck grid).
// Prints aggregate dispatches/s and the mean time per dispatch seen by one stream.
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/ubench_dispatch.hip -o tools/ubench_dispatch
//   GPU_MAX_HW_QUEUES=16 tools/ubench_dispatch
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void tiny_kernel(unsigned* p, unsigned n) {  // touches n words per workgroup
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) p[blockIdx.x * n + i] += i;
}

// a kernel of known length: every workgroup spins `ticks` of the 100 MHz clock, then touches one word
__global__ void busy_kernel(unsigned* p, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
  if (threadIdx.x == 0) p[blockIdx.x * 1024] += 1;
}
// a kernel that reads `words` words per workgroup eight times over (memory-bound; L2-resident when alone)
__global__ void stream_kernel(unsigned* p, unsigned words) {
  unsigned acc = 0;
  for (int rep = 0; rep < 8; ++rep)
    for (unsigned i = threadIdx.x; i < words; i += blockDim.x) acc += p[(size_t)blockIdx.x * words + i];
  if (acc == 0xdeadbeefu) p[0] = acc;
}

// a resident grid: one lane per workgroup polls a device word (agent-scope loads + s_sleep) until the host clears it or
// `ticks` of the 100 MHz clock pass
__global__ __launch_bounds__(256) void spin_kernel(const unsigned* flag, unsigned long long ticks, int fence_every, int nap) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = wall_clock64();
    unsigned it = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 && wall_clock64() - t0 < ticks) {
      if (nap <= 1) __builtin_amdgcn_s_sleep(1);
      else if (nap <= 8) __builtin_amdgcn_s_sleep(8);
      else if (nap <= 32) __builtin_amdgcn_s_sleep(32);
      else __builtin_amdgcn_s_sleep(127);
      if (fence_every && (++it % fence_every) == 0) __threadfence();
    }
  }
  __syncthreads();
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Case {
  int K, n, sync_every, grid, mix, R, rg, fence_every;
  int nap = 1;      // s_sleep argument of the spinners' poll loop (64 clocks each)
  int busy_us = 0;  // > 0: busy_kernel of that length instead of tiny_kernel; < 0: stream_kernel over -busy_us KiB per workgroup
};

static void run(const Case& cs) {
  std::vector<hipStream_t> st(cs.K), rs(cs.R);
  std::vector<unsigned*> buf(cs.K);
  std::vector<unsigned*> pin(cs.K);
  for (int i = 0; i < cs.K; ++i) {
    CHK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
    const size_t bytes = (size_t)cs.grid * (cs.busy_us < 0 ? (size_t)(-cs.busy_us) * 1024 : 4096);
    CHK(hipMalloc(&buf[i], bytes));
    CHK(hipMemset(buf[i], 0, bytes));
    CHK(hipHostMalloc(&pin[i], 4096));
  }
  unsigned* flag = nullptr;
  CHK(hipMalloc(&flag, 4));
  unsigned one = 1, zero = 0;
  CHK(hipMemcpy(flag, &one, 4, hipMemcpyHostToDevice));
  for (int i = 0; i < cs.R; ++i) {
    CHK(hipStreamCreateWithFlags(&rs[i], hipStreamNonBlocking));
    hipLaunchKernelGGL(spin_kernel, dim3(cs.rg), dim3(256), 0, rs[i], flag, 100000000ull * 20, cs.fence_every, cs.nap);  // 20 s bound
  }
  std::atomic<int> go{0};
  std::vector<double> secs(cs.K);
  std::vector<std::thread> th;
  for (int i = 0; i < cs.K; ++i)
    th.emplace_back([&, i] {
      CHK(hipSetDevice(0));
      while (!go.load()) {}
      const double t0 = now();
      for (int j = 0; j < cs.n; ++j) {
        if (cs.mix && j % 3 == 1) CHK(hipMemsetAsync(buf[i], 0, 4096, st[i]));
        else if (cs.mix && j % 3 == 2) CHK(hipMemcpyAsync(pin[i], buf[i], 64, hipMemcpyDeviceToHost, st[i]));
        else if (cs.busy_us > 0) hipLaunchKernelGGL(busy_kernel, dim3(cs.grid), dim3(256), 0, st[i], buf[i], (unsigned long long)cs.busy_us * 100);
        else if (cs.busy_us < 0) hipLaunchKernelGGL(stream_kernel, dim3(cs.grid), dim3(256), 0, st[i], buf[i], (unsigned)(-cs.busy_us) * 256u);
        else hipLaunchKernelGGL(tiny_kernel, dim3(cs.grid), dim3(256), 0, st[i], buf[i], 1024u);
        if (cs.sync_every && (j + 1) % cs.sync_every == 0) CHK(hipStreamSynchronize(st[i]));
      }
      CHK(hipStreamSynchronize(st[i]));
      secs[i] = now() - t0;
    });
  const double t0 = now();
  go.store(1);
  for (auto& t : th) t.join();
  const double wall = now() - t0;
  CHK(hipMemcpyAsync(flag, &zero, 4, hipMemcpyHostToDevice, st[0]));  // a blocking hipMemcpy would wait for the spinners' streams
  CHK(hipStreamSynchronize(st[0]));
  for (int i = 0; i < cs.R; ++i) CHK(hipStreamSynchronize(rs[i]));
  double mean = 0;
  for (double s : secs) mean += s / cs.K;
  printf("{\"K\": %d, \"n\": %d, \"sync_every\": %d, \"grid\": %d, \"mix\": %d, \"resident_streams\": %d, \"resident_wgs\": %d, \"fence_every\": %d, \"busy_us\": %d, \"spinner_nap\": %d, "
         "\"dispatches_per_s\": %.0f, \"us_per_dispatch_per_stream\": %.2f}\n",
         cs.K, cs.n, cs.sync_every, cs.grid, cs.mix, cs.R, cs.rg, cs.fence_every, cs.busy_us, cs.nap, cs.K * (double)cs.n / wall, mean / cs.n * 1e6);
  fflush(stdout);
  for (int i = 0; i < cs.K; ++i) {
    CHK(hipStreamDestroy(st[i]));
    CHK(hipFree(buf[i]));
    CHK(hipHostFree(pin[i]));
  }
  for (int i = 0; i < cs.R; ++i) CHK(hipStreamDestroy(rs[i]));
  CHK(hipFree(flag));
}

int main(int argc, char** argv) {
  CHK(hipSetDevice(0));
  const int n = 4000;
  if (argc > 1 && !strcmp(argv[1], "spinners")) {
    // what resident grids cost the other streams: 8 streams of tiny / 20-us kernels beside 7 spinning grids of 24 workgroups whose
    // pollers nap 1, 8, 32 or 127 x 64 clocks between polls
    for (int busy : {0, 20}) {
      run({8, 2000, 0, 1, 0, 0, 24, 0, 1, busy});
      for (int nap : {1, 8, 32, 127}) run({8, 2000, 0, 1, 0, 7, 24, 0, nap, busy});
      run({8, 2000, 0, 1, 0, 7, 1, 0, 1, busy});  // 7 spinners of ONE workgroup
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "overlap")) {
    // do kernels of different streams execute side by side?  K streams x kernels of a known length (1 / 16 workgroups)
    for (int us : {20, 100})
      for (int g : {1, 16})
        for (int K : {1, 4, 8, 16}) run({K, 1000, 0, g, 0, 0, 0, 0, 1, us});
    // memory-bound kernels: 16 workgroups x 256 KiB
    for (int K : {1, 4, 8, 16}) run({K, 1000, 0, 16, 0, 0, 0, 0, 1, -256});
    // a stream synchronisation after every kernel (host in the loop, as in a round-hand)
    for (int K : {1, 8, 16}) run({K, 1000, 1, 1, 0, 0, 0, 0, 1, 20});
    return 0;
  }
  // 1. the dispatch ceiling: K streams of one-workgroup kernels, no synchronisation
  for (int K : {1, 2, 4, 8, 16}) run({K, n, 0, 1, 0, 0, 0, 0});
  // 2. a stream synchronisation every 4 launches (a round-hand of the multi-kernel path)
  for (int K : {1, 4, 8, 16}) run({K, n, 4, 1, 0, 0, 0, 0});
  // 3. kernels + memsets + small D2H copies, as the prover's streams carry them
  for (int K : {1, 8, 16}) run({K, n, 0, 1, 1, 0, 0, 0});
  for (int K : {1, 8, 16}) run({K, n, 6, 1, 1, 0, 0, 0});
  // 4. wider kernels (64 workgroups)
  for (int K : {1, 8, 16}) run({K, n, 0, 64, 0, 0, 0, 0});
  // 5. 8 streams of tiny kernels beside R resident spinning grids of 24 workgroups (polling only / with a fence per 8 polls);
  //    R + 8 stays below the 16 hardware queues: a stream that shares a queue with a spinner waits for it to end
  for (int R : {0, 4, 7}) run({8, n, 0, 1, 0, R, 24, 0});
  for (int R : {4, 7}) run({8, n, 0, 1, 0, R, 24, 8});
  return 0;
}
