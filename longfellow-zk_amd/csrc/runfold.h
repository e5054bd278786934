// runfold.h -- GF(2^128): one term per lane, runs of equal keys contiguous in the block's lane order; XOR-fold every
// run inside the block and issue ONE pair of 64-bit atomic XORs per run and block.
//
// Why: a flatsha256 layer has runs of 10^5..10^6 terms (every gate that reads the constant-1 wire): folding inside the
// wave alone leaves one atomic pair per wave on ONE address (630 739 terms -> 9 855 pairs, ~250 us of serialised
// atomics in Quad::bind_g of the 32-block circuit); with the run fragments that touch a wave's edges merged across the
// waves of a 1024-thread block it is 616.  Exact and order-independent (addition is XOR).
#ifndef LFGPU_RUNFOLD_H_
#define LFGPU_RUNFOLD_H_
#include "fields.h"

// key == 0xffffffff marks an idle lane (its "run" is dropped).  Every thread of the block must call this.
template <int THREADS>
__device__ __forceinline__ void gf_run_fold_commit(u32 key, elt_t t, u64* __restrict__ dst /* 2 words per key */) {
  constexpr int NW = THREADS / 64;
  __shared__ u32 s_lk[NW], s_rk[NW], s_one[NW];
  __shared__ elt_t s_ls[NW], s_rs[NW];
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // run id inside the wave = number of run heads at or before the lane (monotone), so "same run id" implies every lane
  // in between has the same key and the suffix fold below is exact
  const u32 pkey = __shfl_up(key, 1, 64);
  const bool head = lane == 0 || pkey != key;
  const u64 hmask = __ballot(head);
  const u32 rid = (u32)__popcll(hmask & ((2ull << lane) - 1));
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u64 olo = __shfl_down(t.lo, off, 64), ohi = __shfl_down(t.hi, off, 64);
    const u32 orid = __shfl_down(rid, off, 64);
    if (lane + off < 64 && orid == rid) {
      t.lo ^= olo;
      t.hi ^= ohi;
    }
  }
  // the head lane of each fragment now holds the fragment's sum.  The fragment at lane 0 may continue the previous
  // wave's last one, the fragment that reaches lane 63 may continue in the next wave: those two go through LDS, the
  // others are complete runs (or complete ends of runs) and are committed at once
  const u32 hr = 63u - (u32)__clzll(hmask);  // head lane of the fragment that reaches lane 63
  if (head && lane != 0 && lane != hr && key != 0xffffffffu) {
    atomicXor(&dst[2 * (size_t)key], t.lo);
    atomicXor(&dst[2 * (size_t)key + 1], t.hi);
  }
  if (lane == 0) {
    s_lk[wave] = key;
    s_ls[wave] = t;
    s_one[wave] = hr == 0 ? 1u : 0u;
  }
  if (lane == hr) {
    s_rk[wave] = key;
    s_rs[wave] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 ck = 0xffffffffu;
    elt_t cs = elt_zero();
    auto flush = [&]() {
      if (ck != 0xffffffffu) {
        atomicXor(&dst[2 * (size_t)ck], cs.lo);
        atomicXor(&dst[2 * (size_t)ck + 1], cs.hi);
      }
    };
    for (int w = 0; w < NW; ++w) {
      const elt_t ls = s_ls[w];
      if (s_lk[w] == ck) {
        cs.lo ^= ls.lo;
        cs.hi ^= ls.hi;
      } else {
        flush();
        ck = s_lk[w];
        cs = ls;
      }
      if (!s_one[w]) {  // the left fragment ended inside the wave: the open one is now the wave's last
        flush();
        ck = s_rk[w];
        cs = s_rs[w];
      }
    }
    flush();
  }
  __syncthreads();  // the LDS words are free again: the caller may be in a loop
}
#endif
