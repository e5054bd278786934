// zk256.hip -- ZkProver<Fp256Base, .> on the device: the P-256 base-field half of BASELINE config 5 (the mdoc signature
// circuit: 21 layers, 4.8e5 terms, 32-byte elements, lib/circuits/mdoc/mdoc_zk.cc:70-76,485-522).
//
// Same algorithms as the 16-byte fields, restated over elt32_t (fp256.h) with the straightforward kernel structure -- the
// circuit is small (layers of 2^9 .. 2^16 wires), so every round-hand is a handful of short launches and ONE read-back:
//   eval_circuit            ProverLayers::eval_quad            lib/sumcheck/prover_layers.h:278-305
//   bind_g                  Quad::bind_g + Eqs::raw_eq2        lib/sumcheck/quad.h:152-185, lib/arrays/eqs.h:46-80
//   round body              ProverLayers::layer / evaluations  lib/sumcheck/prover_layers.h:230-263,357-402
//   binds                   Dense::bind, HQuad::bind_h         lib/arrays/dense.h:70-87, lib/sumcheck/hquad.h:90-123
//   Ligero                  LigeroProver::commit / prove       lib/ligero/ligero_prover.h:58-146,171-351
//   driver                  ZkProver::commit / prove, ZkCommon::verifier_constraints, ZkProof::write
//                           lib/zk/zk_prover.h:72-188, lib/zk/zk_common.h:49-136,406-439, lib/zk/zk_proof.h:90-185
// Sums over many terms (run sums of bind_g, the QW scatter) add the 32-bit limbs of canonical residues into 64-bit integer
// accumulators with atomics and reduce once (fp256_reduce_limbs): exact and independent of arrival order, as for Fp128.
// Row extension and column hashing are csrc/p256.hip.  The verifier (ZkVerifier::verify, lib/zk/zk_verifier.h:68-94) is at
// the end of the file.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>

#include "fp256.h"
#include "fs_crypto.h"
#include "zkint.h"

typedef elt32_t E;
#define Z_THREADS 256

namespace {
// ------------------------------------------------------------------ kernels
__device__ __forceinline__ void limbs_atomic_add(u64* a, const E& v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    atomicAdd(&a[2 * k], (u64)(u32)v.l[k]);
    atomicAdd(&a[2 * k + 1], v.l[k] >> 32);
  }
}
// One term per lane; runs of equal keys among consecutive lanes (canonical order keeps equal hand pairs, and mostly also the
// hot target -- the constant wire 0 -- adjacent) are summed inside the wave first, so that a run costs 8 atomics per wave
// instead of 8 per term: the signature circuit has runs of 3e4 terms and one wire that 1e5 terms of a layer read.
// key 0xffffffff marks an idle lane.  Every lane of the wave must call this.
__device__ __forceinline__ void run_fold_commit256(u32 key, E t, u64* __restrict__ acc) {
  const u32 lane = threadIdx.x & 63;
  const u32 pkey = __shfl_up(key, 1, 64);
  const bool head = lane == 0 || pkey != key || key == 0xffffffffu;  // idle lanes never form a run: a partly filled wave skips the fold
  const u64 hmask = __ballot(head);
  if (hmask != ~0ull) {  // wave-uniform: some neighbours share a key
    const u32 rid = (u32)__popcll(hmask & ((2ull << lane) - 1));  // monotone, so equal ids = one contiguous run
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      E o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o.l[k] = __shfl_down(t.l[k], off, 64);
      const u32 orid = __shfl_down(rid, off, 64);
      if (lane + off < 64 && orid == rid) t = fp256_add(t, o);
    }
  }
  if (head && key != 0xffffffffu) limbs_atomic_add(acc + 8 * (size_t)key, t);
}
__device__ __forceinline__ E take_acc256(u64* __restrict__ acc, size_t i, const E& rsq) {
  u64 a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = acc[8 * i + k];
    acc[8 * i + k] = 0;
  }
  return fp256_reduce_limbs(a, rsq);
}
// out[i] = the field element of accumulator i; the accumulators are left zeroed (they serve the next sum)
__global__ __launch_bounds__(Z_THREADS) void limb_normalize256_kernel(size_t n, u64* __restrict__ acc, E rsq, E* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * Z_THREADS + threadIdx.x;
  if (i >= n) return;
  st32(&out[i], take_acc256(acc, i, rsq));
}

// K11: V[g] = sum_{terms of g} kvec[vi] * W[h1] * W[h0]; assert-zero terms must vanish.  One term per lane over the
// by-gate order (a gate's terms are contiguous: one gate of the signature circuit has 769), summed as above.
__global__ __launch_bounds__(Z_THREADS) void eval_quad256_kernel(size_t n, const corner4* __restrict__ terms, const E* __restrict__ kvec,
                                                                 const E* __restrict__ W, u64* __restrict__ acc, int* __restrict__ fail) {
  const size_t i = (size_t)blockIdx.x * Z_THREADS + threadIdx.x;
  u32 key = 0xffffffffu;
  E t = e32_zero();
  if (i < n) {
    const corner4 cr = terms[i];
    key = cr.g;
    const E v = ld32(&kvec[cr.vi]);
    const E p = fp256_mul(ld32(&W[cr.h1]), ld32(&W[cr.h0]));
    if (e32_is_zero(v)) {
      if (!e32_is_zero(p)) atomicOr(fail, 1);
    } else {
      t = fp256_mul(v, p);
    }
  }
  run_fold_commit256(key, t, acc);
}

// eq[i] = EQ(G0, i) + alpha EQ(G1, i), EQ(G, i) = prod_l (bit_l(i) ? G[l] : 1 - G[l]); G = G0 | G1 | 1-G0 | 1-G1
__global__ __launch_bounds__(Z_THREADS) void raw_eq2_256_kernel(u32 logn, u32 n, const E* __restrict__ G, E alpha, E one, E* __restrict__ eq) {
  const u32 i = blockIdx.x * Z_THREADS + threadIdx.x;
  if (i >= n) return;
  E e0 = one, e1 = alpha;
  for (u32 l = 0; l < logn; ++l) {
    const u32 bit = (i >> l) & 1;
    e0 = fp256_mul(e0, ld32(&G[(bit ? 0 : 2 * logn) + l]));
    e1 = fp256_mul(e1, ld32(&G[(bit ? logn : 3 * logn) + l]));
  }
  st32(&eq[i], fp256_add(e0, e1));
}

// The same vector with 2 products per entry instead of 2 logn: EQ(G, i) = LO[i mod 2^lb] * HI[i >> lb].  eq_tables256_kernel
// builds the four factor tables LO0 | HI0 | LO1 | HI1 (alpha folded into HI1), one thread per entry (<= 2^ceil(logn / 2)
// entries each), raw_eq2_split256_kernel combines them.  Exact field arithmetic: the association does not matter.
__global__ __launch_bounds__(Z_THREADS) void eq_tables256_kernel(u32 logn, u32 lb, const E* __restrict__ G, E alpha, E one, E* __restrict__ tab) {
  const u32 hb = logn - lb, nlo = 1u << lb, nhi = 1u << hb;
  const u32 t = blockIdx.x * Z_THREADS + threadIdx.x;
  if (t >= 2 * (nlo + nhi)) return;
  const u32 which = t < nlo ? 0 : t < nlo + nhi ? 1 : t < 2 * nlo + nhi ? 2 : 3;  // LO0, HI0, LO1, HI1
  const u32 j = which == 0 ? t : which == 1 ? t - nlo : which == 2 ? t - nlo - nhi : t - 2 * nlo - nhi;
  const u32 bits = (which & 1) ? hb : lb, shift = (which & 1) ? lb : 0, g = which >> 1;  // g: 0 -> G0, 1 -> G1
  E e = which == 3 ? alpha : one;
  for (u32 l = 0; l < bits; ++l) {
    const u32 bit = (j >> l) & 1;
    e = fp256_mul(e, ld32(&G[(bit ? g * logn : (2 + g) * logn) + shift + l]));
  }
  st32(&tab[t], e);
}
__global__ __launch_bounds__(Z_THREADS) void raw_eq2_split256_kernel(u32 logn, u32 lb, u32 n, const E* __restrict__ tab, E* __restrict__ eq) {
  const u32 i = blockIdx.x * Z_THREADS + threadIdx.x;
  if (i >= n) return;
  const u32 nlo = 1u << lb, nhi = 1u << (logn - lb);
  const u32 lo = i & (nlo - 1), hi = i >> lb;
  const E e0 = fp256_mul(ld32(&tab[lo]), ld32(&tab[nlo + hi]));
  const E e1 = fp256_mul(ld32(&tab[nlo + nhi + lo]), ld32(&tab[2 * nlo + nhi + hi]));
  st32(&eq[i], fp256_add(e0, e1));
}

// Quad::bind_g: every term computes prep_v(v, beta) * eq[g]; the terms of a run (equal hand pair, contiguous in canonical
// order) add into the run's limb accumulators
__device__ __forceinline__ bool is_head256(const corner4* t, size_t i) { return i == 0 || t[i].h0 != t[i - 1].h0 || t[i].h1 != t[i - 1].h1; }
__global__ __launch_bounds__(BG_THREADS) void bindg_emit256_kernel(size_t n, const corner4* __restrict__ t, const E* __restrict__ kvec,
                                                                   const E* __restrict__ eq, E beta, const u32* __restrict__ block_off,
                                                                   uint2* __restrict__ hc_out, u64* __restrict__ acc) {
  __shared__ u32 wave_off[BG_THREADS / 64];
  const size_t i = (size_t)blockIdx.x * BG_THREADS + threadIdx.x;
  const bool valid = i < n;
  const bool head = valid && is_head256(t, i);
  const u64 mask = __ballot(head);
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_off[wave] = (u32)__popcll(mask);
  __syncthreads();
  u32 ri = block_off[blockIdx.x];
  for (u32 w = 0; w < wave; ++w) ri += wave_off[w];
  ri += (u32)__popcll(mask & ((2ull << lane) - 1)) - 1;
  E pv = e32_zero();
  if (valid) {
    const corner4 c0 = t[i];
    E v = ld32(&kvec[c0.vi]);
    if (e32_is_zero(v)) v = beta;
    pv = fp256_mul(v, ld32(&eq[c0.g]));
    if (head) hc_out[ri] = make_uint2(c0.h0, c0.h1);
  }
  run_fold_commit256(valid ? ri : 0xffffffffu, pv, acc);
}

// QW[h[hand]] += v * Wother[h[1-hand]]
__global__ __launch_bounds__(Z_THREADS) void qw_scatter256_kernel(size_t n, const uint2* __restrict__ hc, const E* __restrict__ vc, int hand,
                                                                  const E* __restrict__ Wo, u64* __restrict__ acc) {
  const size_t i = (size_t)blockIdx.x * Z_THREADS + threadIdx.x;
  u32 key = 0xffffffffu;
  E t = e32_zero();
  if (i < n) {
    const uint2 h = hc[i];
    key = hand ? h.y : h.x;
    t = fp256_mul(ld32(&vc[i]), ld32(&Wo[hand ? h.x : h.y]));
  }
  run_fold_commit256(key, t, acc);
}

__device__ __forceinline__ E block_sum256(E v, E* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (u32 s = Z_THREADS / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] = fp256_add(sh[threadIdx.x], sh[threadIdx.x + s]);
    __syncthreads();
  }
  const E r = sh[0];
  __syncthreads();
  return r;
}
// The two sums of a round (ProverLayers::evaluations :357-402), a0 = sum QW[2i] W[2i] and
// a2 = sum (QW[2i+1] - QW[2i]) (W[2i+1] - W[2i]), straight from the scatter's limb accumulators: every accumulator is
// reduced to its field element here and zeroed for the next round-hand; the block sums go, again as limbs, into 16 words
// (out[0..8) = a0, out[8..16) = a2) that the host reduces.
// post != nullptr: the block that finishes last copies the 16 words to coherent pinned host memory (post[0..16)), clears
// them and publishes `seq` at post[16] -- the host spins on that word instead of a copy + stream synchronisation.
__global__ __launch_bounds__(Z_THREADS) void partials256_kernel(size_t n, u64* __restrict__ acc, E rsq, const E* __restrict__ W, u64* __restrict__ out,
                                                                u32* __restrict__ done, volatile u64* __restrict__ post, u64 seq) {
  __shared__ E sh[Z_THREADS];
  __shared__ u32 s_last;
  const size_t nodd = n / 2;
  E a0 = e32_zero(), a2 = e32_zero();
  for (size_t i = (size_t)blockIdx.x * Z_THREADS + threadIdx.x; i < nodd; i += (size_t)gridDim.x * Z_THREADS) {
    const E q0 = take_acc256(acc, 2 * i, rsq), q1 = take_acc256(acc, 2 * i + 1, rsq), w0 = ld32(&W[2 * i]), w1 = ld32(&W[2 * i + 1]);
    a0 = fp256_add(a0, fp256_mul(q0, w0));
    a2 = fp256_add(a2, fp256_mul(fp256_sub(q1, q0), fp256_sub(w1, w0)));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && 2 * nodd < n) {  // odd tail (:381-388)
    const E t = fp256_mul(take_acc256(acc, 2 * nodd, rsq), ld32(&W[2 * nodd]));
    a0 = fp256_add(a0, t);
    a2 = fp256_add(a2, t);
  }
  a0 = block_sum256(a0, sh);
  a2 = block_sum256(a2, sh);
  if (threadIdx.x == 0) {
    limbs_atomic_add(out, a0);
    limbs_atomic_add(out + 8, a2);
    s_last = 0;
    if (post) {
      __threadfence();
      s_last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
  }
  __syncthreads();
  if (s_last && threadIdx.x < 64) {  // every block's adds are visible (each fenced before its increment); 16 lanes of one wave read,
    if (threadIdx.x < 16) {          // clear and copy one word each (the round trips overlap), lane 0 then publishes
      post[threadIdx.x] = __hip_atomic_exchange(&out[threadIdx.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
    }
    if (threadIdx.x == 0) {
      *done = 0;
      __hip_atomic_store((u64*)&post[16], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// out[i] = in[2i] + r (in[2i+1] - in[2i]); tail: in (1 - r)   (dense.h:70-87, affine.h:26-52)
__device__ __forceinline__ void dense_bind256(size_t i, size_t n0, const E& r, const E* __restrict__ in, E* __restrict__ out) {
  if (i >= (n0 + 1) / 2) return;
  const E f0 = ld32(&in[2 * i]);
  E v;
  if (2 * i + 1 < n0) v = fp256_add(f0, fp256_mul(fp256_sub(ld32(&in[2 * i + 1]), f0), r));
  else v = fp256_sub(f0, fp256_mul(f0, r));
  st32(&out[i], v);
}

// HQuad::bind_h as an order-preserving compaction (see sumcheck.hip): a term is the second half of a merged pair iff its
// predecessor has the same other-hand corner and the even index h - 1
__device__ __forceinline__ bool is_second256(const uint2* hc, size_t i, int hand) {
  if (i == 0) return false;
  const uint2 a = hc[i - 1], b = hc[i];
  const u32 ah = hand ? a.y : a.x, ao = hand ? a.x : a.y, bh = hand ? b.y : b.x, bo = hand ? b.x : b.y;
  return ao == bo && (ah >> 1) == (bh >> 1) && bh == ah + 1;
}
__global__ __launch_bounds__(Z_THREADS) void hquad_count256_kernel(size_t n, const uint2* __restrict__ hc, int hand, u32* __restrict__ block_counts) {
  __shared__ u32 cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * Z_THREADS + threadIdx.x;
  const bool head = i < n && !is_second256(hc, i, hand);
  const u64 mask = __ballot(head);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, (u32)__popcll(mask));
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = cnt;
}
__global__ __launch_bounds__(1024) void scan256_kernel(u32 nblocks, u32* __restrict__ block_counts, u32* __restrict__ total) {
  __shared__ u32 sh[1024];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (u32 base = 0; base < nblocks; base += 1024) {
    const u32 i = base + threadIdx.x;
    const u32 v = i < nblocks ? block_counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
      const u32 t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const u32 incl = sh[threadIdx.x];
    if (i < nblocks) block_counts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
// Dense::bind of the hand that just received its challenge (blocks [0, nbd)) and HQuad::bind_h (the blocks after them) in
// one launch
__global__ __launch_bounds__(Z_THREADS) void bind256_kernel(u32 nbd, size_t n0, const E* __restrict__ win, E* __restrict__ wout, size_t n,
                                                            const uint2* __restrict__ hc, const E* __restrict__ vc, E r, int hand,
                                                            const u32* __restrict__ block_off, uint2* __restrict__ hc_out, E* __restrict__ vc_out) {
  __shared__ u32 wave_off[Z_THREADS / 64];
  if (blockIdx.x < nbd) {
    dense_bind256((size_t)blockIdx.x * Z_THREADS + threadIdx.x, n0, r, win, wout);
    return;
  }
  const u32 blk = blockIdx.x - nbd;
  const size_t i = (size_t)blk * Z_THREADS + threadIdx.x;
  const bool head = i < n && !is_second256(hc, i, hand);
  const u64 mask = __ballot(head);
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_off[wave] = (u32)__popcll(mask);
  __syncthreads();
  u32 off = block_off[blk];
  for (u32 w = 0; w < wave; ++w) off += wave_off[w];
  off += (u32)__popcll(mask & ((1ull << lane) - 1));
  if (!head) return;
  uint2 h = hc[i];
  const u32 hh = hand ? h.y : h.x;
  const E v0 = ld32(&vc[i]);
  E v;
  if (i + 1 < n && is_second256(hc, i + 1, hand)) v = fp256_add(v0, fp256_mul(fp256_sub(ld32(&vc[i + 1]), v0), r));  // affine_interpolation
  else if ((hh & 1) == 0) v = fp256_sub(v0, fp256_mul(v0, r));                                                         // ..._nz_z
  else v = fp256_mul(v0, r);                                                                                            // ..._z_nz
  if (hand) h.y = hh >> 1;
  else h.x = hh >> 1;
  hc_out[off] = h;
  st32(&vc_out[off], v);
}

// ---- the rest of a layer in ONE launch of co-resident workgroups (what sc_grid_layer_kernel is for the 16-byte fields,
// sumcheck.hip: same phases, same bounded waits, without its LDS tail and wave-split variants).  Per round-hand:
//   barrier | sums a0, a2 -> per-workgroup slots -> the workgroup that arrives last folds them and posts to the host |
//   layout of HQuad::bind_h (which entries merge, where each result goes: nothing of it needs the challenge, so it hides behind
//   the host's turn) | challenge (eight tagged words) | Dense::bind + HQUAD values + the NEXT evaluation's accumulators from
//   those values (double-buffered) | workgroups beyond ceil(largest array / per_wg) leave.
// Launched as an ordinary kernel of <= G256_WGS <= #CU workgroups (the host checks co-residency); every wait is bounded by an
// abort flag and a wall-clock timeout, so all waves always leave.
#define G256_THREADS 512
#define G256_WGS 128
#define G256_MAX (128u * 1024u)  // largest HQUAD / hand array the grid takes
struct Grid256Sync {  // device memory, zeroed before every launch
  u32 count, gen, abort, arrive;
  u64 chal[8];  // the challenge as the host's eight tagged words
  u64 pad_[6];
  u32 l1_bar[8 * 16];  // first-level arrival counters of the barrier, one cache line each
  u32 l1_arr[8 * 16];  // ... of the sums' arrival ticket
  u64 slots[8 * G256_WGS];  // per workgroup {a0, a2}
};
struct Grid256 {
  uint2* hcA;  // current HQUAD
  E* vcA;
  uint2* hcB;  // the other half of the ping-pong
  E* vcB;
  u32 nh;
  const E* W[2];  // hand arrays at entry
  u32 nW[2];
  E* Wb[2][2];  // bind destinations per hand (ping-pong)
  u64* QW;      // limb accumulators of the evaluation being summed, 8 words per target
  u64* QW2;     // of the next one
  u32 rh0, rh1;  // round-hands [rh0, rh1), rh1 = 2 logw
  u64 seq0, timeout_ticks;
  volatile u64* post;
  const volatile u64* cmd;
  Grid256Sync* gs;
  u32* counts;  // one word per workgroup
  u32* src;     // one word per HQUAD entry: where each bound entry comes from (+ its merge kind in the top bits)
  u32 per_wg;
  u32 wave_tail;  // 1: once everything fits 64 entries, ONE wave finishes the layer
  u32 test_drop;  // test only: the last workgroup leaves at once, as if it had never been placed
  u64 place_ticks;  // how long the FIRST barrier waits: the check that all workgroups were placed together
  E rsq;
};
// two-level arrival ticket (see sc_arrive in sumcheck.hip): true for the one workgroup that arrives last
__device__ __forceinline__ bool g256_arrive(u32* lvl1, u32* lvl2, u32 G, u32 g) {
  const u32 grp = g & 7, ngrp = G < 8 ? G : 8, members = (G - grp + 7) >> 3;
  u32* c1 = lvl1 + grp * 16;
  if (__hip_atomic_fetch_add(c1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
  __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_fetch_add(lvl2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != ngrp - 1) return false;
  __hip_atomic_store(lvl2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
// barrier among the first G workgroups, waiting at most limit_ticks; 0 = passed, 1 = aborted by another workgroup, 2 = this
// workgroup's wait ran out and it raised the abort flag first (every caller returns on non-zero)
__device__ __forceinline__ int g256_barrier_ex(Grid256Sync* gs, u32 G, u32& gen, u64 limit_ticks) {
  __shared__ u32 s_abort;
  if (G == 1) {
    __threadfence_block();
    __syncthreads();
    return 0;
  }
  // every storing wave: its stores are acknowledged before thread 0 releases for them (as in sc_grid_barrier_ex, sumcheck.hip)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 ab = 0;
    const u64 t0 = wall_clock64();
    if (g256_arrive(gs->l1_bar, &gs->count, G, blockIdx.x)) {
      // a workgroup that was placed only after the others gave up arrives last: it must not open the barrier for itself
      if (__hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) ab = 1;
      else __hip_atomic_fetch_add(&gs->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&gs->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        if (__hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          ab = 1;
          break;
        }
        if (wall_clock64() - t0 > limit_ticks) {  // never reached in a healthy run: no wait is unbounded
          ab = __hip_atomic_exchange(&gs->abort, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0 ? 2u : 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    s_abort = ab;
  }
  ++gen;
  __syncthreads();
  return (int)s_abort;
}
__device__ __forceinline__ bool g256_barrier(Grid256Sync* gs, u32 G, u32& gen, u64 timeout_ticks) {
  return g256_barrier_ex(gs, G, gen, 2 * timeout_ticks) == 0;
}

__global__ __launch_bounds__(G256_THREADS) void grid256_layer_kernel(Grid256 a) {
  __shared__ E s_red[2][G256_THREADS / 64];
  __shared__ u32 s_wave[G256_THREADS / 64];
  __shared__ u32 s_carry, s_off, s_tot, s_last;
  __shared__ u64 s_cmd[5];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 g = blockIdx.x;
  u32 G = gridDim.x;  // active workgroups: shrinks with the data
  Grid256Sync* gs = a.gs;
  const uint2* hc = a.hcA;
  const E* vc = a.vcA;
  uint2* hc_o = a.hcB;
  E* vc_o = a.vcB;
  u32 nh = a.nh;
  const E* W[2] = {a.W[0], a.W[1]};
  u32 nW[2] = {a.nW[0], a.nW[1]};
  u32 wsel[2] = {a.W[0] == a.Wb[0][0] ? 1u : 0u, a.W[1] == a.Wb[1][0] ? 1u : 0u};
  u64* QW = a.QW;
  u64* QWn = a.QW2;
  u32 gen = 0;
  u64 seq = a.seq0;
  bool wave_mode = false;
  __shared__ uint2 t_hc[2][64];  // the state of the single-wave tail (<= 64 entries per array)
  __shared__ E t_vc[2][64];
  __shared__ E t_w[2][128];      // per hand: current (64) + two bind destinations (32 each)
  __shared__ u64 t_qw[2][8 * 64];
  __shared__ u32 t_src[64];
  E* Wd[2][2] = {{a.Wb[0][0], a.Wb[0][1]}, {a.Wb[1][0], a.Wb[1][1]}};
  u32* srcp = a.src;
  const E rsq = a.rsq;
  {
    const u32 GT = G * G256_THREADS, gtid = (wave * G + g) * 64 + lane;  // 64-entry chunks dealt round-robin over the workgroups
    const u32 h0 = a.rh0 & 1;
    for (u32 i = gtid; i < 8 * nW[h0]; i += GT) QW[i] = 0;
    for (u32 i = gtid; i < 8 * nW[1 - h0]; i += GT) QWn[i] = 0;
  }
  // placement check (see sc_grid_layer_kernel, sumcheck.hip): only scratch has been written so far; when not every workgroup
  // shows up, the one whose wait runs out first reports status 2 and the host continues with per-launch kernels
  if (a.test_drop && G > 1 && g == G - 1) return;
  {
    const int br = g256_barrier_ex(gs, G, gen, a.place_ticks);
    if (br) {
      if (br == 2 && tid == 0) {
        a.post[9] = 2;
        __threadfence_system();
        __hip_atomic_store((u64*)&a.post[16], a.seq0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
  {  // first evaluation of the hand-off: QW[h[hand]] += v * Wother[h[1-hand]] over the whole HQUAD
    const int hand = (int)(a.rh0 & 1);
    const u32 GT = G * G256_THREADS;
    const E* Wo = W[1 - hand];
    for (u32 base = (wave * G + g) * 64; base < (nh + 63) / 64 * 64; base += GT) {  // whole waves: the fold shuffles
      const u32 i = base + lane;
      u32 key = 0xffffffffu;
      E t = e32_zero();
      if (i < nh) {
        const uint2 h = hc[i];
        key = hand ? h.y : h.x;
        t = fp256_mul(ld32(&vc[i]), ld32(&Wo[hand ? h.x : h.y]));
      }
      run_fold_commit256(key, t, QW);
    }
  }
  for (u32 rh = a.rh0; rh < a.rh1; ++rh, ++seq) {
    const int hand = (int)(rh & 1);
    if (!g256_barrier(gs, G, gen, a.timeout_ticks)) return;  // the accumulators of this evaluation are complete
    {  // shrink: one workgroup per per_wg entries of the largest array; the others are done
      u32 big = nh > nW[0] ? nh : nW[0];
      big = big > nW[1] ? big : nW[1];
      u32 want = (big + a.per_wg - 1) / a.per_wg;
      want = want ? want : 1;
      if (want < G) G = want;
      if (g >= G) return;
    }
    // ---- the last rounds of a layer: at most 64 entries in every array.  Wave 0 finishes the layer alone (see the same
    // block of sc_grid_layer_kernel, sumcheck.hip): both sums in ONE product (lanes 0-31 the a0 terms, lanes 32-63 the a2
    // terms), both binds in one product when they fit the wave together, the layout by ballot, the sums by shuffles; no
    // workgroup barrier is left -- the other waves have ended.  Same field operations on the same operands as below.
    if (G == 1 && a.wave_tail && !wave_mode && nh <= 64 && nW[0] <= 64 && nW[1] <= 64) {
      if (wave != 0) return;  // the barrier at the top of the loop was their last one: everything they wrote is visible
      wave_mode = true;
      // the whole state moves into LDS (21 KiB): every phase is then a product plus an LDS round trip instead of a product
      // plus one or two trips to L2
      if (lane < nh) {
        t_hc[0][lane] = hc[lane];
        st32(&t_vc[0][lane], ld32(&vc[lane]));
      }
      for (int h = 0; h < 2; ++h)
        if (lane < nW[h]) st32(&t_w[h][lane], ld32(&W[h][lane]));
      for (u32 i = lane; i < 8 * 64; i += 64) {
        t_qw[0][i] = i < 8 * nW[hand] ? QW[i] : 0;  // the sums of this evaluation
        t_qw[1][i] = 0;
      }
      hc = t_hc[0]; hc_o = t_hc[1]; vc = t_vc[0]; vc_o = t_vc[1];
      for (int h = 0; h < 2; ++h) {
        W[h] = t_w[h];
        Wd[h][0] = t_w[h] + 64;
        Wd[h][1] = t_w[h] + 96;
        wsel[h] = 0;
      }
      QW = t_qw[0];
      QWn = t_qw[1];
      srcp = t_src;
      __threadfence_block();
      __syncthreads();
    }
    if (wave_mode) {
      const u32 nq = nW[hand], nodd = nq / 2;
      const E* Wh = W[hand];
      auto qw_at = [&](u32 j) -> E {
        u64 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = QW[8 * (size_t)j + k];
        return fp256_reduce_limbs(q, rsq);
      };
      E x = e32_zero(), y = e32_zero();
      {
        const u32 i = lane & 31;
        if (i < nodd) {
          const E q0 = qw_at(2 * i), w0 = ld32(&Wh[2 * i]);
          if (lane < 32) {
            x = q0;
            y = w0;
          } else {
            x = fp256_sub(qw_at(2 * i + 1), q0);
            y = fp256_sub(ld32(&Wh[2 * i + 1]), w0);
          }
        } else if (i == nodd && 2 * nodd < nq) {  // odd tail (prover_layers.h:381-388): in both sums
          x = qw_at(2 * nodd);
          y = ld32(&Wh[2 * nodd]);
        }
      }
      E t = fp256_mul(x, y);
      for (int off = 16; off > 0; off >>= 1) {  // sums inside each half of the wave
        E o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o.l[k] = __shfl_down(t.l[k], off, 32);
        t = fp256_add(t, o);
      }
      E a2;
#pragma unroll
      for (int k = 0; k < 4; ++k) a2.l[k] = __shfl(t.l[k], 32, 64);
      if (lane == 0) {
        u64* po = (u64*)a.post;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __hip_atomic_store(&po[k], t.l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&po[4 + k], a2.l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&po[8], (u64)nh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[9], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lf_wait_stores_before_publish();  // payload acknowledged before the sequence word leaves
        __hip_atomic_store(&po[16], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // layout of HQuad::bind_h (no challenge needed): lane i looks at entry i
      const bool head = lane < nh && !is_second256(hc, lane, hand);
      const u64 hmask = __ballot(head);
      const u32 new_nh = (u32)__popcll(hmask);
      if (head) {
        const u32 off = (u32)__popcll(hmask & ((1ull << lane) - 1));
        uint2 h = hc[lane];
        const u32 hh = hand ? h.y : h.x;
        const u32 kind = (lane + 1 < nh && is_second256(hc, lane + 1, hand)) ? 0u : ((hh & 1) == 0 ? 1u : 2u);
        if (hand) h.y = hh >> 1;
        else h.x = hh >> 1;
        hc_o[off] = h;
        srcp[off] = lane | (kind << 30);
      }
      // the challenge: eight tagged words (see below)
      u64 w = 0;
      {
        const u64 t0 = wall_clock64();
        const u64 tag = seq & 0xffffffffull;
        for (;;) {
          if (lane < 8) w = __hip_atomic_load((const u64*)&a.cmd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (__all(lane >= 8 || (w >> 32) == tag)) break;
          int stop = 0;
          if (lane == 0 && wall_clock64() - t0 > a.timeout_ticks) {  // the host went away: report and leave
            a.post[9] = 1;
            __threadfence_system();
            __hip_atomic_store((u64*)&a.post[16], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            stop = 1;
          }
          if (__shfl(stop, 0, 64)) return;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      E r;
      {
        const u64 lo32 = w & 0xffffffffull;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.l[k] = __shfl(lo32, 2 * k, 64) | (__shfl(lo32, 2 * k + 1, 64) << 32);
      }
      const E* Wold = W[hand];
      const u32 n0 = nW[hand], nout = (n0 + 1) / 2;
      for (u32 i = lane; i < 8 * nout; i += 64) QW[i] = 0;  // next written two round-hands from now, for this hand again
      __threadfence_block();
      __syncthreads();  // one wave: a wait for the stores above (layout, clear)
      // Dense::bind of W[hand] and the HQuad::bind_h values: out = base + d * r for both (hquad.h:94-118)
      E* const Wout = Wd[hand][wsel[hand]];
      const bool merged = new_nh + nout <= 64;  // both binds in one pass: HQUAD values on the low lanes, the hand array on the high ones
      E vbound = e32_zero();
      for (int pass = 0; pass < (merged ? 1 : 2); ++pass) {
        E base = e32_zero(), d = e32_zero();
        int job = 0;  // 1: HQUAD value `lane`, 2: hand entry j
        u32 j = 0;
        if (pass == 0 && lane < new_nh) {
          job = 1;
          const u32 sidx = srcp[lane], i = sidx & 0x3fffffffu, kind = sidx >> 30;
          const E v0 = ld32(&vc[i]);
          if (kind == 0) {
            base = v0;
            d = fp256_sub(ld32(&vc[i + 1]), v0);
          } else if (kind == 1) {
            base = v0;
            d = fp256_sub(e32_zero(), v0);
          } else {
            d = v0;
          }
        } else if (merged ? lane >= 64 - nout : (pass == 1 && lane < nout)) {
          job = 2;
          j = merged ? lane - (64 - nout) : lane;
          const E f0 = ld32(&Wold[2 * j]);
          base = f0;
          d = 2 * j + 1 < n0 ? fp256_sub(ld32(&Wold[2 * j + 1]), f0) : fp256_sub(e32_zero(), f0);
        }
        const E out = fp256_add(base, fp256_mul(d, r));
        if (job == 1) {
          vbound = out;
          st32(&vc_o[lane], out);
        } else if (job == 2) {
          st32(&Wout[j], out);
        }
      }
      __threadfence_block();
      __syncthreads();
      if (rh + 1 < a.rh1) {  // the next evaluation's accumulators (for the other hand)
        u32 key = 0xffffffffu;
        E tt = e32_zero();
        if (lane < new_nh) {
          const uint2 h = hc_o[lane];
          key = hand ? h.x : h.y;
          tt = fp256_mul(vbound, ld32(&Wout[hand ? h.y : h.x]));
        }
        run_fold_commit256(key, tt, QWn);
      }
      W[hand] = Wout;
      nW[hand] = nout;
      wsel[hand] ^= 1;
      {
        nh = new_nh;
        uint2* th = const_cast<uint2*>(hc);
        E* tv = const_cast<E*>(vc);
        hc = hc_o;
        vc = vc_o;
        hc_o = th;
        vc_o = tv;
        u64* tq = QW;
        QW = QWn;
        QWn = tq;
      }
      continue;
    }
    const u32 GT = G * G256_THREADS, gtid = (wave * G + g) * 64 + lane;
    // ---- ProverLayers::evaluations: a0, a2 from the accumulators
    {
      const u32 nq = nW[hand], nodd = nq / 2;
      const E* Wh = W[hand];
      auto qw_at = [&](u32 j) -> E {
        u64 q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) q[k] = QW[8 * (size_t)j + k];
        return fp256_reduce_limbs(q, rsq);
      };
      E a0 = e32_zero(), a2 = e32_zero();
      if (G == 1) {  // one workgroup left: the two products of a pair on different waves (even waves sum a0, odd waves a2)
        const u32 half = G256_THREADS / 2, ht = (wave >> 1) * 64 + lane;
        const bool second = (wave & 1) != 0;
        for (u32 i = ht; i < nodd; i += half) {
          const E q0 = qw_at(2 * i), w0 = ld32(&Wh[2 * i]);
          if (!second) a0 = fp256_add(a0, fp256_mul(q0, w0));
          else a2 = fp256_add(a2, fp256_mul(fp256_sub(qw_at(2 * i + 1), q0), fp256_sub(ld32(&Wh[2 * i + 1]), w0)));
        }
        if (ht == 0 && 2 * nodd < nq) {  // odd tail (prover_layers.h:381-388): in both sums
          const E t = fp256_mul(qw_at(2 * nodd), ld32(&Wh[2 * nodd]));
          if (!second) a0 = fp256_add(a0, t);
          else a2 = fp256_add(a2, t);
        }
      } else {
        for (u32 i = gtid; i < nodd; i += GT) {
          const E q0 = qw_at(2 * i), q1 = qw_at(2 * i + 1), w0 = ld32(&Wh[2 * i]), w1 = ld32(&Wh[2 * i + 1]);
          a0 = fp256_add(a0, fp256_mul(q0, w0));
          a2 = fp256_add(a2, fp256_mul(fp256_sub(q1, q0), fp256_sub(w1, w0)));
        }
        if (gtid == 0 && 2 * nodd < nq) {  // odd tail (prover_layers.h:381-388)
          const E t = fp256_mul(qw_at(2 * nodd), ld32(&Wh[2 * nodd]));
          a0 = fp256_add(a0, t);
          a2 = fp256_add(a2, t);
        }
      }
      auto wg_sum = [&](E& x0, E& x2, bool split) {  // wave shuffles, then the wave sums through LDS; result in thread 0
        if (split) {  // even waves carry only a0, odd waves only a2: one value to fold per wave
          const bool second = (wave & 1) != 0;
          E v = second ? x2 : x0;
          for (int off = 32; off > 0; off >>= 1) {
            E o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o.l[k] = __shfl_down(v.l[k], off, 64);
            v = fp256_add(v, o);
          }
          x0 = second ? e32_zero() : v;
          x2 = second ? v : e32_zero();
        } else {
          for (int off = 32; off > 0; off >>= 1) {
            E o0, o2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              o0.l[k] = __shfl_down(x0.l[k], off, 64);
              o2.l[k] = __shfl_down(x2.l[k], off, 64);
            }
            x0 = fp256_add(x0, o0);
            x2 = fp256_add(x2, o2);
          }
        }
        if (lane == 0) {
          s_red[0][wave] = x0;
          s_red[1][wave] = x2;
        }
        __syncthreads();
        if (wave == 0) {  // the 8 wave sums of a0 on lanes 0-7, of a2 on lanes 8-15: three shuffle steps instead of 14 additions in a row
          E v = lane < 16 ? s_red[lane >> 3][lane & 7] : e32_zero();
#pragma unroll
          for (int off = 4; off > 0; off >>= 1) {
            E o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o.l[k] = __shfl_down(v.l[k], off, 64);
            v = fp256_add(v, o);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            x0.l[k] = __shfl(v.l[k], 0, 64);
            x2.l[k] = __shfl(v.l[k], 8, 64);
          }
        }
        __syncthreads();
      };
      wg_sum(a0, a2, G == 1);
      bool poster = tid == 0;
      if (G > 1) {  // slots + arrival ticket: the last workgroup to arrive folds all slots
        if (tid == 0) {
          u64* sl = &gs->slots[8 * g];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            sl[k] = a0.l[k];
            sl[4 + k] = a2.l[k];
          }
          s_last = g256_arrive(gs->l1_arr, &gs->arrive, G, g) ? 1u : 0u;  // releases the slot
        }
        __syncthreads();
        poster = false;
        if (s_last) {  // uniform per workgroup
          a0 = e32_zero();
          a2 = e32_zero();
          if (tid < G) {
            const u64* sl = &gs->slots[8 * tid];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              a0.l[k] = __hip_atomic_load(&sl[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              a2.l[k] = __hip_atomic_load(&sl[4 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
          wg_sum(a0, a2, false);
          poster = tid == 0;
        }
      }
      if (poster) {  // system-scope stores into the pinned words, then the sequence number
        u64* po = (u64*)a.post;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __hip_atomic_store(&po[k], a0.l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          __hip_atomic_store(&po[4 + k], a2.l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&po[8], (u64)nh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[9], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lf_wait_stores_before_publish();  // payload acknowledged before the sequence word leaves
        __hip_atomic_store(&po[16], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    // ---- while the host works on the challenge: the layout of HQuad::bind_h
    u32 my_off, my_end, new_nh;
    {
      const u32 R = ((nh + G - 1) / G + 63) / 64 * 64;  // range per workgroup, whole waves
      const u32 lo = (u64)g * R < nh ? g * R : nh, hi = (u64)lo + R < nh ? lo + R : nh;
      if (tid == 0) {
        s_carry = 0;
        s_off = 0;
        s_tot = 0;
      }
      __syncthreads();
      if (G > 1) {
        u32 mine = 0;
        for (u32 base = lo; base < hi; base += G256_THREADS) {
          const u32 i = base + tid;
          const bool head = i < hi && !is_second256(hc, i, hand);
          mine += (u32)__popcll(__ballot(head));
        }
        if (lane == 0 && mine) atomicAdd(&s_carry, mine);
        __syncthreads();
        if (tid == 0) a.counts[g] = s_carry;
        if (!g256_barrier(gs, G, gen, a.timeout_ticks)) return;
        if (tid < G) {
          const u32 cnt = a.counts[tid];
          if (cnt) {
            atomicAdd(&s_tot, cnt);
            if (tid < g) atomicAdd(&s_off, cnt);
          }
        }
        __syncthreads();
      }
      my_off = s_off;
      __syncthreads();
      if (tid == 0) s_carry = my_off;
      __syncthreads();
      for (u32 base = lo; base < hi; base += G256_THREADS) {
        const u32 i = base + tid;
        const bool head = i < hi && !is_second256(hc, i, hand);
        const u64 mask = __ballot(head);
        if (lane == 0) s_wave[wave] = (u32)__popcll(mask);
        __syncthreads();
        u32 off = s_carry;
        for (u32 w = 0; w < wave; ++w) off += s_wave[w];
        off += (u32)__popcll(mask & ((1ull << lane) - 1));
        if (head) {
          uint2 h = hc[i];
          const u32 hh = hand ? h.y : h.x;
          const u32 kind = (i + 1 < nh && is_second256(hc, i + 1, hand)) ? 0u : ((hh & 1) == 0 ? 1u : 2u);
          if (hand) h.y = hh >> 1;
          else h.x = hh >> 1;
          hc_o[off] = h;
          a.src[off] = i | (kind << 30);
        }
        __syncthreads();
        if (tid == 0) {
          u32 tot = 0;
          for (u32 w = 0; w < G256_THREADS / 64; ++w) tot += s_wave[w];
          s_carry += tot;
        }
        __syncthreads();
      }
      my_end = s_carry;
      new_nh = G > 1 ? s_tot : my_end;
    }
    // ---- the challenge: eight 64-bit words {tag = low half of the sequence number, 32 bits of the challenge}; a read that
    // finds the tag in all eight has the whole challenge.  Workgroup 0 polls the host and hands it on through a device slot.
    if (wave == 0) {
      const u64 t0 = wall_clock64();
      const u64 tag = seq & 0xffffffffull;
      const bool from_host = g == 0;
      u64 got = 0, w = 0;
      for (;;) {
        if (lane < 8) {
          if (from_host) w = __hip_atomic_load((const u64*)&a.cmd[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          else w = __hip_atomic_load(&gs->chal[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__all(lane >= 8 || (w >> 32) == tag)) {
          got = seq;
          break;
        }
        int stop = 0;
        if (lane == 0) {
          if (G > 1 && __hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) stop = 1;
          else if (wall_clock64() - t0 > (from_host ? 1 : 2) * a.timeout_ticks) {  // the host went away: release every workgroup and report
            __hip_atomic_store(&gs->abort, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (g == 0) {
              a.post[9] = 1;
              __threadfence_system();
              __hip_atomic_store((u64*)&a.post[16], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            stop = 1;
          }
        }
        if (__shfl(stop, 0, 64)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (got == seq && g == 0 && G > 1 && lane < 8) __hip_atomic_store(&gs->chal[lane], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 lo32 = w & 0xffffffffull;
      const u64 even = __shfl(lo32, (lane & 3) * 2, 64), odd = __shfl(lo32, (lane & 3) * 2 + 1, 64);
      if (lane < 4) s_cmd[lane] = even | (odd << 32);
      if (lane == 0) s_cmd[4] = got;
    }
    __syncthreads();
    if (s_cmd[4] != seq) return;  // uniform per workgroup; the barriers of the others see the abort flag
    E r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.l[k] = s_cmd[k];
    // ---- Dense::bind of W[hand] (out of place); the HQUAD values for the layout above; and, from those values, the next
    // evaluation's accumulators -- the bound hand entry is recomputed from the unbound array (one more product) so that no
    // workgroup waits for another one's Dense::bind
    const E* Wold = W[hand];
    const u32 n0 = nW[hand], nout = (n0 + 1) / 2;
    auto bind_at = [&](u32 j) -> E {
      const E f0 = ld32(&Wold[2 * j]);
      if (2 * j + 1 < n0) return fp256_add(f0, fp256_mul(fp256_sub(ld32(&Wold[2 * j + 1]), f0), r));
      return fp256_sub(f0, fp256_mul(f0, r));
    };
    const bool more = rh + 1 < a.rh1;
    for (u32 i = gtid; i < 8 * nout; i += GT) QW[i] = 0;  // next written two round-hands from now, for this hand again
    E* const Wout = a.Wb[hand][wsel[hand]];
    auto bind_value = [&](u32 o) -> E {  // HQuad::bind_h value of output o (hquad.h:94-118)
      const u32 sidx = a.src[o], i = sidx & 0x3fffffffu, kind = sidx >> 30;
      const E v0 = ld32(&vc[i]);
      if (kind == 0) return fp256_add(v0, fp256_mul(fp256_sub(ld32(&vc[i + 1]), v0), r));
      if (kind == 1) return fp256_sub(v0, fp256_mul(v0, r));
      return fp256_mul(v0, r);
    };
    if (G == 1) {
      // One workgroup left: what counts is the longest chain of instructions a single wave issues, so the independent products go
      // to different waves -- the even waves bind the HQUAD values while the odd waves bind the hand array -- and, a workgroup
      // barrier being cheap, the next evaluation's products take the bound hand entries from memory instead of recomputing them
      const u32 half = G256_THREADS / 2, ht = (wave >> 1) * 64 + lane;
      if ((wave & 1) == 0) {
        for (u32 o = ht; o < my_end; o += half) st32(&vc_o[o], bind_value(o));
      } else {
        for (u32 i = ht; i < nout; i += half) st32(&Wout[i], bind_at(i));
      }
      __threadfence_block();
      __syncthreads();
      if (more) {
        for (u32 base = 0; base < my_end; base += G256_THREADS) {
          const u32 o = base + tid;
          u32 key = 0xffffffffu;
          E t = e32_zero();
          if (o < my_end) {
            const uint2 h = hc_o[o];
            key = hand ? h.x : h.y;  // the next evaluation is for the other hand
            t = fp256_mul(ld32(&vc_o[o]), ld32(&Wout[hand ? h.y : h.x]));
          }
          run_fold_commit256(key, t, QWn);
        }
      }
    } else {
    for (u32 i = gtid; i < nout; i += GT) st32(&Wout[i], bind_at(i));
    for (u32 base = my_off; base < my_end; base += G256_THREADS) {
      const u32 o = base + tid;
      u32 key = 0xffffffffu;
      E t = e32_zero();
      if (o < my_end) {
        const u32 sidx = a.src[o], i = sidx & 0x3fffffffu, kind = sidx >> 30;
        const E v0 = ld32(&vc[i]);
        E v;
        if (kind == 0) v = fp256_add(v0, fp256_mul(fp256_sub(ld32(&vc[i + 1]), v0), r));  // HQuad::bind_h (hquad.h:94-118)
        else if (kind == 1) v = fp256_sub(v0, fp256_mul(v0, r));
        else v = fp256_mul(v0, r);
        st32(&vc_o[o], v);
        if (more) {
          const uint2 h = hc_o[o];
          key = hand ? h.x : h.y;  // the next evaluation is for the other hand
          t = fp256_mul(v, bind_at(hand ? h.y : h.x));
        }
      }
      if (more) run_fold_commit256(key, t, QWn);
    }
    }
    W[hand] = Wout;
    nW[hand] = nout;
    wsel[hand] ^= 1;
    {
      nh = new_nh;
      uint2* th = const_cast<uint2*>(hc);
      E* tv = const_cast<E*>(vc);
      hc = hc_o;
      vc = vc_o;
      hc_o = th;
      vc_o = tv;
      u64* tq = QW;
      QW = QWn;
      QWn = tq;
    }
  }
  if (g == 0 && tid == 0) {  // end of the layer: W[R,C], W[L,C] and HQUAD->scalar() (the last binds of this workgroup: G == 1 by now)
    __threadfence();
    const E w0 = nW[0] ? ld32(&W[0][0]) : e32_zero(), w1 = nW[1] ? ld32(&W[1][0]) : e32_zero(), sc = nh ? ld32(&vc[0]) : e32_zero();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a.post[k] = w0.l[k];
      a.post[4 + k] = w1.l[k];
      a.post[12 + k] = sc.l[k];
    }
    a.post[8] = nh;
    a.post[9] = 0;
    __threadfence_system();
    __hip_atomic_store((u64*)&a.post[16], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- Ligero row combinations (ligero.hip / sumcheck.hip, over 32-byte elements)
// y[j] += sum_i u[i] T[i][j]: 64 columns x 4 row slices per workgroup
__global__ __launch_bounds__(256) void rows_axpy256_kernel(u32 nrows, size_t n, E* __restrict__ y, const E* __restrict__ u, const E* __restrict__ T, size_t ld) {
  __shared__ E part[4][64];
  const u32 col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const size_t j = (size_t)blockIdx.x * 64 + col;
  E acc = e32_zero();
  if (j < n)
    for (u32 i = slice; i < nrows; i += 4) acc = fp256_add(acc, fp256_mul(ld32(&T[(size_t)i * ld + j]), ld32(&u[i])));
  part[slice][col] = acc;
  __syncthreads();
  if (slice == 0 && j < n) {
    acc = ld32(&y[j]);
    for (u32 k = 0; k < 4; ++k) acc = fp256_add(acc, part[k][col]);
    st32(&y[j], acc);
  }
}
// y[j] = T0[j] + sum_i A[i][j] T[i][j]   (dot_proof, Blas::vaxpy blas.h:71-78)
__global__ __launch_bounds__(256) void rows_vaxpy256_kernel(u32 nrows, size_t n, const E* __restrict__ T0, const E* __restrict__ A, size_t lda,
                                                            const E* __restrict__ T, size_t ld, E* __restrict__ y) {
  __shared__ E part[4][64];
  const u32 col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const size_t j = (size_t)blockIdx.x * 64 + col;
  E acc = e32_zero();
  if (j < n)
    for (u32 i = slice; i < nrows; i += 4) acc = fp256_add(acc, fp256_mul(ld32(&T[(size_t)i * ld + j]), ld32(&A[(size_t)i * lda + j])));
  part[slice][col] = acc;
  __syncthreads();
  if (slice == 0 && j < n) {
    acc = ld32(&T0[j]);
    for (u32 k = 0; k < 4; ++k) acc = fp256_add(acc, part[k][col]);
    st32(&y[j], acc);
  }
}
// y[j] = Tq[j] + sum_i u[i] (z_i[j] - x_i[j] y_i[j])   (quadratic_proof :311-333)
__global__ __launch_bounds__(Z_THREADS) void quad_combo256_kernel(u32 nt, size_t n, const E* __restrict__ Tq, const E* __restrict__ u, const E* __restrict__ X,
                                                                  const E* __restrict__ Y, const E* __restrict__ Zr, size_t ld, E* __restrict__ y) {
  const size_t j = (size_t)blockIdx.x * Z_THREADS + threadIdx.x;
  if (j >= n) return;
  E acc = ld32(&Tq[j]);
  for (u32 i = 0; i < nt; ++i) {
    const E t = fp256_sub(ld32(&Zr[(size_t)i * ld + j]), fp256_mul(ld32(&X[(size_t)i * ld + j]), ld32(&Y[(size_t)i * ld + j])));
    acc = fp256_add(acc, fp256_mul(ld32(&u[i]), t));
  }
  st32(&y[j], acc);
}
// rows[i][r + j] = scale * dense[i w + j]; then rows[pos(idx)] += val (inner_product_vector + layout_Aext, ligero_param.h:382-430)
__global__ __launch_bounds__(Z_THREADS) void a_rows_dense256_kernel(u32 r, u32 w, size_t ld, E scale, const E* __restrict__ dense, size_t n, E* __restrict__ rows) {
  const size_t t = (size_t)blockIdx.x * Z_THREADS + threadIdx.x;
  if (t >= n) return;
  const size_t i = t / w, j = t % w;
  st32(&rows[i * ld + r + j], fp256_mul(scale, ld32(&dense[t])));
}
__global__ __launch_bounds__(Z_THREADS) void a_rows_sparse256_kernel(u32 r, u32 w, size_t ld, const u64* __restrict__ idx, const E* __restrict__ val, size_t n,
                                                                     E* __restrict__ rows) {
  const size_t t = (size_t)blockIdx.x * Z_THREADS + threadIdx.x;
  if (t >= n) return;
  const size_t i = idx[t] / w, j = idx[t] % w;
  E* dst = &rows[i * ld + r + j];
  st32(dst, fp256_add(ld32(dst), ld32(&val[t])));
}
__global__ __launch_bounds__(Z_THREADS) void gather_columns256_kernel(u32 nrow, size_t ld, size_t col0, const E* __restrict__ T, const u64* __restrict__ idx,
                                                                      u32 nreq, E* __restrict__ req) {
  const u32 t = blockIdx.x * Z_THREADS + threadIdx.x;
  if (t >= nrow * nreq) return;
  const u32 i = t / nreq, j = t % nreq;
  st32(&req[t], ld32(&T[(size_t)i * ld + col0 + idx[j]]));
}

// Quad::bind_gh_all (lib/sumcheck/quad.h:188-210), the verifier's combined bind: the sum over all corners of
// prep_v(v, beta) (EQ(G0,g) + alpha EQ(G1,g)) EQ(H0,h0) EQ(H1,h1); block sums go as limbs into 8 words
__global__ __launch_bounds__(Z_THREADS) void bind_gh_all256_kernel(size_t n, const corner4* __restrict__ t, const E* __restrict__ kvec, const E* __restrict__ eqg,
                                                                   const E* __restrict__ eqh0, const E* __restrict__ eqh1, E beta, u64* __restrict__ acc) {
  __shared__ E sh[Z_THREADS];
  E s = e32_zero();
  for (size_t i = (size_t)blockIdx.x * Z_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * Z_THREADS) {
    const corner4 cr = t[i];
    E v = ld32(&kvec[cr.vi]);
    if (e32_is_zero(v)) v = beta;  // prep_v: assert-zero terms carry beta (quad.h:213-220)
    E qv = fp256_mul(v, ld32(&eqg[cr.g]));
    qv = fp256_mul(qv, ld32(&eqh0[cr.h0]));
    s = fp256_add(s, fp256_mul(qv, ld32(&eqh1[cr.h1])));
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) limbs_atomic_add(acc, s);
}

inline u32 nblk(size_t n, u32 per = Z_THREADS) { return (u32)((n + per - 1) / per ? (n + per - 1) / per : 1); }
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ------------------------------------------------------------------ host field (FpGeneric over the P-256 prime)
struct F256 {
  E zero = e32_zero(), one, pts[3], invden[3], rsq;
  F256() {
    rsq = h256_rsq();
    one = h256_of_scalar(1);
    pts[0] = zero;  // poly_evaluation_points 0, 1, 2 (fp_generic.h:114-121)
    pts[1] = one;
    pts[2] = h256_of_scalar(2);
    for (int i = 0; i < 3; ++i) {
      E d = one;
      for (int j = 0; j < 3; ++j)
        if (j != i) d = fp256_mul(d, fp256_sub(pts[i], pts[j]));
      invden[i] = h256_inv(d);
    }
  }
  static E add(const E& a, const E& b) { return fp256_add(a, b); }
  static E sub(const E& a, const E& b) { return fp256_sub(a, b); }
  static E mul(const E& a, const E& b) { return fp256_mul(a, b); }
  // Poly<3>::eval_monomial (lib/algebra/poly.h:100-108)
  E eval_monomial(const E coef[3], const E& x) const { return add(mul(add(mul(coef[2], x), coef[1]), x), coef[0]); }
  // the quadratic through (pts[i], ev[i]) at x = Poly<3>::eval_lagrange (poly.h:72-98)
  E eval_lagrange(const E ev[3], const E& x) const {
    E acc = zero;
    for (int i = 0; i < 3; ++i) {
      E num = one;
      for (int j = 0; j < 3; ++j)
        if (j != i) num = mul(num, sub(x, pts[j]));
      acc = add(acc, mul(ev[i], mul(num, invden[i])));
    }
    return acc;
  }
};

// ------------------------------------------------------------------ device steps
int raw_eq2_256(lfgpu_ctx* c, const F256& F, size_t logn, size_t n, const E* G0, const E* G1, const E& alpha, E* d_eq) {
  if (n == 0) return LFGPU_OK;
  std::vector<E> Gt(4 * logn + 1);
  for (size_t l = 0; l < logn; ++l) {
    Gt[l] = G0[l];
    Gt[logn + l] = G1[l];
    Gt[2 * logn + l] = F.sub(F.one, G0[l]);
    Gt[3 * logn + l] = F.sub(F.one, G1[l]);
  }
  const u32 lb = (u32)(logn / 2), hb = (u32)logn - lb;
  const size_t ntab = 2 * (((size_t)1 << lb) + ((size_t)1 << hb));
  void* d_G = nullptr;
  LF_TRY(lf_scratch2(c, (Gt.size() + ntab) * 32 + 64, &d_G));
  E* d_tab = (E*)d_G + Gt.size();
  if (Gt.size() * 32 <= LF_STAGE_SLOT) {  // 4 logn + 1 <= 128 elements: through the pinned ring, no synchronisation
    LF_TRY(lf_stage_upload(c, d_G, Gt.data(), Gt.size() * 32));
  } else {
    LF_HIP(c, hipMemcpyAsync(d_G, Gt.data(), Gt.size() * 32, hipMemcpyHostToDevice, c->stream));
    LF_HIP(c, hipStreamSynchronize(c->stream));  // Gt is a local
  }
  if (logn < 6) {  // tiny: the direct product per entry
    hipLaunchKernelGGL(raw_eq2_256_kernel, dim3(nblk(n)), dim3(Z_THREADS), 0, c->stream, (u32)logn, (u32)n, (const E*)d_G, alpha, F.one, d_eq);
  } else {
    hipLaunchKernelGGL(eq_tables256_kernel, dim3(nblk(ntab)), dim3(Z_THREADS), 0, c->stream, (u32)logn, lb, (const E*)d_G, alpha, F.one, d_tab);
    hipLaunchKernelGGL(raw_eq2_split256_kernel, dim3(nblk(n)), dim3(Z_THREADS), 0, c->stream, (u32)logn, lb, (u32)n, (const E*)d_tab, d_eq);
  }
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

typedef void (*round256_fn)(void* user, size_t hand, size_t rnd, const E ev[3], E* chal);

// One layer of the sumcheck (ProverLayers::layer): bind_g, then per round and hand the QW scatter, the two partial sums
// (one read-back), the caller's transcript step, Dense::bind and HQuad::bind_h.  d_W (the layer's inputs) is left intact:
// both hands bind into ping-pong buffers.
int sumcheck_layer256(lfgpu_quad* q, const F256& F, size_t logv, const E* G0, const E* G1, const E& alpha, const E& beta, size_t logw, size_t nw,
                      const E* d_W, const E wc_in[2], round256_fn round, void* user, E wc_out[2], E* g_out /*[2][logw]*/, E* bound_quad) {
  lfgpu_ctx* c = q->c;
  if (nw == 0 || logw > 40 || nw > ((size_t)1 << logw) || nw <= q->hmax) return lf_fail(c, LFGPU_ERR_ARG, "sumcheck_layer256: nw must exceed the largest hand index");
  if (logv > 40 || ((size_t)1 << logv) < q->nv) return lf_fail(c, LFGPU_ERR_ARG, "sumcheck_layer256: 2^logv < nv");
  const size_t nt = q->n, nh0 = q->nh0, half = (nw + 1) / 2;
  // scratch: eq | limb accumulators (bind_g runs, then the QW of every round-hand; self-cleaning) | hc[2] | vc[2] |
  // 4 half hand buffers | the round's two sums as limbs
  const size_t acc_n = std::max(nh0, nw);
  const size_t grid_bytes = acc_n * 64 + sizeof(Grid256Sync) + G256_WGS * 4 + nh0 * 4 + 256;  // second accumulator array, barrier state, counts, src
  const size_t bytes = q->nv * 32 + acc_n * 64 + 2 * nh0 * 8 + 2 * nh0 * 32 + 4 * half * 32 + 256 + 1024 + grid_bytes;
  void* sc = nullptr;
  LF_TRY(lf_scratch(c, bytes, &sc));
  uint8_t* b = (uint8_t*)sc;
  E* d_eq = (E*)b;                     b += q->nv * 32;
  u64* acc = (u64*)b;                  b += acc_n * 64;
  uint2* hc[2] = {(uint2*)b, (uint2*)(b + nh0 * 8)};  b += 2 * nh0 * 8;
  b = (uint8_t*)(((uintptr_t)b + 31) & ~(uintptr_t)31);
  E* vc[2] = {(E*)b, (E*)b + nh0};     b += 2 * nh0 * 32;
  E* wb[2][2] = {{(E*)b, (E*)b + half}, {(E*)b + 2 * half, (E*)b + 3 * half}};  b += 4 * half * 32;
  u64* d_sums = (u64*)b;
  u32* d_done = (u32*)(d_sums + 16);
  b += 256;
  u64* acc2 = (u64*)b;                 b += acc_n * 64;
  Grid256Sync* gsync = (Grid256Sync*)b; b += sizeof(Grid256Sync);
  u32* g_counts = (u32*)b;             b += G256_WGS * 4;
  u32* g_src = (u32*)b;
  volatile u64* post = c->poll_h + 256;  // coherent pinned words a running kernel writes and the host polls (ctx.h)
  volatile u64* cmd = c->poll_h + 320;   // ... and the host's answers (the challenge as tagged words)
  // Quad::bind_g
  LF_TRY(raw_eq2_256(c, F, logv, q->nv, G0, G1, alpha, d_eq));
  LF_HIP(c, hipMemsetAsync(acc, 0, acc_n * 64, c->stream));
  LF_HIP(c, hipMemsetAsync(d_sums, 0, 192, c->stream));  // the 16 sum words + the block counter
  hipLaunchKernelGGL(bindg_emit256_kernel, dim3(nblk(nt, BG_THREADS)), dim3(BG_THREADS), 0, c->stream, nt, (const corner4*)q->d_morton, (const E*)q->d_kvec,
                     (const E*)d_eq, beta, (const u32*)q->d_runoff, hc[0], acc);
  hipLaunchKernelGGL(limb_normalize256_kernel, dim3(nblk(nh0)), dim3(Z_THREADS), 0, c->stream, nh0, acc, F.rsq, vc[0]);
  LF_HIP(c, hipGetLastError());
  size_t nh = nh0;
  int cur = 0;
  E sum = F.add(wc_in[0], F.mul(alpha, wc_in[1]));
  const E* WH[2] = {d_W, d_W};
  size_t nW[2] = {nw, nw};
  int wsel[2] = {0, 0};
  if (q->bind_shape.size() < 2 * logw) q->bind_shape.resize(2 * logw, lfgpu_quad::BindShape{nullptr, 0, 0});
  u64 h_sums[16];
  E* h_out = (E*)c->mailbox_h;
  auto wait_post = [&](u64 seq) -> int {  // bounded: a kernel that is over without posting is an error
    // The stream is only looked at after 50 ms without the post (a dead kernel must not hang the caller): in a healthy run no HIP
    // call is made while a resident kernel waits for this thread.  It matters -- another thread's hipFree holds the runtime's
    // lock while it waits for every stream of the device, hence for that kernel; a hipStreamQuery here would wait for the lock
    // and the kernel for its challenge until its timeout.
    u64 spins = 0;
    double t_first = 0;
    while (__atomic_load_n((const u64*)&post[16], __ATOMIC_ACQUIRE) != seq) {
      if (spins > 0x8000 && (spins & 0xff) == 0) sched_yield();  // see sc_wait_post (sumcheck.hip)
      if ((++spins & 0xfff) == 0) {
        const double t = now_ms();
        if (t_first == 0) t_first = t;
        if (t - t_first < 50.0) continue;
        const hipError_t qe = hipStreamQuery(c->stream);
        if (qe == hipSuccess) {
          if (__atomic_load_n((const u64*)&post[16], __ATOMIC_ACQUIRE) == seq) break;
          return lf_fail(c, LFGPU_ERR_ASSERT, "sumcheck_layer256: kernel finished without posting");
        }
        if (qe != hipErrorNotReady) return lf_fail(c, LFGPU_ERR_HIP, "sumcheck_layer256: %s", hipGetErrorString(qe));
      }
    }
    return LFGPU_OK;
  };
  // LFGPU_P256_GRID [1]: once the HQUAD and both hand arrays have <= G256_MAX entries the rest of the layer is ONE launch of
  // co-resident workgroups that take every challenge through pinned memory (grid256_layer_kernel); 0 = three launches per
  // round-hand throughout (A/B, and the fallback where a running kernel cannot see host writes)
  static const int grid_env = getenv("LFGPU_P256_GRID") ? atoi(getenv("LFGPU_P256_GRID")) : 1;
  bool grid_ok = grid_env && lf_sc_resident_ok(c) && c->grid_strikes < 2;
  struct CuGuard {  // the grid's CUs go back to the device's budget (ctx.h) however this function is left, after the kernel has ended
    lfgpu_ctx* c;
    ~CuGuard() {
      if (c->cu_held) {
        (void)hipStreamSynchronize(c->stream);
        lf_cu_release(c, -1);
      }
    }
  } cu_guard{c};
  for (size_t rnd = 0; rnd < logw; ++rnd)
    for (int hand = 0; hand < 2; ++hand) {
      // hand-off point: above it a round-hand is three launches spread over the whole chip.  Measured on the mdoc signature
      // circuit (sumcheck ms): no grid 19.8; grid from 1024 / 2048 / 8192 / 32768 / 131072 entries on: 19.1 / 19.1 / 19.8 / 20.0 /
      // 20.1.  With LFGPU_VERBOSE the host's turn shows as 1.2 us per round-hand and the wait for the device as 23 us for a
      // layer that starts at 409 entries (one workgroup) up to 41 us from 69150: ~5000 instructions on the critical path of a
      // round-hand (0.82 us per product, tools/ubench_p256; the 8-wave sum with its serial tail; the recomputed bound hand
      // entry) at the one-instruction-per-5.6-cycles issue rate of a lone wave -- not launches, barriers or memory round trips
      // (an LDS-resident tail for <= 512 entries was built and measured: no change)
      static const size_t grid_max = [] {
        const char* e = getenv("LFGPU_P256_GRID_MAX");
        return std::min<size_t>(e ? (size_t)atol(e) : (size_t)2048, G256_MAX);
      }();
      static const u32 grid_per_wg = getenv("LFGPU_P256_PER_WG") ? (u32)std::max(64, atoi(getenv("LFGPU_P256_PER_WG"))) : 512u;
      u32 want_G = 0;  // workgroups of the grid this round-hand could hand the layer to (0: not now)
      if (grid_ok && nh <= grid_max && nW[0] <= grid_max && nW[1] <= grid_max) {
        int pc = 0;  // all of them must be resident together (they synchronise through device memory): one fits a CU ...
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, (const void*)grid256_layer_kernel, G256_THREADS, 0) != hipSuccess) pc = 0;
        if (pc < 1) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "sumcheck_layer256: the grid kernel does not fit a CU");
        const size_t big = std::max(nh, std::max(nW[0], nW[1]));
        want_G = std::min<u32>(std::max<u32>((u32)((big + grid_per_wg - 1) / grid_per_wg), 1), std::min<u32>(G256_WGS, (u32)c->num_cu));
      }
      // ... and the device's CU budget must have room for them (ctx.h); without it this round-hand takes the three launches
      // below and the next one asks again
      if (want_G && lf_cu_acquire(c, (int)want_G)) {
        const u32 per_wg = grid_per_wg, G = want_G;
        const size_t big = std::max(nh, std::max(nW[0], nW[1]));
        Grid256 a{};
        a.hcA = hc[cur];
        a.vcA = vc[cur];
        a.hcB = hc[1 - cur];
        a.vcB = vc[1 - cur];
        a.nh = (u32)nh;
        for (int h = 0; h < 2; ++h) {
          a.W[h] = WH[h];
          a.nW[h] = (u32)nW[h];
          a.Wb[h][0] = wb[h][0];
          a.Wb[h][1] = wb[h][1];
        }
        a.QW = acc;
        a.QW2 = acc2;
        a.rh0 = (u32)(2 * rnd + hand);
        a.rh1 = (u32)(2 * logw);
        a.seq0 = c->poll_seq + 1;
        c->poll_seq += (a.rh1 - a.rh0) + 1;
        a.timeout_ticks = 5000ull * c->wall_khz;
        a.post = post;
        a.cmd = cmd;
        a.gs = gsync;
        a.counts = g_counts;
        a.src = g_src;
        a.per_wg = per_wg;
        static const int wave_tail_env = getenv("LFGPU_P256_WAVE_TAIL") ? atoi(getenv("LFGPU_P256_WAVE_TAIL")) : 1;
        a.wave_tail = (u32)wave_tail_env;
        a.rsq = F.rsq;
        static const u64 place_ms = getenv("LFGPU_SC_PLACE_MS") ? (u64)std::max(1, atoi(getenv("LFGPU_SC_PLACE_MS"))) : 250ull;
        a.place_ticks = place_ms * c->wall_khz;
        {  // test hook: the first LFGPU_P256_TEST_DROP grid launches of the process lose their last workgroup
          static std::atomic<int> drops{getenv("LFGPU_P256_TEST_DROP") ? atoi(getenv("LFGPU_P256_TEST_DROP")) : 0};
          if (G > 1 && drops.load() > 0 && drops.fetch_sub(1) > 0) a.test_drop = 1;
        }
        LF_HIP(c, hipMemsetAsync(gsync, 0, sizeof(Grid256Sync), c->stream));
        hipLaunchKernelGGL(grid256_layer_kernel, dim3(G), dim3(G256_THREADS), 0, c->stream, a);
        LF_HIP(c, hipGetLastError());
        u64 seq = a.seq0;
        size_t r2 = rnd;
        bool not_placed = false;
        static const bool verbose = getenv("LFGPU_VERBOSE") != nullptr;
        double t_wait = 0, t_host = 0, tp0 = verbose ? now_ms() : 0;
        for (u32 rh = a.rh0; rh < a.rh1; ++rh, ++seq) {
          const int hd = (int)(rh & 1);
          r2 = rh >> 1;
          LF_TRY(wait_post(seq));
          if (verbose) {
            const double t = now_ms();
            t_wait += t - tp0;
            tp0 = t;
          }
          if (post[9] == 2 && rh == a.rh0) {  // the grid was not placed whole: nothing but scratch was written
            not_placed = true;
            break;
          }
          if (post[9] != 0)
            return lf_fail(c, LFGPU_ERR_ASSERT, "sumcheck_layer256: the grid kernel timed out waiting for a challenge (status %llu at round-hand %u of [%u, %u), %u workgroups)",
                           (unsigned long long)post[9], rh, a.rh0, a.rh1, G);
          E coef[3], ev[3], r;
          for (int k = 0; k < 4; ++k) {
            coef[0].l[k] = post[k];
            coef[2].l[k] = post[4 + k];
          }
          coef[1] = F.sub(F.sub(F.sub(sum, coef[0]), coef[0]), coef[2]);
          for (int k = 0; k < 3; ++k) ev[k] = F.eval_monomial(coef, F.pts[k]);
          round(user, (size_t)hd, r2, ev, &r);
          g_out[hd * logw + r2] = r;
          sum = F.eval_lagrange(ev, r);
          const u64 tag = (seq & 0xffffffffull) << 32;  // eight tagged words: valid as soon as all eight carry the tag
          for (int k = 0; k < 8; ++k) __atomic_store_n((u64*)&cmd[k], tag | ((r.l[k >> 1] >> (32 * (k & 1))) & 0xffffffffull), __ATOMIC_RELAXED);
          __atomic_thread_fence(__ATOMIC_RELEASE);
          if (verbose) {
            const double t = now_ms();
            t_host += t - tp0;
            tp0 = t;
          }
        }
        if (verbose)
          fprintf(stderr, "lfgpu sumcheck_layer256 grid: %u round-hands from %zu entries: waiting for the device %.1f us, host's turn %.1f us per round-hand\n",
                  a.rh1 - a.rh0, big, 1e3 * t_wait / (a.rh1 - a.rh0), 1e3 * t_host / (a.rh1 - a.rh0));
        if (not_placed) {  // another process holds CUs (the budget rules it out within this one): per-launch kernels from here on
          LF_HIP(c, hipStreamSynchronize(c->stream));
          lf_cu_release(c, -1);
          grid_ok = false;
          ++c->grid_strikes;
          LF_HIP(c, hipMemsetAsync(acc, 0, acc_n * 64, c->stream));  // the kernel's clears / first sums may have touched the accumulators
          if (verbose) fprintf(stderr, "lfgpu sumcheck_layer256: resident grid not placed (strike %d): per-launch kernels for this layer\n", c->grid_strikes);
        } else {
        LF_TRY(wait_post(seq));  // the layer's last post: W[0][0], W[1][0], the HQUAD scalar
        lf_cu_release(c, -1);
        if (post[9] != 0 || post[8] != 1) return lf_fail(c, LFGPU_ERR_ASSERT, "sumcheck_layer256: HQUAD did not fold to one entry (%llu)", (unsigned long long)post[8]);
        for (int k = 0; k < 4; ++k) {
          wc_out[0].l[k] = post[k];
          wc_out[1].l[k] = post[4 + k];
          if (bound_quad) bound_quad->l[k] = post[12 + k];
        }
        return LFGPU_OK;
        }
      }
      // QW scatter (prover_layers.h:239-243) + evaluations: two launches, one read-back
      if (nh) hipLaunchKernelGGL(qw_scatter256_kernel, dim3(nblk(nh)), dim3(Z_THREADS), 0, c->stream, nh, (const uint2*)hc[cur], (const E*)vc[cur], hand,
                                 WH[1 - hand], acc);
      const u64 seq = ++c->poll_seq;
      hipLaunchKernelGGL(partials256_kernel, dim3(std::min<u32>(nblk(nW[hand] / 2), 256)), dim3(Z_THREADS), 0, c->stream, nW[hand], acc, F.rsq, WH[hand], d_sums,
                         d_done, post, seq);
      LF_HIP(c, hipGetLastError());
      LF_TRY(wait_post(seq));
      for (int k = 0; k < 16; ++k) h_sums[k] = post[k];
      // coef[0] = a0, coef[2] = a2, coef[1] from the running sum (prover_layers.h:390-396, logc = 0)
      E coef[3], ev[3], r;
      coef[0] = fp256_reduce_limbs(h_sums, F.rsq);
      coef[2] = fp256_reduce_limbs(h_sums + 8, F.rsq);
      coef[1] = F.sub(F.sub(F.sub(sum, coef[0]), coef[0]), coef[2]);
      for (int k = 0; k < 3; ++k) ev[k] = F.eval_monomial(coef, F.pts[k]);
      round(user, (size_t)hand, rnd, ev, &r);
      g_out[hand * logw + rnd] = r;
      sum = F.eval_lagrange(ev, r);
      // Dense::bind of this hand + HQuad::bind_h: one launch.  The merge structure of a round-hand is a circuit constant:
      // counted at the first proof, kept with the layer from then on
      lfgpu_quad::BindShape& bs = q->bind_shape[2 * rnd + hand];
      const u32 nbh = nh ? nblk(nh) : 0, nbd = nblk((nW[hand] + 1) / 2);
      if (nh && !(bs.d_off && bs.n_in == nh)) {
        if (bs.d_off) (void)hipFree(bs.d_off);
        bs = lfgpu_quad::BindShape{nullptr, nh, 0};
        if (hipMalloc((void**)&bs.d_off, (size_t)nbh * 4) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "sumcheck_layer256: bind offsets");
        u32* total = (u32*)((uint8_t*)c->mailbox_d + 64);
        hipLaunchKernelGGL(hquad_count256_kernel, dim3(nbh), dim3(Z_THREADS), 0, c->stream, nh, (const uint2*)hc[cur], hand, bs.d_off);
        hipLaunchKernelGGL(scan256_kernel, dim3(1), dim3(1024), 0, c->stream, nbh, bs.d_off, total);
        u32 tot = 0;
        LF_HIP(c, hipMemcpyAsync(&tot, total, 4, hipMemcpyDeviceToHost, c->stream));
        LF_HIP(c, hipStreamSynchronize(c->stream));
        bs.n_out = tot;
      }
      E* dst = wb[hand][wsel[hand]];
      hipLaunchKernelGGL(bind256_kernel, dim3(nbd + nbh), dim3(Z_THREADS), 0, c->stream, nbd, nW[hand], WH[hand], dst, nh, (const uint2*)hc[cur],
                         (const E*)vc[cur], r, hand, (const u32*)bs.d_off, hc[1 - cur], vc[1 - cur]);
      LF_HIP(c, hipGetLastError());
      WH[hand] = dst;
      wsel[hand] ^= 1;
      nW[hand] = (nW[hand] + 1) / 2;
      if (nh) {
        nh = bs.n_out;
        cur = 1 - cur;
      }
    }
  // W[0][0], W[1][0] and the bound quad (the HQUAD has shrunk to one entry)
  if (nh != 1) return lf_fail(c, LFGPU_ERR_ASSERT, "sumcheck_layer256: HQUAD did not fold to one entry (%zu)", nh);
  LF_HIP(c, hipMemcpyAsync(&h_out[0], WH[0], 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipMemcpyAsync(&h_out[1], WH[1], 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipMemcpyAsync(&h_out[2], vc[cur], 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  wc_out[0] = h_out[0];
  wc_out[1] = h_out[1];
  if (bound_quad) *bound_quad = h_out[2];
  return LFGPU_OK;
}
}  // namespace

int lf256_eval_quad_async(lfgpu_quad* q, const void* d_W, void* d_V, int* d_fail) {
  lfgpu_ctx* c = q->c;
  void* acc = nullptr;
  LF_TRY(lf_scratch2(c, q->nv * 64 + 64, &acc));
  LF_HIP(c, hipMemsetAsync(acc, 0, q->nv * 64, c->stream));
  hipLaunchKernelGGL(eval_quad256_kernel, dim3(nblk(q->n)), dim3(Z_THREADS), 0, c->stream, q->n, (const corner4*)q->d_bygate, (const E*)q->d_kvec, (const E*)d_W,
                     (u64*)acc, d_fail);
  hipLaunchKernelGGL(limb_normalize256_kernel, dim3(nblk(q->nv)), dim3(Z_THREADS), 0, c->stream, q->nv, (u64*)acc, h256_rsq(), (E*)d_V);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

// ------------------------------------------------------------------ Ligero over 32-byte elements (one GPU holds all rows)
namespace {
struct Lig256 {
  lfgpu_ctx* c = nullptr;
  lfgpu_ligero_param p{};
  E* d_T = nullptr;  // [nrow][block_enc]
  uint8_t* d_layers = nullptr;
  std::vector<uint8_t> nonces;
  // rows [row_lo, row_hi) live in d_T: all of them on one GPU, a slab with a communicator (lfgpu_zk_prover_set_comm above the
  // size threshold): the prove functions then fold the ranks' partial vectors with the field's addition, as ligero.hip does
  size_t row_lo = 0, row_hi = 0;
  bool sharded = false;
  lfgpu_comm_ops comm{};
  std::vector<std::pair<size_t, size_t>> spans;
  bool has(size_t i) const { return i >= row_lo && i < row_hi; }
  E* row(size_t i) const { return d_T + (i - row_lo) * p.block_enc; }
  size_t slab_bytes() const { return std::max<size_t>(row_hi - row_lo, 1) * p.block_enc * 32; }
  ~Lig256() {
    if (d_T) {  // the tableau holds the witness, the pads and the blinding rows: scrub it (enqueued in order behind the prover's
                // last kernels), then keep the buffer for the next commit of this shape: a hipFree per proof would wait for every
                // stream of the device (ctx.h, lf_pool_put)
      (void)hipMemsetAsync(d_T, 0, slab_bytes(), c->stream);
      lf_pool_put(c, d_T, slab_bytes());
    }
    if (d_layers) lf_pool_put(c, d_layers, 2 * p.block_ext * 32);
  }
};

// the host half of LigeroProver::commit (ligero_prover.h:171-270 + merkle_commitment.h:52-54): every RandomEngine draw in the
// reference's order (FpGeneric::sample and sample_subfield are the same function, fp_generic.h:360-376)
int lig256_layout(const lfgpu_ligero_param& p, const E* W, const size_t* lqc, lfgpu_rng_fn rng, void* user, E* H, uint8_t* nonces, char* err, bool exact) {
  const size_t hw = p.dblock;
  auto elts = [&](E* out, size_t n) {
    if (exact) {  // one call per attempt, as FpGeneric::sample draws (fp_generic.h:360-371)
      for (size_t i = 0; i < n; ++i) h256_sample_many(out + i, 1, [&](uint8_t* b, size_t nb) { rng(user, b, nb); });
    } else {
      h256_sample_many(out, n, [&](uint8_t* b, size_t nb) { rng(user, b, nb); });
    }
  };
  auto row = [&](size_t i) { return H + i * hw; };
  std::fill(H, H + p.nrow * hw, e32_zero());
  elts(row(p.ildt), p.block);  // layout_blinding_rows (:171-205)
  {
    E* d = row(p.idot);
    elts(d, p.dblock);
    E sum = e32_zero();
    for (size_t j = 0; j < p.w; ++j) sum = fp256_add(sum, d[p.r + j]);
    d[p.r] = fp256_sub(d[p.r], sum);
  }
  {
    E* q = row(p.iquad);
    elts(q, p.dblock);
    for (size_t j = 0; j < p.w; ++j) q[p.r + j] = e32_zero();
  }
  for (size_t i = 0; i < p.nwrow; ++i) {  // layout_witness_rows (:207-231)
    E* t = row(i + p.iw);
    elts(t, p.r);
    const size_t max_col = std::min(p.w, p.nw - i * p.w);
    for (size_t j = 0; j < max_col; ++j) t[p.r + j] = W[i * p.w + j];
  }
  const size_t iqx = p.iq, iqy = iqx + p.nqtriples, iqz = iqy + p.nqtriples;  // layout_quadratic_rows (:233-270)
  for (size_t i = 0; i < p.nqtriples; ++i) {
    elts(row(iqx + i), p.r);
    elts(row(iqy + i), p.r);
    elts(row(iqz + i), p.r);
    for (size_t j = 0; j < p.w && j + i * p.w < p.nq; ++j) {
      const size_t* l = &lqc[3 * (j + i * p.w)];
      if (l[0] >= p.nw || l[1] >= p.nw || l[2] >= p.nw) {
        snprintf(err, 256, "ligero_commit: lqc index >= nw");
        return LFGPU_ERR_ARG;
      }
      if (!e32_eq(fp256_mul(W[l[0]], W[l[1]]), W[l[2]])) {
        snprintf(err, 256, "ligero_commit: invalid quadratic constraints (ligero_prover.h:259-260)");
        return LFGPU_ERR_ASSERT;
      }
      row(iqx + i)[j + p.r] = W[l[0]];
      row(iqy + i)[j + p.r] = W[l[1]];
      row(iqz + i)[j + p.r] = W[l[2]];
    }
  }
  if (exact) for (size_t j = 0; j < p.block_ext; ++j) rng(user, nonces + 32 * j, 32);
  else rng(user, nonces, 32 * p.block_ext);  // MerkleCommitment::commit: one nonce per leaf, after the layout
  return LFGPU_OK;
}

// field sum over the ranks of a host vector every rank holds a partial of (all_gather + fold: RCCL has no mod-p reduction)
static int comm_fold256(const Lig256* L, E* y, size_t n) {
  const lfgpu_comm_ops& cm = L->comm;
  if (!L->sharded || cm.world == 1) return LFGPU_OK;
  std::vector<E> all((size_t)cm.world * n);
  if (cm.all_gather(cm.user, y, all.data(), n * 32, 0, nullptr)) return lf_fail(L->c, LFGPU_ERR_HIP, "ligero256 (sharded): all_gather hook failed");
  for (size_t j = 0; j < n; ++j) y[j] = all[j];
  for (int q = 1; q < cm.world; ++q)
    for (size_t j = 0; j < n; ++j) y[j] = fp256_add(y[j], all[(size_t)q * n + j]);
  return LFGPU_OK;
}
// witness / quadratic rows [a_lo, a_hi) (relative to iw) that the slab holds
static void owned_wq256(const Lig256* L, size_t* a_lo, size_t* a_hi) {
  const lfgpu_ligero_param& p = L->p;
  const size_t lo = std::min(std::max(L->row_lo, p.iw), p.iw + p.nwqrow), hi = std::min(std::max(L->row_hi, p.iw), p.iw + p.nwqrow);
  *a_lo = lo - p.iw;
  *a_hi = std::max(hi, lo) - p.iw;
}

// cm != nullptr (more than one rank; the caller has made every rank's `rng` yield the same bytes): every rank lays out the whole
// host image (it is small next to the encoded tableau), keeps and encodes its row slab, and the columns are committed as in
// lfgpu_ligero_commit_sharded (ligero.hip): one all_to_all of the encoded columns, local leaves, all_gather of the digests, the
// tree on every rank
int lig256_commit(lfgpu_ctx* c, const lfgpu_ligero_param& p, const E* W, const size_t* lqc, lfgpu_rng_fn rng, void* user, uint8_t root[32], Lig256** out,
                  const lfgpu_comm_ops* cm = nullptr) {
  if (p.ildt != 0 || p.idot != 1 || p.iquad != 2 || p.iw != 3) return lf_fail(c, LFGPU_ERR_ARG, "ligero256: row order");
  std::unique_ptr<Lig256> L(new Lig256());
  L->c = c;
  L->p = p;
  L->nonces.resize(32 * p.block_ext);
  L->row_lo = 0;
  L->row_hi = p.nrow;
  const int world = cm ? cm->world : 1, rank = cm ? cm->rank : 0;
  if (cm && world > 1) {
    L->sharded = true;
    L->comm = *cm;
    L->spans.resize(world);
    for (int q = 0; q < world; ++q) (void)lfgpu_ligero_row_shard(&p, q, world, &L->spans[q].first, &L->spans[q].second);
    L->row_lo = L->spans[rank].first;
    L->row_hi = L->spans[rank].second;
  }
  std::vector<E> H(p.nrow * p.dblock);
  LF_SCRUB_ON_EXIT(H);
  char err[256] = {0};
  const int rc = lig256_layout(p, W, lqc, rng, user, H.data(), L->nonces.data(), err, c->rng_exact != 0);
  if (rc) return lf_fail(c, rc, "%s", err);
  const size_t ld = p.block_enc, nr = L->row_hi - L->row_lo, ncols = p.block_ext;
  if (lf_pool_get(c, L->slab_bytes(), (void**)&L->d_T) != LFGPU_OK || lf_pool_get(c, 2 * p.block_ext * 32, (void**)&L->d_layers) != LFGPU_OK)
    return lf_fail(c, LFGPU_ERR_NOMEM, "ligero256: tableau alloc");
  if (nr) LF_HIP(c, hipMemcpy2DAsync(L->d_T, ld * 32, H.data() + L->row_lo * p.dblock, p.dblock * 32, p.dblock * 32, nr, hipMemcpyHostToDevice, c->stream));
  // rows IDOT / IQUAD carry dblock values, every other row block (ligero_prover.h:175,184,203,210,237): the slab's share of each group
  auto encode = [&](size_t g_lo, size_t g_hi, size_t n) -> int {
    const size_t lo = std::max(g_lo, L->row_lo), hi = std::min(g_hi, L->row_hi);
    return hi > lo ? lfgpu_fp256_rs_encode_rows(c, hi - lo, n, p.block_enc, L->row(lo), ld) : LFGPU_OK;
  };
  LF_TRY(encode(0, 1, p.block));
  LF_TRY(encode(1, 3, p.dblock));
  LF_TRY(encode(3, p.nrow, p.block));
  void* d_non = nullptr;
  if (!L->sharded) {
    LF_TRY(lf_scratch3(c, p.block_ext * 32, &d_non));
    LF_HIP(c, hipMemcpyAsync(d_non, L->nonces.data(), p.block_ext * 32, hipMemcpyHostToDevice, c->stream));
    LF_TRY(lfgpu_column_commit(c, LFGPU_FIELD_P256, p.nrow, ld, p.dblock, p.block_ext, L->d_T, d_non, L->d_layers, root));
    LF_HIP(c, hipStreamSynchronize(c->stream));  // H is a local
    *out = L.release();
    return LFGPU_OK;
  }
  // ---- sharded column commit
  auto split = [&](size_t n, int q, size_t* start, size_t* count) {
    const size_t base = n / (size_t)world, rem = n % (size_t)world, r = (size_t)q;
    *start = r * base + std::min(r, rem);
    *count = base + (r < rem ? 1 : 0);
  };
  std::vector<size_t> soff(world), sbytes(world), roff(world), rbytes(world);
  size_t mc0 = 0, mcn = 0, send_tot = 0, maxn = 0, dummy = 0;
  split(ncols, rank, &mc0, &mcn);
  split(ncols, 0, &dummy, &maxn);
  for (int q = 0; q < world; ++q) {
    size_t c0, cn;
    split(ncols, q, &c0, &cn);
    soff[q] = send_tot;
    sbytes[q] = nr * cn * 32;
    send_tot += sbytes[q];
    roff[q] = L->spans[q].first * mcn * 32;
    rbytes[q] = (L->spans[q].second - L->spans[q].first) * mcn * 32;
  }
  const size_t send_bytes = std::max<size_t>(send_tot, 32), cols_bytes = std::max<size_t>(p.nrow * mcn * 32, 32);
  void *d_send = nullptr, *d_cols = nullptr;
  if (lf_pool_get(c, send_bytes, &d_send) != LFGPU_OK) return lf_fail(c, LFGPU_ERR_NOMEM, "ligero256 (sharded): exchange buffers");
  if (lf_pool_get(c, cols_bytes, &d_cols) != LFGPU_OK) {
    lf_pool_put(c, d_send, send_bytes);
    return lf_fail(c, LFGPU_ERR_NOMEM, "ligero256 (sharded): exchange buffers");
  }
  auto done = [&](int code) {  // encoded rows of the tableau: scrubbed like the tableau before they go back to the pool
    (void)hipMemsetAsync(d_send, 0, send_bytes, c->stream);
    (void)hipMemsetAsync(d_cols, 0, cols_bytes, c->stream);
    lf_pool_put(c, d_send, send_bytes);
    lf_pool_put(c, d_cols, cols_bytes);
    return code;
  };
  for (int q = 0; q < world && nr; ++q) {
    size_t c0, cn;
    split(ncols, q, &c0, &cn);
    if (cn && hipMemcpy2DAsync((uint8_t*)d_send + soff[q], cn * 32, L->d_T + p.dblock + c0, ld * 32, cn * 32, nr, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
      return done(lf_fail(c, LFGPU_ERR_HIP, "ligero256 (sharded): pack"));
  }
  if (cm->all_to_all(cm->user, d_send, soff.data(), sbytes.data(), d_cols, roff.data(), rbytes.data(), 1, c->stream))
    return done(lf_fail(c, LFGPU_ERR_HIP, "ligero256 (sharded): all_to_all hook failed"));
  int rc2 = lf_scratch3(c, ncols * 32 + (size_t)(world + 1) * maxn * 32 + 64, &d_non);
  if (rc2) return done(rc2);
  uint8_t* d_mine = (uint8_t*)d_non + ncols * 32;
  uint8_t* d_all = d_mine + maxn * 32;
  if (hipMemcpyAsync(d_non, L->nonces.data(), ncols * 32, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipMemsetAsync(d_mine, 0, maxn * 32, c->stream) != hipSuccess)
    return done(lf_fail(c, LFGPU_ERR_HIP, "ligero256 (sharded): nonce upload"));
  if (mcn && (rc2 = lf_column_leaves32(c, p.nrow, mcn, 0, mcn, d_cols, (const uint8_t*)d_non + mc0 * 32, d_mine, 0))) return done(rc2);
  if (cm->all_gather(cm->user, d_mine, d_all, maxn * 32, 1, c->stream)) return done(lf_fail(c, LFGPU_ERR_HIP, "ligero256 (sharded): all_gather hook failed"));
  if (hipMemsetAsync(L->d_layers, 0, ncols * 32, c->stream) != hipSuccess) return done(lf_fail(c, LFGPU_ERR_HIP, "ligero256 (sharded): layers"));
  for (int q = 0; q < world; ++q) {
    size_t c0, cn;
    split(ncols, q, &c0, &cn);
    if (cn && hipMemcpyAsync(L->d_layers + (ncols + c0) * 32, d_all + (size_t)q * maxn * 32, cn * 32, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
      return done(lf_fail(c, LFGPU_ERR_HIP, "ligero256 (sharded): leaves"));
  }
  if ((rc2 = lfgpu_merkle_build_tree(c, ncols, L->d_layers, root))) return done(rc2);  // synchronises: H may go
  done(LFGPU_OK);
  *out = L.release();
  return LFGPU_OK;
}

int lig256_low_degree(Lig256* L, const E* u, E* y) {  // low_degree_proof (:281-291)
  lfgpu_ctx* c = L->c;
  const lfgpu_ligero_param& p = L->p;
  void *dy = nullptr, *du = nullptr;
  LF_TRY(lf_scratch3(c, p.block * 32, &dy));
  LF_TRY(lf_scratch2(c, p.nwqrow * 32 + 64, &du));
  if (L->has(p.ildt)) LF_HIP(c, hipMemcpyAsync(dy, L->row(p.ildt), p.block * 32, hipMemcpyDeviceToDevice, c->stream));
  else LF_HIP(c, hipMemsetAsync(dy, 0, p.block * 32, c->stream));
  LF_HIP(c, hipMemcpyAsync(du, u, p.nwqrow * 32, hipMemcpyHostToDevice, c->stream));
  size_t a_lo, a_hi;
  owned_wq256(L, &a_lo, &a_hi);
  if (a_hi > a_lo)
    hipLaunchKernelGGL(rows_axpy256_kernel, dim3(nblk(p.block, 64)), dim3(256), 0, c->stream, (u32)(a_hi - a_lo), p.block, (E*)dy, (const E*)du + a_lo,
                       (const E*)L->row(p.iw + a_lo), p.block_enc);
  LF_HIP(c, hipGetLastError());
  LF_TRY(lfgpu_memcpy_d2h(c, y, dy, p.block * 32));
  return comm_fold256(L, y, p.block);
}

// dot_proof (:293-309) with A given as inner_product_vector builds it: a dense block scale * dense[0..ndense) over the first
// flat positions plus sparse (index, value) terms (strictly increasing indices)
int lig256_dot(Lig256* L, const E* d_dense, size_t ndense, const E& scale, const uint64_t* idx, const E* val, size_t nsparse, E* y) {
  lfgpu_ctx* c = L->c;
  const lfgpu_ligero_param& p = L->p;
  const size_t lda = p.dblock;
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, (p.nwqrow * lda + 2 * p.dblock) * 32 + 64, &sc));
  E* dA = (E*)sc;
  E* dy = dA + p.nwqrow * lda;
  LF_HIP(c, hipMemsetAsync(dA, 0, p.nwqrow * lda * 32, c->stream));
  if (ndense) hipLaunchKernelGGL(a_rows_dense256_kernel, dim3(nblk(ndense)), dim3(Z_THREADS), 0, c->stream, (u32)p.r, (u32)p.w, lda, scale, d_dense, ndense, dA);
  if (nsparse) {
    void* d_sp = nullptr;
    LF_TRY(lf_scratch2(c, nsparse * 40 + 64, &d_sp));
    E* d_val = (E*)d_sp;
    u64* d_idx = (u64*)(d_val + nsparse);
    LF_HIP(c, hipMemcpyAsync(d_val, val, nsparse * 32, hipMemcpyHostToDevice, c->stream));
    LF_HIP(c, hipMemcpyAsync(d_idx, idx, nsparse * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(a_rows_sparse256_kernel, dim3(nblk(nsparse)), dim3(Z_THREADS), 0, c->stream, (u32)p.r, (u32)p.w, lda, (const u64*)d_idx, (const E*)d_val,
                       nsparse, dA);
    LF_HIP(c, hipGetLastError());
    LF_HIP(c, hipStreamSynchronize(c->stream));  // the staging area is the Reed-Solomon encoder's work space next
  }
  size_t a_lo, a_hi;
  owned_wq256(L, &a_lo, &a_hi);
  if (a_hi > a_lo) LF_TRY(lfgpu_fp256_rs_encode_rows(c, a_hi - a_lo, p.block, p.dblock, dA + a_lo * lda, lda));
  const E* T0 = nullptr;
  if (L->has(p.idot)) {
    T0 = L->row(p.idot);
  } else {  // a zero row behind dy
    LF_HIP(c, hipMemsetAsync(dy + p.dblock, 0, p.dblock * 32, c->stream));
    T0 = dy + p.dblock;
  }
  hipLaunchKernelGGL(rows_vaxpy256_kernel, dim3(nblk(p.dblock, 64)), dim3(256), 0, c->stream, (u32)(a_hi - a_lo), p.dblock, T0, (const E*)(dA + a_lo * lda), lda,
                     (const E*)(a_hi > a_lo ? L->row(p.iw + a_lo) : L->d_T), p.block_enc, dy);
  LF_HIP(c, hipGetLastError());
  LF_TRY(lfgpu_memcpy_d2h(c, y, dy, p.dblock * 32));
  return comm_fold256(L, y, p.dblock);
}

int lig256_quadratic(Lig256* L, const E* u, E* y0, E* y2) {  // quadratic_proof (:311-344)
  lfgpu_ctx* c = L->c;
  const lfgpu_ligero_param& p = L->p;
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, (p.nqtriples + 1 + 2 * p.dblock) * 32 + 64, &sc));
  E* du = (E*)sc;
  E* dy = du + p.nqtriples + 1;
  if (p.nqtriples) LF_HIP(c, hipMemcpyAsync(du, u, p.nqtriples * 32, hipMemcpyHostToDevice, c->stream));
  const size_t ld = p.block_enc;
  // a triple is multiplied element-wise: the slab holds all quadratic rows (the last rank) or none
  const size_t q_lo = p.iq, q_hi = p.iq + 3 * p.nqtriples;
  const bool all_q = p.nqtriples == 0 || (L->row_lo <= q_lo && q_hi <= L->row_hi);
  if (!all_q && !(L->row_hi <= q_lo || L->row_lo >= q_hi)) return lf_fail(c, LFGPU_ERR_ARG, "quadratic_proof256: the slab splits the quadratic rows");
  const size_t nt = all_q ? p.nqtriples : 0;
  const E* Tq = nullptr;
  if (L->has(p.iquad)) {
    Tq = L->row(p.iquad);
  } else {
    LF_HIP(c, hipMemsetAsync(dy + p.dblock, 0, p.dblock * 32, c->stream));
    Tq = dy + p.dblock;
  }
  const E* X = nt ? L->row(p.iq) : L->d_T;
  const E* Y = X + p.nqtriples * ld;
  const E* Zr = Y + p.nqtriples * ld;
  hipLaunchKernelGGL(quad_combo256_kernel, dim3(nblk(p.dblock)), dim3(Z_THREADS), 0, c->stream, (u32)nt, p.dblock, Tq, (const E*)du, X, Y, Zr, ld, dy);
  LF_HIP(c, hipGetLastError());
  std::vector<E> y(p.dblock);
  LF_TRY(lfgpu_memcpy_d2h(c, y.data(), dy, p.dblock * 32));
  LF_TRY(comm_fold256(L, y.data(), p.dblock));
  for (size_t j = 0; j < p.w; ++j)
    if (!e32_is_zero(y[p.r + j])) return lf_fail(c, LFGPU_ERR_ASSERT, "quadratic_proof: W part is nonzero");
  memcpy(y0, y.data(), p.r * 32);
  memcpy(y2, y.data() + p.block, (p.dblock - p.block) * 32);
  return LFGPU_OK;
}

int lig256_open(Lig256* L, const size_t* idx, E* req, uint8_t* nonces, uint8_t* path, size_t path_cap, size_t* npath) {  // compute_req + MerkleCommitment::open
  lfgpu_ctx* c = L->c;
  const lfgpu_ligero_param& p = L->p;
  for (size_t i = 0; i < p.nreq; ++i)
    if (idx[i] >= p.block_ext) return lf_fail(c, LFGPU_ERR_ARG, "ligero_open: index out of range");
  void *dreq = nullptr, *di = nullptr;
  const size_t nr = L->row_hi - L->row_lo;
  size_t maxr = nr;
  for (const auto& sp : L->spans) maxr = std::max(maxr, sp.second - sp.first);
  LF_TRY(lf_scratch3(c, std::max<size_t>(nr, 1) * p.nreq * 32, &dreq));
  LF_TRY(lf_scratch2(c, p.nreq * 8 + 64, &di));
  std::vector<u64> ix(idx, idx + p.nreq);
  LF_HIP(c, hipMemcpyAsync(di, ix.data(), p.nreq * 8, hipMemcpyHostToDevice, c->stream));
  std::vector<uint8_t> mine;  // a slab's rows of req, padded to the largest slab for the all_gather
  void* h_dst = req;
  if (L->sharded) {
    mine.assign(maxr * p.nreq * 32, 0);
    h_dst = mine.data();
  }
  if (nr) {
    hipLaunchKernelGGL(gather_columns256_kernel, dim3(nblk(nr * p.nreq)), dim3(Z_THREADS), 0, c->stream, (u32)nr, p.block_enc, p.dblock, (const E*)L->d_T,
                       (const u64*)di, (u32)p.nreq, (E*)dreq);
    LF_HIP(c, hipGetLastError());
    LF_TRY(lfgpu_memcpy_d2h(c, h_dst, dreq, nr * p.nreq * 32));
  } else {
    LF_HIP(c, hipStreamSynchronize(c->stream));  // ix is a local
  }
  if (L->sharded) {
    const lfgpu_comm_ops& cm = L->comm;
    std::vector<uint8_t> all((size_t)cm.world * mine.size());
    if (cm.all_gather(cm.user, mine.data(), all.data(), mine.size(), 0, nullptr)) return lf_fail(c, LFGPU_ERR_HIP, "ligero256_open (sharded): all_gather hook failed");
    for (int q = 0; q < cm.world; ++q)
      memcpy((uint8_t*)req + L->spans[q].first * p.nreq * 32, all.data() + (size_t)q * mine.size(), (L->spans[q].second - L->spans[q].first) * p.nreq * 32);
  }
  for (size_t i = 0; i < p.nreq; ++i) memcpy(nonces + 32 * i, &L->nonces[32 * idx[i]], 32);
  return lfgpu_merkle_open(c, p.block_ext, L->d_layers, idx, p.nreq, path, path_cap, npath);
}
}  // namespace

// ------------------------------------------------------------------ ZkProver<Fp256Base>
struct Zk256 {
  lfgpu_ctx* c = nullptr;
  const lfgpu_circuit* C = nullptr;
  lfgpu_ligero_param param{};
  size_t npub = 0, n_witness = 0, pad_size = 0;
  struct LayerPad {  // Proof-shaped pad (zk_prover.h:152-188): hp[hand][2 round + {0, 1}] = {p(0), p(2)}, wc[2]
    std::vector<E> hp[2];
    E wc[2];
  };
  std::vector<LayerPad> pad, proof;
  std::vector<E> aux;  // ProofAux::bound_quad per layer
  std::vector<size_t> lqc;
  Lig256* lp = nullptr;
  uint8_t root[32] = {0};
  std::vector<E> y_ldt, y_dot, y_q0, y_q2, req;
  std::vector<uint8_t> nonces, path;
  size_t npath = 0;
  bool have_proof = false;
  mutable std::vector<uint8_t> wire;  // ZkProof::write bytes of the held proof (zk256_proof_write fills it once)
  mutable bool wire_valid = false;
  std::vector<void*> d_in;  // the layers' inputs (eval_circuit), resident for the sumcheck
  void* d_V = nullptr;
  void* h_V = nullptr;  // pinned: outputs then the assert-zero flag
  void* d_eq = nullptr;  // EQ table of the input constraint
  double ms[6] = {0, 0, 0, 0, 0, 0};
  // lfgpu_zk_prover_set_comm: a tableau of at least comm_min_bytes is committed with its rows sharded over the communicator's GPUs
  bool have_comm = false;
  lfgpu_comm_ops comm{};
  size_t comm_min_bytes = 0;
  ~Zk256() {
    delete lp;
    // the layers' wire values are functions of the witness: scrubbed before the memory goes back to the allocator (as the
    // tableau is, Lig256), and so are the host copies of the pads
    for (size_t l = 0; l < d_in.size(); ++l)
      if (d_in[l]) (void)hipMemsetAsync(d_in[l], 0, C->layers[l].nw * 32, c->stream);
    if (d_V) (void)hipMemsetAsync(d_V, 0, C->info.nv * 32, c->stream);
    if (d_eq) (void)hipMemsetAsync(d_eq, 0, C->info.ninputs * 32, c->stream);
    (void)hipStreamSynchronize(c->stream);
    for (void* p : d_in)
      if (p) (void)hipFree(p);
    if (d_V) (void)hipFree(d_V);
    if (d_eq) (void)hipFree(d_eq);
    if (h_V) {
      memset(h_V, 0, C->info.nv * 32 + 16);
      (void)hipHostFree(h_V);
    }
    for (auto& P : pad) {
      for (auto& v : P.hp) std::fill(v.begin(), v.end(), E{});
      P.wc[0] = P.wc[1] = E{};
    }
  }
};

namespace {
constexpr size_t kMaxBindings256 = 40;  // Proof::kMaxBindings (lib/sumcheck/circuit.h:84)
inline size_t layer_size256(size_t logw) { return 4 * logw + 3; }  // PadLayout::layer_size (zk_common.h:210-222)

struct Ts256 {  // the caller's transcript seen through the hooks
  const lfgpu_transcript_ops* o;
  void* u;
  void write_bytes(const uint8_t* d, size_t n) const { o->write_bytes(u, d, n); }
  void write_elt(const E& e) const {
    uint8_t b[32];
    h256_to_bytes(e, b);
    o->write_elt_sized(u, b, 32);
  }
  void write_array(const E* e, size_t n) const {
    std::vector<uint8_t> b(32 * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) h256_to_bytes(e[i], &b[32 * i]);
    o->write_elt_array_sized(u, b.data(), n, 32);
  }
  E elt() const {
    return h256_sample([&](uint8_t* b, size_t n) { o->gen_bytes(u, b, n); });
  }
  size_t nat(size_t n) const {  // RandomEngine::nat (lib/random/random.h:57-87)
    size_t l = 0, mask = 0;
    for (size_t nn = n; nn; nn >>= 8) ++l;
    while ((n & mask) != n) mask = (mask << 1) | 1;
    for (;;) {
      uint8_t b[8] = {0};
      o->gen_bytes(u, b, l);
      size_t r = 0;
      for (size_t i = 0; i < l; ++i) r |= (size_t)b[i] << (8 * i);
      r &= mask;
      if (r < n) return r;
    }
  }
  void choose(size_t n, size_t k, size_t* res) const {  // RandomEngine::choose (:89-105)
    std::vector<size_t> A(n);
    for (size_t i = 0; i < n; ++i) A[i] = i;
    for (size_t i = 0; i < k; ++i) {
      const size_t j = i + nat(n - i);
      std::swap(A[i], A[j]);
      res[i] = A[i];
    }
  }
};

struct Round256 {  // round_h of the padded prover (prover_layers.h:320-329): transmit poly - pad
  const Ts256* tst;
  const Zk256::LayerPad* pad;
  Zk256::LayerPad* out;
};
void zk256_round_cb(void* user, size_t hand, size_t rnd, const E ev[3], E* chal) {
  Round256* r = (Round256*)user;
  const E t0 = fp256_sub(ev[0], r->pad->hp[hand][2 * rnd]), t2 = fp256_sub(ev[2], r->pad->hp[hand][2 * rnd + 1]);
  r->out->hp[hand][2 * rnd] = t0;
  r->out->hp[hand][2 * rnd + 1] = t2;
  r->tst->write_elt(t0);
  r->tst->write_elt(t2);
  *chal = r->tst->elt();
}

// ZkCommon::verifier_constraints (zk_common.h:49-136) + input_constraint (:406-439) on the prover's side (aux = the bound
// quads the sumcheck recorded): the sparse rows of A, b, and the EQ vector of the input constraint on the device
struct LinTerm256 {
  size_t c, w;
  E k;
};
struct Constraints256 {
  std::vector<LinTerm256> a;
  std::vector<E> b;
  size_t n = 0;
};
int bind_gh_all256(lfgpu_quad* q, const F256& F, size_t logv, const E* G0, const E* G1, const E& alpha, const E& beta, size_t logw, size_t nw, const E* H0,
                   const E* H1, E* out) {
  lfgpu_ctx* c = q->c;
  if (logv > 40 || logw > 40 || ((size_t)1 << logv) < q->nv || nw == 0 || ((size_t)1 << logw) < nw || nw <= q->hmax)
    return lf_fail(c, LFGPU_ERR_ARG, "bind_gh_all256: table sizes (nw must exceed the largest hand index %zu)", q->hmax);
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, (q->nv + 2 * nw) * 32 + 128, &sc));
  E* d_eqg = (E*)sc;
  E* d_eqh0 = d_eqg + q->nv;
  E* d_eqh1 = d_eqh0 + nw;
  u64* d_acc = (u64*)(d_eqh1 + nw);
  LF_TRY(raw_eq2_256(c, F, logv, q->nv, G0, G1, alpha, d_eqg));
  LF_TRY(raw_eq2_256(c, F, logw, nw, H0, H0, F.zero, d_eqh0));  // EQ(H0, i) + 0 * (...)
  LF_TRY(raw_eq2_256(c, F, logw, nw, H1, H1, F.zero, d_eqh1));
  LF_HIP(c, hipMemsetAsync(d_acc, 0, 64, c->stream));
  hipLaunchKernelGGL(bind_gh_all256_kernel, dim3(std::min<u32>(nblk(q->n), 1024)), dim3(Z_THREADS), 0, c->stream, q->n, (const corner4*)q->d_morton,
                     (const E*)q->d_kvec, (const E*)d_eqg, (const E*)d_eqh0, (const E*)d_eqh1, beta, d_acc);
  LF_HIP(c, hipGetLastError());
  u64 w[8];
  LF_TRY(lfgpu_memcpy_d2h(c, w, d_acc, 64));
  *out = fp256_reduce_limbs(w, F.rsq);
  return LFGPU_OK;
}

// aux = the bound quads the prover's sumcheck recorded, or nullptr (verifier): Quad::bind_gh_all on the device per layer
int build_constraints256(lfgpu_ctx* c, const lfgpu_circuit* C, const F256& F, const Ts256& ts, const std::vector<Zk256::LayerPad>& proof,
                         const std::vector<E>* aux, const E* pub, E* d_eq, Constraints256& out) {
  const lfgpu_circuit_info& I = C->info;
  const size_t nl = C->layers.size(), npub = I.npub_in;
  std::vector<E> gh[2], G[2];
  for (size_t i = 0; i < kMaxBindings256; ++i) (void)ts.elt();  // begin_circuit: Q (unused for logc = 0), then G
  G[0].resize(kMaxBindings256);
  for (size_t i = 0; i < kMaxBindings256; ++i) G[0][i] = ts.elt();
  G[1] = G[0];
  size_t logv = I.logv;
  size_t ci = 0, pi = I.ninputs - npub;
  E claims[2] = {F.zero, F.zero};
  std::vector<E> sym;
  for (size_t ly = 0; ly < nl; ++ly) {
    const size_t logw = C->layers[ly].logw;
    const E alpha = ts.elt(), beta = ts.elt();
    const size_t n = 3 + layer_size256(logw);
    E known = F.zero;
    sym.assign(n, F.zero);
    auto axpy = [&](size_t var, const E& kv, const E& k) {  // Expression::axpy
      known = F.add(known, F.mul(k, kv));
      sym[var] = F.add(sym[var], k);
    };
    auto axmy = [&](size_t var, const E& kv, const E& k) {  // Expression::axmy
      known = F.sub(known, F.mul(k, kv));
      sym[var] = F.sub(sym[var], k);
    };
    axpy(0, claims[0], F.one);  // ConstraintBuilder::first
    axpy(1, claims[1], alpha);
    gh[0].assign(logw ? logw : 1, F.zero);
    gh[1].assign(logw ? logw : 1, F.zero);
    const auto& P = proof[ly];
    for (size_t rnd = 0; rnd < logw; ++rnd)
      for (int hand = 0; hand < 2; ++hand) {
        const size_t r = 2 * rnd + hand;
        const E t0e = P.hp[hand][2 * rnd], t2e = P.hp[hand][2 * rnd + 1];
        ts.write_elt(t0e);
        ts.write_elt(t2e);
        const E chal = ts.elt();
        gh[hand][rnd] = chal;
        E lag[3];  // dot_interpolation: p(chal) = sum_i lag[i] p(P_i)
        for (int i = 0; i < 3; ++i) {
          E num = F.one;
          for (int j = 0; j < 3; ++j)
            if (j != i) num = F.mul(num, F.sub(chal, F.pts[j]));
          lag[i] = F.mul(num, F.invden[i]);
        }
        axmy(3 + 2 * r, t0e, F.one);   // ConstraintBuilder::next: p(1) = claim - p(0)
        known = F.mul(known, lag[1]);  // scale
        for (auto& s : sym)
          if (!e32_is_zero(s)) s = F.mul(s, lag[1]);
        axpy(3 + 2 * r, t0e, lag[0]);
        axpy(3 + 2 * r + 1, t2e, lag[2]);
      }
    E eqq;  // EQ[Q,C] QUAD[R,L] (Eq::eval with logc = 0 is 1)
    if (aux) eqq = (*aux)[ly];
    else LF_TRY(bind_gh_all256(C->layers[ly].q, F, logv, G[0].data(), G[1].data(), alpha, beta, logw, C->layers[ly].nw, gh[0].data(), gh[1].data(), &eqq));
    const size_t cp = 3 + 4 * logw, skip = ly == 0 ? 3 : 0;  // ConstraintBuilder::finalize
    out.b.push_back(F.sub(F.mul(eqq, F.mul(P.wc[0], P.wc[1])), known));
    sym[cp] = F.sub(sym[cp], F.mul(eqq, P.wc[1]));
    sym[cp + 1] = F.sub(sym[cp + 1], F.mul(eqq, P.wc[0]));
    sym[cp + 2] = F.sub(sym[cp + 2], eqq);
    for (size_t i = skip; i < n; ++i) out.a.push_back({ci, pi + i - 3, sym[i]});
    ++ci;
    ts.write_array(P.wc, 2);
    claims[0] = P.wc[0];
    claims[1] = P.wc[1];
    for (int h = 0; h < 2; ++h) {
      G[h].assign(kMaxBindings256, F.zero);
      for (size_t r = 0; r < logw; ++r) G[h][r] = gh[h][r];
    }
    logv = logw;
    pi += layer_size256(logw);
  }
  const E alpha = ts.elt();
  out.a.push_back({ci, pi - 3, F.sub(F.zero, F.one)});  // input_constraint: -1, -alpha on the input layer's claim pads
  out.a.push_back({ci, pi - 2, F.sub(F.zero, alpha)});
  out.n = ci + 1;
  const size_t logn = C->layers[nl - 1].logw;
  LF_TRY(raw_eq2_256(c, F, logn, I.ninputs, gh[0].data(), gh[1].data(), alpha, d_eq));
  std::vector<E> eq_in(npub ? npub : 1);
  if (npub) LF_TRY(lfgpu_memcpy_d2h(c, eq_in.data(), d_eq, npub * 32));
  const auto& P = proof[nl - 1];
  E pub_binding = F.zero;
  for (size_t i = 0; i < npub; ++i) pub_binding = F.add(pub_binding, F.mul(eq_in[i], pub[i]));
  out.b.push_back(F.sub(F.add(P.wc[0], F.mul(alpha, P.wc[1])), pub_binding));
  return LFGPU_OK;
}

// LigeroCommon::inner_product_vector (ligero_param.h:382-421), host share: the sparse terms of A as (flat index, value),
// sorted with duplicates folded (the dense private-input block is built on the device)
void inner_product_sparse256(const F256& F, const lfgpu_ligero_param& p, const Constraints256& cs, const std::vector<E>& alphal, const std::vector<size_t>& lqc,
                             const std::vector<E>& alphaq, std::vector<uint64_t>& idx, std::vector<E>& val) {
  std::vector<std::pair<uint64_t, E>> t;
  t.reserve(cs.a.size() + 6 * p.nq);
  for (const LinTerm256& l : cs.a) t.emplace_back((uint64_t)l.w, F.mul(l.k, alphal[l.c]));
  const size_t base = p.nwrow * p.w;
  const size_t Ax = base, Ay = base + p.nqtriples * p.w, Az = base + 2 * p.nqtriples * p.w;
  for (size_t iw = 0; iw < p.nq; ++iw) {
    const size_t off[3] = {Ax + iw, Ay + iw, Az + iw};
    for (int j = 0; j < 3; ++j) {
      const E aq = alphaq[3 * iw + j];
      t.emplace_back((uint64_t)off[j], aq);
      t.emplace_back((uint64_t)lqc[3 * iw + j], F.sub(F.zero, aq));
    }
  }
  std::stable_sort(t.begin(), t.end(), [](const std::pair<uint64_t, E>& a, const std::pair<uint64_t, E>& b) { return a.first < b.first; });
  idx.clear();
  val.clear();
  for (const auto& e : t) {
    if (!idx.empty() && idx.back() == e.first) val.back() = F.add(val.back(), e.second);
    else {
      idx.push_back(e.first);
      val.push_back(e.second);
    }
  }
}
}  // namespace

int zk256_new(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, Zk256** out) {
  std::unique_ptr<Zk256> z(new Zk256());
  z->c = c;
  z->C = C;
  z->npub = C->info.npub_in;
  z->n_witness = C->info.ninputs - C->info.npub_in;
  for (const auto& l : C->layers) z->pad_size += layer_size256(l.logw);
  LF_TRY(lfgpu_ligero_param_init(&z->param, LFGPU_FIELD_P256, 0, z->n_witness + z->pad_size, C->info.nl, rateinv, nreq, block_enc));
  LF_HIP(c, hipSetDevice(c->device));
  z->d_in.assign(C->layers.size(), nullptr);
  for (size_t l = 0; l < C->layers.size(); ++l)
    if (hipMalloc(&z->d_in[l], C->layers[l].nw * 32) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "zk256: layer %zu inputs", l);
  if (hipMalloc(&z->d_V, C->info.nv * 32) != hipSuccess || hipMalloc(&z->d_eq, C->info.ninputs * 32) != hipSuccess)
    return lf_fail(c, LFGPU_ERR_NOMEM, "zk256: outputs");
  if (hipHostMalloc(&z->h_V, C->info.nv * 32 + 16, hipHostMallocDefault) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "zk256: pinned outputs");
  *out = z.release();
  return LFGPU_OK;
}
int zk256_param(const Zk256* z, lfgpu_ligero_param* p) {
  *p = z->param;
  return LFGPU_OK;
}
void zk256_set_comm(Zk256* z, const lfgpu_comm_ops* comm, size_t min_tableau_bytes) {
  z->have_comm = comm != nullptr;
  if (comm) z->comm = *comm;
  z->comm_min_bytes = min_tableau_bytes;
}
void zk256_free(Zk256* z) { delete z; }
int zk256_timings(const Zk256* z, double ms[6]) {
  memcpy(ms, z->ms, sizeof(z->ms));
  return LFGPU_OK;
}

static int ts256_ok(lfgpu_ctx* c, const lfgpu_transcript_ops* ts) {
  if (!ts->write_elt_sized || !ts->write_elt_array_sized)
    return lf_fail(c, LFGPU_ERR_ARG, "zk256: the transcript hooks lack write_elt_sized / write_elt_array_sized (32-byte elements)");
  return LFGPU_OK;
}

// ZkProver::commit (zk_prover.h:72-96): fill_pad from rng, Ligero-commit witness || pad, root -> transcript
// draws_only: perform every RandomEngine draw of the commit (pads, then the Ligero layout) and stop -- rank 0 of a communicator
// records its engine's stream this way before any collective runs
int zk256_commit(Zk256* z, const void* h_W, lfgpu_rng_fn rng, void* rng_user, const lfgpu_transcript_ops* ts, uint8_t root_out[32], bool draws_only) {
  lfgpu_ctx* c = z->c;
  LF_TRY(ts256_ok(c, ts));
  const double t0 = now_ms();
  const lfgpu_circuit* C = z->C;
  const size_t nl = C->layers.size();
  LF_HIP(c, hipSetDevice(c->device));
  std::vector<E> pads(z->pad_size - nl);  // every element fill_pad draws (the product wc0 * wc1 is computed), in order
  LF_SCRUB_ON_EXIT(pads);
  if (c->rng_exact) for (size_t i = 0; i < pads.size(); ++i) h256_sample_many(&pads[i], 1, [&](uint8_t* b, size_t n) { rng(rng_user, b, n); });
  else h256_sample_many(pads.data(), pads.size(), [&](uint8_t* b, size_t n) { rng(rng_user, b, n); });
  size_t pd = 0;
  auto draw = [&] { return pads[pd++]; };
  std::vector<E> Wv(z->param.nw);  // witness || pads
  LF_SCRUB_ON_EXIT(Wv);
  memcpy(Wv.data(), (const E*)h_W + z->npub, z->n_witness * 32);
  z->pad.assign(nl, {});
  z->lqc.assign(3 * nl, 0);
  size_t pi = z->n_witness;
  for (size_t ly = 0; ly < nl; ++ly) {  // fill_pad (zk_prover.h:152-188, logc = 0)
    const size_t logw = C->layers[ly].logw;
    auto& P = z->pad[ly];
    P.hp[0].resize(2 * logw);
    P.hp[1].resize(2 * logw);
    size_t w = pi;
    for (size_t j = 0; j < logw; ++j)
      for (int h = 0; h < 2; ++h) {
        P.hp[h][2 * j] = draw();
        P.hp[h][2 * j + 1] = draw();
        Wv[w++] = P.hp[h][2 * j];
        Wv[w++] = P.hp[h][2 * j + 1];
      }
    P.wc[0] = draw();
    P.wc[1] = draw();
    Wv[w++] = P.wc[0];
    Wv[w++] = P.wc[1];
    Wv[w++] = fp256_mul(P.wc[0], P.wc[1]);
    const size_t cp = pi + 4 * logw;  // setup_lqc (zk_common.h:149-160)
    z->lqc[3 * ly] = cp;
    z->lqc[3 * ly + 1] = cp + 1;
    z->lqc[3 * ly + 2] = cp + 2;
    pi += layer_size256(logw);
  }
  if (pi != z->param.nw) return lf_fail(c, LFGPU_ERR_ASSERT, "zk256_commit: witness layout");
  if (draws_only) {
    std::vector<E> H(z->param.nrow * z->param.dblock);
    LF_SCRUB_ON_EXIT(H);
    std::vector<uint8_t> nz(32 * z->param.block_ext);
    char err[256] = {0};
    const int rc = lig256_layout(z->param, Wv.data(), z->lqc.data(), rng, rng_user, H.data(), nz.data(), err, c->rng_exact != 0);
    return rc ? lf_fail(c, rc, "%s", err) : LFGPU_OK;
  }
  delete z->lp;
  z->lp = nullptr;
  z->have_proof = false;
  z->wire_valid = false;
  const bool shard_rows = z->have_comm && z->comm.world > 1 && z->param.nrow * z->param.block_enc * 32 >= z->comm_min_bytes;
  LF_TRY(lig256_commit(c, z->param, Wv.data(), z->lqc.data(), rng, rng_user, z->root, &z->lp, shard_rows ? &z->comm : nullptr));
  ts->write_bytes(ts->user, z->root, 32);  // LigeroTranscript::write_commitment
  if (root_out) memcpy(root_out, z->root, 32);
  z->ms[0] = now_ms() - t0;
  return LFGPU_OK;
}

// ZkProver::prove (zk_prover.h:98-149)
int zk256_prove(Zk256* z, const void* h_W, const lfgpu_transcript_ops* tso, int* ok) {
  lfgpu_ctx* c = z->c;
  LF_TRY(ts256_ok(c, tso));
  if (!z->lp) return lf_fail(c, LFGPU_ERR_ARG, "zk256_prove: must run commit before prove");
  const double t_start = now_ms();
  const lfgpu_circuit* C = z->C;
  const lfgpu_circuit_info& I = C->info;
  const size_t nl = C->layers.size();
  const E* W = (const E*)h_W;
  const F256 F;
  const Ts256 ts{tso, tso->user};
  *ok = 0;
  z->have_proof = false;
  z->wire_valid = false;
  LF_HIP(c, hipSetDevice(c->device));

  // eval_circuit (prover_layers.h:52-104): all layers back to back while the host hashes the Fiat-Shamir preamble
  double t0 = now_ms();
  const E* V = (const E*)z->h_V;
  const int* failed = (const int*)((const uint8_t*)z->h_V + I.nv * 32);
  {
    LF_HIP(c, hipMemcpyAsync(z->d_in[nl - 1], W, I.ninputs * 32, hipMemcpyHostToDevice, c->stream));
    int* d_fail = (int*)((uint8_t*)c->mailbox_d + 128);
    LF_HIP(c, hipMemsetAsync(d_fail, 0, 4, c->stream));
    for (size_t l = nl; l-- > 0;) LF_TRY(lf256_eval_quad_async(C->layers[l].q, z->d_in[l], l ? z->d_in[l - 1] : z->d_V, d_fail));
    LF_HIP(c, hipMemcpyAsync(z->h_V, z->d_V, I.nv * 32, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(c, hipMemcpyAsync((uint8_t*)z->h_V + I.nv * 32, d_fail, 4, hipMemcpyDeviceToHost, c->stream));
  }
  const double t_enq = now_ms() - t0;
  // initialize_sumcheck_fiat_shamir (zk_common.h:163-180)
  ts.write_bytes(I.id, 32);
  for (size_t i = 0; i < z->npub; ++i) ts.write_elt(W[i]);
  ts.write_elt(F.zero);
  ts.write_bytes(C->zeros->data(), I.nterms);
  void* cl = tso->clone(tso->user);
  if (!cl) {
    (void)hipStreamSynchronize(c->stream);
    return lf_fail(c, LFGPU_ERR_NOMEM, "zk256_prove: transcript clone");
  }
  struct CloneGuard {
    const lfgpu_transcript_ops* o;
    void* u;
    ~CloneGuard() { o->free_clone(u); }
  } cg{tso, cl};
  const Ts256 tst{tso, cl};
  t0 = now_ms();
  LF_HIP(c, hipStreamSynchronize(c->stream));
  if (*failed) return LFGPU_OK;  // an assert-zero term is non-zero: eval_circuit returns nullptr
  for (size_t i = 0; i < I.nv; ++i)
    if (!e32_is_zero(V[i])) return LFGPU_OK;
  z->ms[2] = t_enq + now_ms() - t0;

  // padded sumcheck (ProverLayers::prove with pad, on the transcript copy)
  t0 = now_ms();
  z->proof.assign(nl, {});
  z->aux.assign(nl, F.zero);
  std::vector<E> G[2];
  {
    for (size_t i = 0; i < kMaxBindings256; ++i) (void)tst.elt();  // begin_circuit: Q then G (transcript_sumcheck.h:49-52)
    G[0].resize(kMaxBindings256);
    for (size_t i = 0; i < kMaxBindings256; ++i) G[0][i] = tst.elt();
    G[1] = G[0];
  }
  size_t logv = I.logv;
  E WC[2] = {F.zero, F.zero};
  std::vector<E> gout;
  for (size_t ly = 0; ly < nl; ++ly) {
    const auto& L = C->layers[ly];
    const E alpha = tst.elt(), beta = tst.elt();
    auto& P = z->proof[ly];
    P.hp[0].resize(2 * L.logw);
    P.hp[1].resize(2 * L.logw);
    Round256 rc{&tst, &z->pad[ly], &P};
    gout.assign(2 * L.logw + 1, F.zero);
    E wc_out[2], bq;
    LF_TRY(sumcheck_layer256(L.q, F, logv, G[0].data(), G[1].data(), alpha, beta, L.logw, L.nw, (const E*)z->d_in[ly], WC, zk256_round_cb, &rc, wc_out,
                             gout.data(), &bq));
    P.wc[0] = F.sub(wc_out[0], z->pad[ly].wc[0]);  // end_layer (:331-344): transmit wc - pad
    P.wc[1] = F.sub(wc_out[1], z->pad[ly].wc[1]);
    tst.write_array(P.wc, 2);
    z->aux[ly] = bq;
    WC[0] = wc_out[0];
    WC[1] = wc_out[1];
    for (int h = 0; h < 2; ++h) {
      G[h].assign(kMaxBindings256, F.zero);
      for (size_t r = 0; r < L.logw; ++r) G[h][r] = gout[h * L.logw + r];
    }
    logv = L.logw;
  }
  z->ms[3] = now_ms() - t0;

  // verifier_constraints with aux: replay the verifier symbolically on the ORIGINAL transcript
  t0 = now_ms();
  Constraints256 cs;
  LF_TRY(build_constraints256(c, C, F, ts, z->proof, &z->aux, W, (E*)z->d_eq, cs));
  const lfgpu_ligero_param& p = z->param;
  z->ms[4] = now_ms() - t0;

  // LigeroProver::prove (ligero_prover.h:84-146)
  t0 = now_ms();
  {
    uint8_t hash_of_A[32] = {0xde, 0xad, 0xbe, 0xef};  // zk_prover.h:143
    ts.write_bytes(hash_of_A, 32);
    std::vector<E> u_ldt(p.nwqrow);
    for (auto& e : u_ldt) e = ts.elt();
    z->y_ldt.assign(p.block, F.zero);
    LF_TRY(lig256_low_degree(z->lp, u_ldt.data(), z->y_ldt.data()));
    std::vector<E> alphal(cs.n), alphaq(3 * p.nq);
    for (auto& e : alphal) e = ts.elt();
    for (auto& e : alphaq) e = ts.elt();
    std::vector<uint64_t> a_idx;
    std::vector<E> a_val;
    inner_product_sparse256(F, p, cs, alphal, z->lqc, alphaq, a_idx, a_val);
    z->y_dot.assign(p.dblock, F.zero);
    LF_TRY(lig256_dot(z->lp, (const E*)z->d_eq + z->npub, z->n_witness, alphal[cs.n - 1], a_idx.data(), a_val.data(), a_idx.size(), z->y_dot.data()));
    std::vector<E> u_quad(p.nqtriples ? p.nqtriples : 1);
    for (size_t i = 0; i < p.nqtriples; ++i) u_quad[i] = ts.elt();
    z->y_q0.assign(p.r, F.zero);
    z->y_q2.assign(p.dblock - p.block, F.zero);
    LF_TRY(lig256_quadratic(z->lp, u_quad.data(), z->y_q0.data(), z->y_q2.data()));
    ts.write_array(z->y_ldt.data(), z->y_ldt.size());
    ts.write_array(z->y_dot.data(), z->y_dot.size());
    ts.write_array(z->y_q0.data(), z->y_q0.size());
    ts.write_array(z->y_q2.data(), z->y_q2.size());
    std::vector<size_t> idx(p.nreq);
    ts.choose(p.block_ext, p.nreq, idx.data());
    z->req.assign(p.nrow * p.nreq, F.zero);
    z->nonces.assign(p.nreq * 32, 0);
    const size_t cap = p.nreq * p.mc_pathlen + 1;
    z->path.assign(cap * 32, 0);
    LF_TRY(lig256_open(z->lp, idx.data(), z->req.data(), z->nonces.data(), z->path.data(), cap, &z->npath));
  }
  z->ms[5] = now_ms() - t0;
  z->ms[1] = now_ms() - t_start;
  z->have_proof = true;
  *ok = 1;
  return LFGPU_OK;
}

// ZkProof::write (zk_proof.h:90-185); the subfield of a prime field is the field, so the opened columns are one
// "subfield" run after an empty full-field run
int zk256_proof_write(const Zk256* z, uint8_t* buf, size_t cap, size_t* nbytes) {
  if (!z->have_proof) return lf_fail(z->c, LFGPU_ERR_ARG, "zk_proof_write: no proof");
  std::vector<uint8_t>& o = z->wire;  // serialised once per proof: the size query and the copy share it
  if (!z->wire_valid) {
  o.clear();
  auto pute = [&](const E& e) {
    uint8_t b[32];
    h256_to_bytes(e, b);
    o.insert(o.end(), b, b + 32);
  };
  auto putsz = [&](size_t g) {
    for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(g >> (8 * i)));
  };
  o.insert(o.end(), z->root, z->root + 32);
  for (size_t ly = 0; ly < z->proof.size(); ++ly) {
    const auto& P = z->proof[ly];
    const size_t logw = z->C->layers[ly].logw;
    for (size_t wi = 0; wi < logw; ++wi)
      for (int k = 0; k < 2; ++k) {
        pute(P.hp[0][2 * wi + k]);
        pute(P.hp[1][2 * wi + k]);
      }
    pute(P.wc[0]);
    pute(P.wc[1]);
  }
  for (const E& e : z->y_ldt) pute(e);
  for (const E& e : z->y_dot) pute(e);
  for (const E& e : z->y_q0) pute(e);
  for (const E& e : z->y_q2) pute(e);
  o.insert(o.end(), z->nonces.begin(), z->nonces.end());
  constexpr size_t kMaxRunLen = (size_t)1 << 25;
  const size_t nreq_elts = z->req.size();
  size_t ci = 0;
  bool subfield_run = false;
  while (ci < nreq_elts) {
    size_t runlen = 0;
    if (subfield_run) runlen = std::min(nreq_elts - ci, kMaxRunLen);  // in_subfield(e) is true for every element
    putsz(runlen);
    for (size_t i = ci; i < ci + runlen; ++i) pute(z->req[i]);
    ci += runlen;
    subfield_run = !subfield_run;
  }
  putsz(z->npath);
  o.insert(o.end(), z->path.begin(), z->path.begin() + 32 * z->npath);
  z->wire_valid = true;
  }
  *nbytes = o.size();
  if (buf) {
    if (cap < o.size()) return lf_fail(z->c, LFGPU_ERR_ARG, "zk_proof_write: buffer too small (%zu < %zu)", cap, o.size());
    memcpy(buf, o.data(), o.size());
  }
  return LFGPU_OK;
}

// ------------------------------------------------------------------ ZkVerifier<Fp256Base>
// ZkVerifier::recv_commitment + verify (lib/zk/zk_verifier.h:68-94) over the wire bytes, as zk.hip does for the 16-byte
// fields: ZkProof::read (zk_proof.h:107-112,218-345), verifier_constraints with aux == nullptr (Quad::bind_gh_all on the
// device), LigeroVerifier::verify (lib/ligero/ligero_verifier.h:42-270: Reed-Solomon extension of the rows of A and of the
// three y vectors on the device, checks at the opened columns on the host), MerkleCommitmentVerifier::verify.
namespace {
struct Parsed256 {
  uint8_t root[32];
  std::vector<Zk256::LayerPad> sc;
  std::vector<E> y_ldt, y_dot, y_q0, y_q2, req;
  std::vector<uint8_t> nonces, path;
  size_t npath = 0;
};
bool parse_proof256(const lfgpu_circuit* C, const lfgpu_ligero_param& p, const uint8_t* buf, size_t len, Parsed256& pr) {
  const uint8_t* q = buf;
  size_t left = len;
  bool bad = false;
  auto have = [&](size_t n) { return left >= n; };
  auto next = [&](size_t n) {
    const uint8_t* r = q;
    q += n;
    left -= n;
    return r;
  };
  auto elt = [&] {
    E e = e32_zero();
    if (!h256_of_bytes(next(32), e)) bad = true;  // of_bytes_field: value >= p
    return e;
  };
  auto size4 = [&] {
    const uint8_t* b = next(4);
    return (size_t)b[0] | (size_t)b[1] << 8 | (size_t)b[2] << 16 | (size_t)b[3] << 24;
  };
  if (!have(32)) return false;
  memcpy(pr.root, next(32), 32);
  pr.sc.assign(C->layers.size(), {});
  for (size_t ly = 0; ly < C->layers.size(); ++ly) {
    const size_t logw = C->layers[ly].logw;
    if (!have((logw * 4 + 2) * 32)) return false;
    auto& P = pr.sc[ly];
    P.hp[0].resize(2 * logw);
    P.hp[1].resize(2 * logw);
    for (size_t wi = 0; wi < logw; ++wi)
      for (int k = 0; k < 2; ++k) {
        P.hp[0][2 * wi + k] = elt();
        P.hp[1][2 * wi + k] = elt();
      }
    P.wc[0] = elt();
    P.wc[1] = elt();
  }
  auto vec = [&](std::vector<E>& v, size_t n) {
    if (!have(n * 32)) return false;
    v.resize(n);
    for (auto& e : v) e = elt();
    return true;
  };
  if (!vec(pr.y_ldt, p.block) || !vec(pr.y_dot, p.dblock) || !vec(pr.y_q0, p.r) || !vec(pr.y_q2, p.dblock - p.block)) return false;
  if (!have(p.nreq * 32)) return false;
  pr.nonces.assign(q, q + p.nreq * 32);
  next(p.nreq * 32);
  const size_t total = p.nreq * p.nrow;
  constexpr size_t kMaxRunLen = (size_t)1 << 25, kMaxNumDigests = (size_t)1 << 25;
  pr.req.assign(total, e32_zero());
  size_t ci = 0;
  while (ci < total) {  // alternating full-field / subfield runs; both are 32-byte images for a prime field
    if (!have(4)) return false;
    const size_t runlen = size4();
    if (runlen >= kMaxRunLen || ci + runlen > total || !have(runlen * 32)) return false;
    for (size_t i = ci; i < ci + runlen; ++i) pr.req[i] = elt();
    ci += runlen;
  }
  if (!have(4)) return false;
  const size_t sz = size4();
  if (sz < p.nreq || sz >= kMaxNumDigests || sz > p.nreq * p.mc_pathlen || !have(sz * 32)) return false;
  pr.npath = sz;
  pr.path.assign(q, q + sz * 32);
  next(sz * 32);
  return !bad;
}
}  // namespace

int zk256_verify(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, const uint8_t* proof, size_t proof_len, const void* h_pub,
                 const lfgpu_transcript_ops* tso, bool committed, int* ok, const char** why_out) {
  static const char* kWhy[] = {"ok", "proof does not parse", "merkle_check failed", "low_degree_check failed", "dot_check failed", "wrong dot product",
                               "quadratic_check failed"};
  *ok = 0;
  LF_TRY(ts256_ok(c, tso));
  auto fail = [&](int w) {
    if (why_out) *why_out = kWhy[w];
    return LFGPU_OK;
  };
  const lfgpu_circuit_info& I = C->info;
  const size_t nl = C->layers.size(), npub = I.npub_in, n_witness = I.ninputs - npub;
  size_t pad_size = 0;
  for (const auto& l : C->layers) pad_size += layer_size256(l.logw);
  lfgpu_ligero_param p{};
  LF_TRY(lfgpu_ligero_param_init(&p, LFGPU_FIELD_P256, 0, n_witness + pad_size, nl, rateinv, nreq, block_enc));
  Parsed256 pr;
  if (!parse_proof256(C, p, proof, proof_len, pr)) return fail(1);
  LF_HIP(c, hipSetDevice(c->device));
  const F256 F;
  const Ts256 ts{tso, tso->user};
  const E* pub = (const E*)h_pub;
  // recv_commitment (unless the caller did it), initialize_sumcheck_fiat_shamir
  if (!committed) ts.write_bytes(pr.root, 32);
  ts.write_bytes(I.id, 32);
  for (size_t i = 0; i < npub; ++i) ts.write_elt(pub[i]);
  ts.write_elt(F.zero);
  ts.write_bytes(C->zeros->data(), I.nterms);
  // device buffers: EQ table of the input constraint | rows [0, nwqrow) = [0^r | A_i], then y_ldt, y_dot, y_quad | gathered columns
  const size_t nrows_dev = p.nwqrow + 3, ld = p.block_enc;
  void* dv = nullptr;
  LF_TRY(lf_scratch4(c, (I.ninputs + nrows_dev * ld + nrows_dev * p.nreq) * 32 + 256, &dv));
  E* d_eq = (E*)dv;
  E* d_T = d_eq + I.ninputs;
  E* d_req = d_T + nrows_dev * ld;
  Constraints256 cs;
  LF_TRY(build_constraints256(c, C, F, ts, pr.sc, nullptr, pub, d_eq, cs));
  std::vector<size_t> lqc(3 * nl);
  {
    size_t pi = n_witness;
    for (size_t ly = 0; ly < nl; ++ly) {  // setup_lqc (zk_common.h:149-160)
      const size_t cp = pi + 4 * C->layers[ly].logw;
      lqc[3 * ly] = cp;
      lqc[3 * ly + 1] = cp + 1;
      lqc[3 * ly + 2] = cp + 2;
      pi += layer_size256(C->layers[ly].logw);
    }
  }
  // LigeroVerifier::verify: replay the challenges
  uint8_t hash_of_A[32] = {0xde, 0xad, 0xbe, 0xef};
  ts.write_bytes(hash_of_A, 32);
  std::vector<E> u_ldt(p.nwqrow), alphal(cs.n), alphaq(3 * p.nq), u_quad(p.nqtriples ? p.nqtriples : 1);
  for (auto& e : u_ldt) e = ts.elt();
  for (auto& e : alphal) e = ts.elt();
  for (auto& e : alphaq) e = ts.elt();
  for (size_t i = 0; i < p.nqtriples; ++i) u_quad[i] = ts.elt();
  ts.write_array(pr.y_ldt.data(), pr.y_ldt.size());
  ts.write_array(pr.y_dot.data(), pr.y_dot.size());
  ts.write_array(pr.y_q0.data(), pr.y_q0.size());
  ts.write_array(pr.y_q2.data(), pr.y_q2.size());
  std::vector<size_t> idx(p.nreq);
  ts.choose(p.block_ext, p.nreq, idx.data());
  auto req_at = [&](size_t i, size_t j) -> const E& { return pr.req[i * p.nreq + j]; };
  {  // merkle_check: leaf r = SHA-256(nonce_r || column r of the opening)
    std::vector<uint8_t> leaves(p.nreq * 32);
    for (size_t r = 0; r < p.nreq; ++r) {
      Sha256 sh;
      sh.update(&pr.nonces[32 * r], 32);
      for (size_t i = 0; i < p.nrow; ++i) {
        uint8_t eb[32];
        h256_to_bytes(req_at(i, r), eb);
        sh.update(eb, 32);
      }
      sh.digest(&leaves[32 * r]);
    }
    if (!lf_merkle_verify(p.block_ext, pr.root, pr.path.data(), pr.npath, leaves.data(), idx.data(), p.nreq)) return fail(2);
  }
  // rows of A (inner_product_vector + layout_Aext), y vectors; extension to block_enc; the opened columns
  std::vector<uint64_t> a_idx;
  std::vector<E> a_val;
  inner_product_sparse256(F, p, cs, alphal, lqc, alphaq, a_idx, a_val);
  LF_HIP(c, hipMemset2DAsync(d_T, ld * 32, 0, p.dblock * 32, nrows_dev, c->stream));
  hipLaunchKernelGGL(a_rows_dense256_kernel, dim3(nblk(n_witness)), dim3(Z_THREADS), 0, c->stream, (u32)p.r, (u32)p.w, ld, alphal[cs.n - 1], (const E*)d_eq + npub,
                     n_witness, d_T);
  if (!a_idx.empty()) {
    void* d_sp = nullptr;
    LF_TRY(lf_scratch2(c, a_idx.size() * 40 + 64, &d_sp));
    E* d_val = (E*)d_sp;
    u64* d_idx = (u64*)(d_val + a_idx.size());
    LF_HIP(c, hipMemcpyAsync(d_val, a_val.data(), a_idx.size() * 32, hipMemcpyHostToDevice, c->stream));
    LF_HIP(c, hipMemcpyAsync(d_idx, a_idx.data(), a_idx.size() * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(a_rows_sparse256_kernel, dim3(nblk(a_idx.size())), dim3(Z_THREADS), 0, c->stream, (u32)p.r, (u32)p.w, ld, (const u64*)d_idx, (const E*)d_val,
                       a_idx.size(), d_T);
  }
  LF_HIP(c, hipGetLastError());
  LF_HIP(c, hipMemcpyAsync(d_T + (p.nwqrow + 0) * ld, pr.y_ldt.data(), p.block * 32, hipMemcpyHostToDevice, c->stream));
  LF_HIP(c, hipMemcpyAsync(d_T + (p.nwqrow + 1) * ld, pr.y_dot.data(), p.dblock * 32, hipMemcpyHostToDevice, c->stream));
  E* yq = d_T + (p.nwqrow + 2) * ld;  // y_quad = y_quad_0 | 0^w | y_quad_2
  LF_HIP(c, hipMemcpyAsync(yq, pr.y_q0.data(), p.r * 32, hipMemcpyHostToDevice, c->stream));
  LF_HIP(c, hipMemcpyAsync(yq + p.block, pr.y_q2.data(), (p.dblock - p.block) * 32, hipMemcpyHostToDevice, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));  // the staging area is the Reed-Solomon encoder's work space next
  LF_TRY(lfgpu_fp256_rs_encode_rows(c, p.nwqrow + 1, p.block, p.block_enc, d_T, ld));                    // A rows and y_ldt
  LF_TRY(lfgpu_fp256_rs_encode_rows(c, 2, p.dblock, p.block_enc, d_T + (p.nwqrow + 1) * ld, ld));         // y_dot, y_quad
  {
    void* di = nullptr;
    LF_TRY(lf_scratch2(c, p.nreq * 8 + 64, &di));
    std::vector<u64> ix(idx.begin(), idx.end());
    LF_HIP(c, hipMemcpyAsync(di, ix.data(), p.nreq * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(gather_columns256_kernel, dim3(nblk(nrows_dev * p.nreq)), dim3(Z_THREADS), 0, c->stream, (u32)nrows_dev, ld, p.dblock, (const E*)d_T,
                       (const u64*)di, (u32)p.nreq, d_req);
    LF_HIP(c, hipGetLastError());
    LF_HIP(c, hipStreamSynchronize(c->stream));
  }
  std::vector<E> ext(nrows_dev * p.nreq);
  LF_TRY(lfgpu_memcpy_d2h(c, ext.data(), d_req, ext.size() * 32));
  auto ext_at = [&](size_t row, size_t j) -> const E& { return ext[row * p.nreq + j]; };
  for (size_t j = 0; j < p.nreq; ++j) {  // low_degree_check
    E yc = req_at(p.ildt, j);
    for (size_t i = 0; i < p.nwqrow; ++i) yc = F.add(yc, F.mul(u_ldt[i], req_at(i + p.iw, j)));
    if (!e32_eq(yc, ext_at(p.nwqrow, j))) return fail(3);
  }
  for (size_t j = 0; j < p.nreq; ++j) {  // dot_check
    E yc = req_at(p.idot, j);
    for (size_t i = 0; i < p.nwqrow; ++i) yc = F.add(yc, F.mul(ext_at(i, j), req_at(i + p.iw, j)));
    if (!e32_eq(yc, ext_at(p.nwqrow + 1, j))) return fail(4);
  }
  {  // the putative value of the inner product
    E want = F.zero, got = F.zero;
    for (size_t k = 0; k < cs.n; ++k) want = F.add(want, F.mul(cs.b[k], alphal[k]));
    for (size_t j = 0; j < p.w; ++j) got = F.add(got, pr.y_dot[p.r + j]);
    if (!e32_eq(want, got)) return fail(5);
  }
  {  // quadratic_check
    const size_t iqx = p.iq, iqy = iqx + p.nqtriples, iqz = iqy + p.nqtriples;
    for (size_t j = 0; j < p.nreq; ++j) {
      E yc = req_at(p.iquad, j);
      for (size_t i = 0; i < p.nqtriples; ++i) {
        const E tmp = F.sub(req_at(iqz + i, j), F.mul(req_at(iqx + i, j), req_at(iqy + i, j)));  // z - x y
        yc = F.add(yc, F.mul(u_quad[i], tmp));
      }
      if (!e32_eq(yc, ext_at(p.nwqrow + 2, j))) return fail(6);
    }
  }
  *ok = 1;
  return fail(0);
}
