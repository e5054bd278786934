// sumcheck.hip -- K7 (round evaluation), K8 (QW scatter), K9 (dense / hquad bind),
// K12 (Ligero row combinations, column gather).
//
// Reference: per-round body of ProverLayers::layer (lib/sumcheck/prover_layers.h:230-263),
// ProverLayers::evaluations (:357-402), Dense::bind (lib/arrays/dense.h:70-87),
// HQuad::bind_h (lib/sumcheck/hquad.h:90-123), LigeroProver::low_degree_proof /
// compute_req (lib/ligero/ligero_prover.h:281-291,346-351), Blas (lib/algebra/blas.h:62-110).
//
// The Fiat-Shamir transcript stays on the host, so each round returns two field
// elements through a pinned mailbox and receives one challenge.
#include "ctx.h"
#include "runfold.h"
#include <sched.h>

#include <atomic>
#include <chrono>
#include <utility>
#include <vector>

#define SC_THREADS 256
#define SC_MAX_BLOCKS 1024

template <int F>
__device__ __forceinline__ elt_t block_reduce(elt_t v, elt_t* sh) {
  // wave reduction by shuffles of the four dwords, then across waves through LDS
  for (int off = 32; off > 0; off >>= 1) {
    elt_t o;
    o.lo = __shfl_down(v.lo, off, 64);
    o.hi = __shfl_down(v.hi, off, 64);
    v = Fld<F>::add(v, o);
  }
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (u32 w = 1; w < SC_THREADS / 64; ++w) v = Fld<F>::add(v, sh[w]);
  }
  __syncthreads();
  return v;
}

// partial[2*b] = a0 part, partial[2*b+1] = a2 part of block b
// clean != nullptr (== QW): every accumulator is zeroed by the lane that has read it, so that the NEXT evaluation's scatter finds
// its targets cleared without a memset of its own (one dispatch less per round-hand of the per-launch path)
template <int F>
__global__ __launch_bounds__(SC_THREADS) void sumcheck_partials_kernel(size_t n, const elt_t* QW /* may alias `clean` */,
                                                                       const elt_t* __restrict__ W,
                                                                       elt_t* __restrict__ partial, elt_t* clean) {
  __shared__ elt_t sh[SC_THREADS / 64];
  const size_t nodd = n / 2;
  elt_t a0 = elt_zero(), a2 = elt_zero();
  for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < nodd; i += (size_t)gridDim.x * SC_THREADS) {
    elt_t q0 = ld16(&QW[2 * i]), q1 = ld16(&QW[2 * i + 1]);
    elt_t w0 = ld16(&W[2 * i]), w1 = ld16(&W[2 * i + 1]);
    if (clean) {
      st16(&clean[2 * i], elt_zero());
      st16(&clean[2 * i + 1], elt_zero());
    }
    a0 = Fld<F>::add(a0, Fld<F>::mul(q0, w0));
    a2 = Fld<F>::add(a2, Fld<F>::mul(Fld<F>::sub(q1, q0), Fld<F>::sub(w1, w0)));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && 2 * nodd < n) {  // odd tail (prover_layers.h:381-388)
    elt_t t = Fld<F>::mul(ld16(&QW[2 * nodd]), ld16(&W[2 * nodd]));
    if (clean) st16(&clean[2 * nodd], elt_zero());
    a0 = Fld<F>::add(a0, t);
    a2 = Fld<F>::add(a2, t);
  }
  a0 = block_reduce<F>(a0, sh);
  a2 = block_reduce<F>(a2, sh);
  if (threadIdx.x == 0) {
    st16(&partial[2 * blockIdx.x], a0);
    st16(&partial[2 * blockIdx.x + 1], a2);
  }
}
// post != nullptr: the sums go to the coherent pinned words the resident kernels post to (a0, a2, then the sequence number),
// so the host spins on its own memory instead of a copy + hipStreamSynchronize
template <int F>
__global__ __launch_bounds__(SC_THREADS) void sumcheck_final_kernel(u32 nblocks, const elt_t* __restrict__ partial,
                                                                    elt_t* __restrict__ out, volatile u64* post, u64 seq) {
  __shared__ elt_t sh[SC_THREADS / 64];
  elt_t a0 = elt_zero(), a2 = elt_zero();
  for (u32 b = threadIdx.x; b < nblocks; b += SC_THREADS) {
    a0 = Fld<F>::add(a0, ld16(&partial[2 * b]));
    a2 = Fld<F>::add(a2, ld16(&partial[2 * b + 1]));
  }
  a0 = block_reduce<F>(a0, sh);
  a2 = block_reduce<F>(a2, sh);
  if (threadIdx.x == 0) {
    if (post) {
      u64* po = (u64*)post;
      __hip_atomic_store(&po[0], a0.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&po[1], a0.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&po[2], a2.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&po[3], a2.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&po[8], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      __hip_atomic_store(&po[5], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      st16(&out[0], a0);
      st16(&out[1], a2);
    }
  }
}

// QW[h[hand]] ^= v * Wother[h[1-hand]]  -- GF(2^128): addition is XOR, so two 64-bit atomic XORs per
// term are exact and order-independent.  One wire (the constant 1) is the target of up to 2*10^5 terms of a
// flatsha256 layer and its terms are mostly adjacent in canonical order, so equal-target neighbours are first
// folded inside the wave with shuffles; only the first lane of each run of equal targets issues atomics.
#define SCAT_THREADS 1024  // the fold of equal-target runs spans the block (runfold.h)
__global__ __launch_bounds__(SCAT_THREADS) void qw_scatter_gf_kernel(size_t n, const uint2* __restrict__ hc,
                                                                     const elt_t* __restrict__ vc, int hand,
                                                                     const elt_t* __restrict__ Wo, u64* __restrict__ QW) {
  const size_t i = (size_t)blockIdx.x * SCAT_THREADS + threadIdx.x;
  u32 key = 0xffffffffu;
  elt_t t = elt_zero();
  if (i < n) {
    uint2 h = hc[i];
    key = hand ? h.y : h.x;
    t = gf_mul(ld16(&vc[i]), ld16(&Wo[hand ? h.x : h.y]));
  }
  gf_run_fold_commit<SCAT_THREADS>(key, t, QW);  // runs of CONTIGUOUS equal targets: one atomic pair per run and block
}

// Fp128 has no 128-bit atomic, but residues add as plain integers: every product v*W (a canonical residue
// < p < 2^128) is split into four 32-bit limbs that are accumulated with 64-bit atomic adds into a
// 4 x u64 accumulator per target (no carry can be lost below 2^32 terms per target); a second kernel
// recombines the limbs and reduces mod p once.  Exact and independent of arrival order.
__global__ __launch_bounds__(SC_THREADS) void qw_scatter_fp_kernel(size_t n, const uint2* __restrict__ hc,
                                                                   const elt_t* __restrict__ vc, int hand,
                                                                   const elt_t* __restrict__ Wo, u64* __restrict__ acc) {
  const size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (i >= n) return;
  uint2 h = hc[i];
  const u32 p0 = hand ? h.y : h.x, p1 = hand ? h.x : h.y;
  elt_t t = fp_mul(ld16(&vc[i]), ld16(&Wo[p1]));
  u64* a = acc + 4 * (size_t)p0;
  atomicAdd(&a[0], (u64)(u32)t.lo);
  atomicAdd(&a[1], t.lo >> 32);
  atomicAdd(&a[2], (u64)(u32)t.hi);
  atomicAdd(&a[3], t.hi >> 32);
}
// S = sum_k acc[k] * 2^(32k) mod p on PLAIN integers (S is then again a Montgomery image, because images add):
// fp_reduce_limbs (fields.h) folds the 160-bit integer with 2^128 = 2^108 - 1.
__global__ __launch_bounds__(SC_THREADS) void fp_limb_normalize_kernel(size_t n, const u64* __restrict__ acc, elt_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (i >= n) return;
  st16(&out[i], fp_reduce_limbs(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]));
}

// out[i] = in[2i] + r*(in[2i+1]-in[2i]);  tail: in*(1-r)   (dense.h:70-87, affine.h:26-52)
template <int F>
__device__ __forceinline__ void dense_bind_body(u32 bx, size_t n0, elt_t r, const elt_t* __restrict__ in, elt_t* __restrict__ out) {
  size_t i = (size_t)bx * SC_THREADS + threadIdx.x;
  size_t nout = (n0 + 1) / 2;
  if (i >= nout) return;
  elt_t f0 = ld16(&in[2 * i]);
  elt_t v;
  if (2 * i + 1 < n0) {
    elt_t f1 = ld16(&in[2 * i + 1]);
    v = Fld<F>::add(f0, Fld<F>::mul(Fld<F>::sub(f1, f0), r));
  } else {
    v = Fld<F>::sub(f0, Fld<F>::mul(f0, r));
  }
  st16(&out[i], v);
}
template <int F>
__global__ __launch_bounds__(SC_THREADS) void dense_bind_kernel(size_t n0, elt_t r, const elt_t* __restrict__ in,
                                                                elt_t* __restrict__ out) {
  dense_bind_body<F>(blockIdx.x, n0, r, in, out);
}

// ---- HQuad::bind_h as an order-preserving compaction.
// A term is the SECOND half of a merged pair iff its predecessor has the same other-hand
// corner, the same h>>1 and h+1 == own h (hquad.h:99-103); such pairs cannot chain (the
// first has even h, the second odd), so the flag is local.
__device__ __forceinline__ bool is_second(const uint2* hc, size_t i, int hand) {
  if (i == 0) return false;
  uint2 a = hc[i - 1], b = hc[i];
  u32 ah = hand ? a.y : a.x, ao = hand ? a.x : a.y, bh = hand ? b.y : b.x, bo = hand ? b.x : b.y;
  return ao == bo && (ah >> 1) == (bh >> 1) && bh == ah + 1;
}
__global__ __launch_bounds__(SC_THREADS) void hquad_count_kernel(size_t n, const uint2* __restrict__ hc, int hand,
                                                                 u32* __restrict__ block_counts) {
  __shared__ u32 cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  bool head = i < n && !is_second(hc, i, hand);
  u64 mask = __ballot(head);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, (u32)__popcll(mask));
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = cnt;
}
// exclusive scan of block counts by one workgroup (nblocks small: n / 256)
__global__ __launch_bounds__(1024) void hquad_scan_kernel(u32 nblocks, u32* __restrict__ block_counts,
                                                          u32* __restrict__ total) {
  __shared__ u32 sh[1024];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (u32 base = 0; base < nblocks; base += 1024) {
    u32 i = base + threadIdx.x;
    u32 v = i < nblocks ? block_counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (u32 off = 1; off < 1024; off <<= 1) {
      u32 t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    u32 incl = sh[threadIdx.x];
    if (i < nblocks) block_counts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
template <int F>
__device__ __forceinline__ void hquad_emit_body(u32 bx, u32* wave_off /* LDS, SC_THREADS / 64 words */, size_t n, const uint2* __restrict__ hc,
                                                const elt_t* __restrict__ vc, elt_t r, int hand, const u32* __restrict__ block_off,
                                                uint2* __restrict__ hc_out, elt_t* __restrict__ vc_out) {
  size_t i = (size_t)bx * SC_THREADS + threadIdx.x;
  bool head = i < n && !is_second(hc, i, hand);
  u64 mask = __ballot(head);
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_off[wave] = (u32)__popcll(mask);
  __syncthreads();
  u32 off = block_off[bx];
  for (u32 w = 0; w < wave; ++w) off += wave_off[w];
  off += (u32)__popcll(mask & ((1ull << lane) - 1));
  if (!head) return;
  uint2 h = hc[i];
  u32 hh = hand ? h.y : h.x;
  elt_t v0 = ld16(&vc[i]), v;
  if (i + 1 < n && is_second(hc, i + 1, hand)) {
    elt_t v1 = ld16(&vc[i + 1]);
    v = Fld<F>::add(v0, Fld<F>::mul(Fld<F>::sub(v1, v0), r));  // affine_interpolation
  } else if ((hh & 1) == 0) {
    v = Fld<F>::sub(v0, Fld<F>::mul(v0, r));  // affine_interpolation_nz_z
  } else {
    v = Fld<F>::mul(v0, r);  // affine_interpolation_z_nz
  }
  if (hand) h.y = hh >> 1; else h.x = hh >> 1;
  hc_out[off] = h;
  st16(&vc_out[off], v);
}
template <int F>
__global__ __launch_bounds__(SC_THREADS) void hquad_emit_kernel(size_t n, const uint2* __restrict__ hc,
                                                                const elt_t* __restrict__ vc, elt_t r, int hand,
                                                                const u32* __restrict__ block_off,
                                                                uint2* __restrict__ hc_out, elt_t* __restrict__ vc_out) {
  __shared__ u32 wave_off[SC_THREADS / 64];
  hquad_emit_body<F>(blockIdx.x, wave_off, n, hc, vc, r, hand, block_off, hc_out, vc_out);
}
// Dense::bind of the round-hand's own array and HQuad::bind_h (recorded offsets) in ONE launch: both need only the challenge,
// touch different arrays, and a per-launch round-hand is a chain of dependent dispatches in which every dispatch counts
// (DESIGN.md 4.9).  Blocks [0, nbD) bind the dense array, the rest emit the HQUAD.
template <int F>
__global__ __launch_bounds__(SC_THREADS) void bind_both_kernel(u32 nbD, size_t n0, elt_t r, const elt_t* __restrict__ in, elt_t* __restrict__ out,
                                                               size_t n, const uint2* __restrict__ hc, const elt_t* __restrict__ vc, int hand,
                                                               const u32* __restrict__ block_off, uint2* __restrict__ hc_out,
                                                               elt_t* __restrict__ vc_out) {
  __shared__ u32 wave_off[SC_THREADS / 64];
  if (blockIdx.x < nbD) dense_bind_body<F>(blockIdx.x, n0, r, in, out);
  else hquad_emit_body<F>(blockIdx.x - nbD, wave_off, n, hc, vc, r, hand, block_off, hc_out, vc_out);
}

// ---- fused single-workgroup steps for the small rounds of a layer.
// Once the HQUAD and both hand arrays have <= LF_SC_SMALL_MAX entries the per-round work is microseconds and the
// round is bound by launches and synchronisations (6 kernels + 2 stream syncs per round-hand on the path above).
// One 1024-thread workgroup then does everything between two transcript interactions: the binds of the previous
// round-hand (Dense::bind + HQuad::bind_h with the challenge the host just drew) and the two partial sums of the
// next one (QW scatter + evaluations), posted to coherent pinned memory the host polls.  Two drivers:
//   sc_small_step_kernel   one launch per round-hand, no stream synchronisation
//   sc_small_layer_kernel  ONE launch for all remaining round-hands of the layer: the workgroup stays resident and
//                          receives each challenge through a second coherent word (bounded wait, see below)
// Same arithmetic and the same order-preserving compaction as the kernels above, so the results are identical.
#define SM_THREADS 1024
struct ScState {  // the layer's device state as the workgroup tracks it
  const uint2* hc;
  const elt_t* vc;
  uint2* hc_other;
  elt_t* vc_other;
  u32 nh;
  elt_t* W[2];
  u32 nW[2];
};
struct ScShared {
  u32 wave[SM_THREADS / 64];
  u32 carry;
  elt_t red[2][SM_THREADS / 64];
  u64 cmd[3];
};

// Dense::bind of hand bh into Wdst (may alias the source) + HQuad::bind_h into the other half of the ping-pong
template <int F>
__device__ __forceinline__ void sc_bind(ScState& st, ScShared& sh, int bh, elt_t r, elt_t* Wdst) {
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {  // chunk k reads inputs [2048k, 2048k+2048) and then writes outputs [1024k, 1024k+1024): the writes never reach
     // the inputs of a later chunk, and a barrier separates them from this chunk's own reads
    const u32 n0 = st.nW[bh], nout = (n0 + 1) / 2;
    const elt_t* in = st.W[bh];
    for (u32 base = 0; base < nout; base += SM_THREADS) {
      const u32 i = base + tid;
      elt_t v = elt_zero();
      if (i < nout) {
        const elt_t f0 = ld16(&in[2 * i]);
        if (2 * i + 1 < n0) {
          const elt_t f1 = ld16(&in[2 * i + 1]);
          v = Fld<F>::add(f0, Fld<F>::mul(Fld<F>::sub(f1, f0), r));
        } else {
          v = Fld<F>::sub(f0, Fld<F>::mul(f0, r));
        }
      }
      __syncthreads();
      if (i < nout) st16(&Wdst[i], v);
    }
    st.W[bh] = Wdst;
    st.nW[bh] = nout;
  }
  if (tid == 0) sh.carry = 0;
  __syncthreads();
  const uint2* hc = st.hc;
  const elt_t* vc = st.vc;
  const u32 nh = st.nh;
  for (u32 base = 0; base < nh; base += SM_THREADS) {
    const u32 i = base + tid;
    const bool head = i < nh && !is_second(hc, i, bh);
    const u64 mask = __ballot(head);
    if (lane == 0) sh.wave[wave] = (u32)__popcll(mask);
    __syncthreads();
    u32 off = sh.carry;
    for (u32 w = 0; w < wave; ++w) off += sh.wave[w];
    off += (u32)__popcll(mask & ((1ull << lane) - 1));
    if (head) {
      uint2 h = hc[i];
      const u32 hh = bh ? h.y : h.x;
      const elt_t v0 = ld16(&vc[i]);
      elt_t v;
      if (i + 1 < nh && is_second(hc, i + 1, bh)) {
        const elt_t v1 = ld16(&vc[i + 1]);
        v = Fld<F>::add(v0, Fld<F>::mul(Fld<F>::sub(v1, v0), r));
      } else if ((hh & 1) == 0) {
        v = Fld<F>::sub(v0, Fld<F>::mul(v0, r));
      } else {
        v = Fld<F>::mul(v0, r);
      }
      if (bh) h.y = hh >> 1; else h.x = hh >> 1;
      st.hc_other[off] = h;
      st16(&st.vc_other[off], v);
    }
    __syncthreads();
    if (tid == 0) {
      u32 tot = 0;
      for (u32 w = 0; w < SM_THREADS / 64; ++w) tot += sh.wave[w];
      sh.carry += tot;
    }
    __syncthreads();
  }
  st.nh = sh.carry;
  uint2* oh = const_cast<uint2*>(st.hc);
  elt_t* ov = const_cast<elt_t*>(st.vc);
  st.hc = st.hc_other;
  st.vc = st.vc_other;
  st.hc_other = oh;
  st.vc_other = ov;
  __threadfence_block();
  __syncthreads();
}

// QW scatter + ProverLayers::evaluations for hand eh; the sums are valid in thread 0
template <int F>
__device__ __forceinline__ void sc_eval(const ScState& st, ScShared& sh, int eh, u64* QW, elt_t& a0, elt_t& a2) {
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 nq = st.nW[eh], nh = st.nh;
  const elt_t* Wo = st.W[1 - eh];
  const uint2* hc = st.hc;
  const elt_t* vc = st.vc;
  const u32 words = (F == FIELD_GF2_128 ? 2u : 4u) * nq;
  for (u32 i = tid; i < words; i += SM_THREADS) QW[i] = 0;
  __syncthreads();
  for (u32 base = 0; base < nh; base += SM_THREADS) {  // QW[h[hand]] += v * Wother[h[1-hand]]
    const u32 i = base + tid;
    const bool valid = i < nh;
    u32 key = 0xffffffffu;
    elt_t t = elt_zero();
    if (valid) {
      const uint2 h = hc[i];
      key = eh ? h.y : h.x;
      t = Fld<F>::mul(ld16(&vc[i]), ld16(&Wo[eh ? h.x : h.y]));
    }
    if (F == FIELD_GF2_128) {  // fold runs of equal targets inside the wave first (see qw_scatter_gf_kernel)
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
      if (valid && head) {
        atomicXor(&QW[2 * (size_t)key], t.lo);
        atomicXor(&QW[2 * (size_t)key + 1], t.hi);
      }
    } else if (valid) {  // integer limb accumulators (see qw_scatter_fp_kernel)
      u64* acc = QW + 4 * (size_t)key;
      atomicAdd(&acc[0], (u64)(u32)t.lo);
      atomicAdd(&acc[1], t.lo >> 32);
      atomicAdd(&acc[2], (u64)(u32)t.hi);
      atomicAdd(&acc[3], t.hi >> 32);
    }
  }
  __threadfence_block();
  __syncthreads();
  const elt_t* Wh = st.W[eh];
  const u32 nodd = nq / 2;
  auto qw_at = [&](u32 j) -> elt_t {
    if (F == FIELD_GF2_128) return elt_t{QW[2 * (size_t)j], QW[2 * (size_t)j + 1]};
    const u64* q = QW + 4 * (size_t)j;  // recombine the limbs and reduce once
    return fp_reduce_limbs(q[0], q[1], q[2], q[3]);
  };
  a0 = elt_zero();
  a2 = elt_zero();
  for (u32 i = tid; i < nodd; i += SM_THREADS) {
    const elt_t q0 = qw_at(2 * i), q1 = qw_at(2 * i + 1);
    const elt_t w0 = ld16(&Wh[2 * i]), w1 = ld16(&Wh[2 * i + 1]);
    a0 = Fld<F>::add(a0, Fld<F>::mul(q0, w0));
    a2 = Fld<F>::add(a2, Fld<F>::mul(Fld<F>::sub(q1, q0), Fld<F>::sub(w1, w0)));
  }
  if (tid == 0 && 2 * nodd < nq) {  // odd tail (prover_layers.h:381-388)
    const elt_t t = Fld<F>::mul(qw_at(2 * nodd), ld16(&Wh[2 * nodd]));
    a0 = Fld<F>::add(a0, t);
    a2 = Fld<F>::add(a2, t);
  }
  for (int off = 32; off > 0; off >>= 1) {
    elt_t o0, o2;
    o0.lo = __shfl_down(a0.lo, off, 64); o0.hi = __shfl_down(a0.hi, off, 64);
    o2.lo = __shfl_down(a2.lo, off, 64); o2.hi = __shfl_down(a2.hi, off, 64);
    a0 = Fld<F>::add(a0, o0);
    a2 = Fld<F>::add(a2, o2);
  }
  if (lane == 0) {
    sh.red[0][wave] = a0;
    sh.red[1][wave] = a2;
  }
  __syncthreads();
  if (tid == 0) {
    a0 = sh.red[0][0];
    a2 = sh.red[1][0];
    for (u32 w = 1; w < SM_THREADS / 64; ++w) {
      a0 = Fld<F>::add(a0, sh.red[0][w]);
      a2 = Fld<F>::add(a2, sh.red[1][w]);
    }
  }
  __syncthreads();
}

// thread 0: post {x0, x1, nh, scalar} and then the sequence number, visible to the host in that order
__device__ __forceinline__ void sc_post(const ScState& st, elt_t x0, elt_t x1, u64 seq, u64 status, volatile u64* post) {
  const elt_t s = st.nh ? ld16(&st.vc[0]) : elt_zero();
  post[0] = x0.lo; post[1] = x0.hi; post[2] = x1.lo; post[3] = x1.hi;
  post[4] = st.nh;
  post[6] = s.lo; post[7] = s.hi;
  post[8] = status;
  __threadfence_system();
  __hip_atomic_store((u64*)&post[5], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ ScState sc_state_of(const ScSmall& a) {
  ScState st;
  st.hc = a.hc_in; st.vc = a.vc_in; st.hc_other = a.hc_out; st.vc_other = a.vc_out;
  st.nh = a.nh;
  st.W[0] = a.W[0]; st.W[1] = a.W[1];
  st.nW[0] = a.nW[0]; st.nW[1] = a.nW[1];
  return st;
}

template <int F>
__global__ __launch_bounds__(SM_THREADS) void sc_small_step_kernel(ScSmall a, u64 seq, volatile u64* __restrict__ post) {
  __shared__ ScShared sh;
  ScState st = sc_state_of(a);
  if (a.do_bind) sc_bind<F>(st, sh, a.bind_hand, a.r, a.Wdst);
  elt_t a0 = elt_zero(), a2 = elt_zero();
  if (a.do_eval) sc_eval<F>(st, sh, a.eval_hand, a.QW, a0, a2);
  if (threadIdx.x == 0) {
    if (!a.do_eval) {  // end of the layer: the two bound hand arrays (and the HQUAD scalar)
      a0 = st.nW[0] ? ld16(&st.W[0][0]) : elt_zero();
      a2 = st.nW[1] ? ld16(&st.W[1][0]) : elt_zero();
    }
    sc_post(st, a0, a2, seq, 0, post);
  }
}

// All remaining round-hands [rh0, rh1) of a layer in one launch.  After posting a round's sums under sequence
// number seq0 + k the workgroup waits until the host answers with the challenge in cmd[0..1] and cmd[2] == seq0 + k.
// The wait is BOUNDED: after `timeout_ticks` of the constant-rate wall clock the kernel posts status 1 and every
// wave leaves, so a host that went away can never leave the workgroup resident.
template <int F>
__global__ __launch_bounds__(SM_THREADS) void sc_small_layer_kernel(ScSmall a, u32 rh0, u32 rh1, elt_t* d_W_shared, elt_t* wtmp,
                                                                    u64 seq0, u64 timeout_ticks, volatile u64* __restrict__ post,
                                                                    const volatile u64* __restrict__ cmd) {
  __shared__ ScShared sh;
  ScState st = sc_state_of(a);
  u64 seq = seq0;
  for (u32 rh = rh0; rh < rh1; ++rh, ++seq) {
    const int hand = (int)(rh & 1);
    elt_t a0, a2;
    sc_eval<F>(st, sh, hand, a.QW, a0, a2);
    if (threadIdx.x == 0) {
      sc_post(st, a0, a2, seq, 0, post);
      const u64 t0 = wall_clock64();
      u64 got = 0;
      for (;;) {
        got = __hip_atomic_load((const u64*)&cmd[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (got == seq) break;
        if (wall_clock64() - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(8);
      }
      sh.cmd[2] = got;
      sh.cmd[0] = __hip_atomic_load((const u64*)&cmd[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      sh.cmd[1] = __hip_atomic_load((const u64*)&cmd[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (sh.cmd[2] != seq) {  // uniform: the host did not answer in time
      if (threadIdx.x == 0) sc_post(st, elt_zero(), elt_zero(), seq, 1, post);
      return;
    }
    const elt_t r{sh.cmd[0], sh.cmd[1]};
    __syncthreads();
    elt_t* dst = (hand == 0 && st.W[0] == d_W_shared) ? wtmp : st.W[hand];  // hand 0 detaches from the shared input
    sc_bind<F>(st, sh, hand, r, dst);
  }
  if (threadIdx.x == 0) {
    const elt_t w0 = st.nW[0] ? ld16(&st.W[0][0]) : elt_zero();
    const elt_t w1 = st.nW[1] ? ld16(&st.W[1][0]) : elt_zero();
    sc_post(st, w0, w1, seq, 0, post);
  }
}

// y[j] += sum_i u[i]*T[i][j].  Workgroup = 64 columns x 16 row slices folded through LDS (one lane per column
// alone leaves a handful of workgroups with nrows sequential products each).
template <int F>
__global__ __launch_bounds__(1024) void rows_axpy_kernel(u32 nrows, size_t n, elt_t* __restrict__ y, const elt_t* __restrict__ u,
                                                         const elt_t* __restrict__ T, size_t ld) {
  __shared__ elt_t part[16][64];
  const u32 col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const size_t j = (size_t)blockIdx.x * 64 + col;
  elt_t acc = elt_zero();
  if (j < n)
    for (u32 i = slice; i < nrows; i += 16) acc = Fld<F>::add(acc, Fld<F>::mul(ld16(&T[(size_t)i * ld + j]), ld16(&u[i])));
  part[slice][col] = acc;
  __syncthreads();
  if (slice == 0 && j < n) {
    acc = ld16(&y[j]);
    for (u32 k = 0; k < 16; ++k) acc = Fld<F>::add(acc, part[k][col]);
    st16(&y[j], acc);
  }
}
__global__ void gather_columns_kernel(u32 nrow, size_t ld, size_t col0, const elt_t* __restrict__ T,
                                      const u64* __restrict__ idx, u32 nreq, elt_t* __restrict__ req) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nrow * nreq) return;
  u32 i = t / nreq, j = t % nreq;
  st16(&req[t], ld16(&T[(size_t)i * ld + col0 + idx[j]]));
}

// out[i] = a[i] (op) b[i]: op 0 add, 1 sub, 2 mul   (Field::addf/subf/mulf element-wise)
template <int F>
__global__ __launch_bounds__(SC_THREADS) void field_binop_kernel(int op, size_t n, const elt_t* __restrict__ a,
                                                                 const elt_t* __restrict__ b, elt_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (i >= n) return;
  elt_t x = ld16(&a[i]), y = ld16(&b[i]);
  st16(&out[i], op == 0 ? Fld<F>::add(x, y) : op == 1 ? Fld<F>::sub(x, y) : Fld<F>::mul(x, y));
}

// ------------------------------------------------------------------ C ABI
#define DISPATCH_FIELD(field, KERNEL, grid, block, ...)                                              \
  do {                                                                                               \
    if ((field) == LFGPU_FIELD_GF2_128)                                                              \
      hipLaunchKernelGGL(KERNEL<FIELD_GF2_128>, grid, block, 0, c->stream, __VA_ARGS__);             \
    else if ((field) == LFGPU_FIELD_FP128)                                                           \
      hipLaunchKernelGGL(KERNEL<FIELD_FP128>, grid, block, 0, c->stream, __VA_ARGS__);               \
    else                                                                                             \
      return lf_fail(c, LFGPU_ERR_ARG, "unknown field %d", (int)(field));                            \
  } while (0)


// ---- the rest of a layer in ONE launch on a grid that shrinks with the data.
// A workgroup's share of the products runs on one CU and a GF(2^128) product is ~510 instruction-equivalents, so the
// phases are issue-bound per CU long before the chip is busy: the layer runs on one 1024-thread workgroup per
// `per_wg` entries (512 for GF(2^128), 1024 for Fp128), work dealt in 64-entry chunks round-robin over the
// workgroups, synchronised by device-wide barriers.  Per round-hand (see sc_grid_layer_kernel):
//   barrier | sums a0, a2 -> slots -> the last-arriving workgroup folds and posts to the host |
//   layout of HQuad::bind_h (merge kinds, counts, barrier, offsets, halved corners: nothing of it needs the
//   challenge, so it hides behind the host round trip) | challenge |
//   Dense::bind + HQUAD values + the NEXT evaluation's QW from those values (double-buffered QW) |
//   workgroups beyond ceil(max size / per_wg) leave
// A device barrier costs ~2 us for 8 workgroups, 3.5 us for 32, 10 us for 128 (tools/ubench_sync.hip): hence the
// shrinking grid, down to ONE workgroup whose barriers are plain __syncthreads and whose state lives in LDS.  Only
// workgroup 0 polls host memory; the others take the challenge from a device-memory slot.  Every wait is bounded
// (abort flag + wall-clock timeout), so all waves always leave.
// Residency: the launch is an ORDINARY one (hipLaunchCooperativeKernel makes every rocprofv3 --kernel-trace run of
// ROCm 7.2 segfault at process exit, inside the HIP runtime's own finaliser -- tools/coop_exit_repro.hip shows it with a
// one-thread kernel and nothing of this library loaded; profiles/r02/rocprof_exit_crash.md).  The grid has at most
// LF_SC_GRID_WGS <= #CU workgroups, the host checks with the occupancy API that one workgroup fits a CU, and the stream is
// in order, so when the dispatch starts every workgroup is placed at once; should another queue hold CUs, its kernels
// end and the rest of the grid follows (a barrier that waits longer than the timeout aborts the layer with an error).
struct ScGridSync {  // device memory, zeroed before every launch
  u32 count, gen, abort, arrive;
  u64 chal4[4];  // the challenge as the host's four tagged words
  u64 pad_[2];
  u32 l1_bar[8 * 16];  // first-level arrival counters of the barrier, one cache line each (see sc_arrive)
  u32 l1_arr[8 * 16];  // ... of the sums' arrival ticket
  u64 slots[4 * LF_SC_GRID_WGS];  // per workgroup {a0, a2}
};
struct ScGrid {
  int field;
  uint2* hcA;       // current HQUAD
  elt_t* vcA;
  uint2* hcB;       // the other half of the ping-pong
  elt_t* vcB;
  u32 nh;
  const u32* d_nh;  // non-null: the HQUAD size is read from this device word (bind_g enqueued just before)
  elt_t* W[2];      // hand arrays at entry
  u32 nW[2];
  elt_t* Wb[2][2];  // bind destinations per hand (ping-pong), (nw+1)/2 elements each
  u64* QW;
  u64* QW2;         // second accumulator array (same size)
  u32 rh0, rh1;     // round-hands [rh0, rh1), rh1 = 2 * logw
  u64 seq0, timeout_ticks;
  volatile u64* post;
  const volatile u64* cmd;
  ScGridSync* gs;
  u32* counts;      // one word per workgroup
  u32* src;         // one word per HQUAD entry: where each bound entry comes from
  u32 tail_lds;     // 1: the launch reserved SC_TAIL_LDS_BYTES of dynamic LDS for the tail
  u32 all_poll;     // 1: every workgroup polls the host for the challenge (A/B)
  u32 two_level;    // 1: two-level arrival tickets (A/B)
  u32 split_waves;  // 1: with one workgroup left, independent products go to different waves
  u32 per_wg;       // entries of the largest array per active workgroup (the grid shrinks to keep it)
  u32 wave_tail;    // 1: once everything fits 64 entries, ONE wave finishes the layer (no workgroup barriers)
  u32* off_cache;   // ScGridOffCache::d
  u32 off_mode;     // 0 no cache, 1 record, 2 replay
  u64 place_ticks;  // how long the FIRST barrier waits: it is the check that all workgroups were placed together
  u32 test_drop;    // test only: the last workgroup leaves at once, as if it had never been placed
};
#define SC_WAVE_TAIL 64u
#define SC_TAIL 1024u
#define SC_TAIL_LDS_BYTES (2 * SC_TAIL * 8 + 2 * SC_TAIL * 16 + 4 * SC_TAIL * 16 + SC_TAIL * 32 + SC_TAIL * 4)

// Two-level arrival ticket.  Device-scope atomics on ONE address serialise (~80 ns each, tools/ubench_sync.hip), so
// workgroups g, g + 8, g + 16, ... share one of eight first-level counters (eight addresses take their atomics in
// parallel) and the last arrival of each group takes a ticket of the second level: G/8 + 8 serialised atomics instead
// of G.  True for the one workgroup that arrives last overall; both levels are acquire-release, so what every
// workgroup wrote before arriving is visible to it.  The counters reset themselves.
__device__ __forceinline__ bool sc_arrive(u32* lvl1, u32* lvl2, u32 G, u32 g, bool two_level) {
  const u32 grp = two_level ? (g & 7) : 0, ngrp = two_level ? (G < 8 ? G : 8) : 1, members = two_level ? ((G - grp + 7) >> 3) : G;
  u32* c1 = lvl1 + grp * 16;
  if (__hip_atomic_fetch_add(c1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != members - 1) return false;
  __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (__hip_atomic_fetch_add(lvl2, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) != ngrp - 1) return false;
  __hip_atomic_store(lvl2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// barrier among the first `G` workgroups, waiting at most `limit_ticks`; 0 = passed, 1 = aborted by another workgroup,
// 2 = THIS workgroup's wait ran out and it raised the abort flag first (every caller returns on non-zero)
__device__ __forceinline__ int sc_grid_barrier_ex(ScGridSync* gs, u32 G, u32& gen, u64 limit_ticks, bool two_level = true) {
  __shared__ u32 s_abort;
  if (G == 1) {
    __threadfence_block();
    __syncthreads();
    return 0;
  }
  // every storing wave: its stores are acknowledged before thread 0 releases for them (a workgroup barrier orders nothing in
  // memory, and the agent-scope release below writes back what has reached the L2)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // one release (with the arrival), one acquire (after the wait): every further fence is an L2 write-back or
    // invalidate of this XCD that the barrier's latency would pay for nothing
    u32 ab = 0;
    const u64 t0 = wall_clock64();
    if (sc_arrive(gs->l1_bar, &gs->count, G, blockIdx.x, two_level)) {
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
__device__ __forceinline__ bool sc_grid_barrier(ScGridSync* gs, u32 G, u32& gen, u64 timeout_ticks, bool two_level = true) {
  return sc_grid_barrier_ex(gs, G, gen, 2 * timeout_ticks, two_level) == 0;
}

template <int F>
__global__ __launch_bounds__(SM_THREADS) void sc_grid_layer_kernel(ScGrid a) {
  __shared__ ScShared sh;
  __shared__ u32 s_off, s_tot, s_last;
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 g = blockIdx.x;
  u32 G = gridDim.x;  // active workgroups: shrinks with the data
  const u32 qwords = F == FIELD_GF2_128 ? 2u : 4u;
  ScGridSync* gs = a.gs;
  const uint2* hc = a.hcA;
  const elt_t* vc = a.vcA;
  uint2* hc_o = a.hcB;
  elt_t* vc_o = a.vcB;
  u32 nh = a.d_nh ? *a.d_nh : a.nh;
  const elt_t* W[2] = {a.W[0], a.W[1]};
  u32 nW[2] = {a.nW[0], a.nW[1]};
  u32 wsel[2] = {a.W[0] == a.Wb[0][0] ? 1u : 0u, a.W[1] == a.Wb[1][0] ? 1u : 0u};
  // pointers the tail may redirect into LDS (generic address space from here on)
  u64* QW = a.QW;    // accumulators of the evaluation being summed
  u64* QWn = a.QW2;  // of the next evaluation: filled while the current hand is bound (one buffer once in LDS)
  u32* src = a.src;
  elt_t* Wdst[2][2] = {{a.Wb[0][0], a.Wb[0][1]}, {a.Wb[1][0], a.Wb[1][1]}};
  bool in_lds = false, wave_mode = false;
  extern __shared__ __attribute__((aligned(16))) unsigned char sc_dyn[];
  u32 gen = 0;
  u64 seq = a.seq0;
  const bool split_waves = a.split_waves != 0;
  // QW[key] += t for one term per lane; every lane of the wave calls it (valid = false for the idle ones)
  auto qw_add = [&](u64* Q, u32 key, elt_t t, bool valid) {
    if (F == FIELD_GF2_128) {  // fold runs of equal targets inside the wave first (see qw_scatter_gf_kernel)
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
      if (valid && head) {
        atomicXor(&Q[2 * (size_t)key], t.lo);
        atomicXor(&Q[2 * (size_t)key + 1], t.hi);
      }
    } else if (valid) {  // integer limb accumulators (see qw_scatter_fp_kernel)
      u64* acc = Q + 4 * (size_t)key;
      atomicAdd(&acc[0], (u64)(u32)t.lo);
      atomicAdd(&acc[1], t.lo >> 32);
      atomicAdd(&acc[2], (u64)(u32)t.hi);
      atomicAdd(&acc[3], t.hi >> 32);
    }
  };
  {
    const u32 GT = G * SM_THREADS, gtid = (wave * G + g) * 64 + lane;  // 64-entry chunks dealt round-robin: few entries = few waves on EVERY workgroup
    const u32 h0 = a.rh0 & 1;
    for (u32 i = gtid; i < qwords * nW[h0]; i += GT) QW[i] = 0;
    for (u32 i = gtid; i < qwords * nW[1 - h0]; i += GT) QWn[i] = 0;
  }
  // Placement check.  Nothing but scratch (QW) has been written yet: when not every workgroup shows up within place_ticks --
  // the CU budget (lf_cu_acquire) rules that out inside one process, another process holding CUs does not -- the workgroup whose wait
  // runs out first reports status 2 and all leave; the host then drives this round-hand with per-launch kernels on the untouched state.
  if (a.test_drop && G > 1 && g == G - 1) return;
  {
    const int br = sc_grid_barrier_ex(gs, G, gen, a.place_ticks, a.two_level != 0);
    if (br) {
      if (br == 2 && tid == 0) {
        a.post[8] = 2;
        __threadfence_system();
        __hip_atomic_store((u64*)&a.post[5], a.seq0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
  {  // first evaluation of the hand-off: QW[h[hand]] += v * Wother[h[1-hand]] over the whole HQUAD
    const int hand = (int)(a.rh0 & 1);
    const u32 GT = G * SM_THREADS;
    const elt_t* Wo = W[1 - hand];
    for (u32 base = (wave * G + g) * 64; base < (nh + 63) / 64 * 64; base += GT) {  // whole waves: qw_add shuffles
      const u32 i = base + lane;
      const bool valid = i < nh;
      u32 key = 0xffffffffu;
      elt_t t = elt_zero();
      if (valid) {
        const uint2 h = hc[i];
        key = hand ? h.y : h.x;
        t = Fld<F>::mul(ld16(&vc[i]), ld16(&Wo[hand ? h.x : h.y]));
      }
      qw_add(QW, key, t, valid);
    }
  }
#ifdef LF_SC_PROF
  u64 pt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // [8]: round-hands counted
  u64 tl = wall_clock64();
#ifdef LF_SC_PROF_LO  // only the round-hands whose largest array holds (LF_SC_PROF_LO, LF_SC_PROF_HI] entries
#define SC_LAP(k) do { const u64 tn_ = wall_clock64(); const u32 bg_ = max(nh, max(nW[0], nW[1])); if (bg_ > LF_SC_PROF_LO && bg_ <= LF_SC_PROF_HI) { pt[k] += tn_ - tl; if (k == 1) ++pt[8]; } tl = tn_; } while (0)
#else
#define SC_LAP(k) do { const u64 tn_ = wall_clock64(); pt[k] += tn_ - tl; if (k == 1) ++pt[8]; tl = tn_; } while (0)
#endif
#else
#define SC_LAP(k) do { } while (0)
#endif
  // One round-hand = [barrier] sums a0, a2 -> post | layout of the HQUAD bind (hidden behind the host) | challenge |
  // bind both structures AND accumulate the next evaluation's QW from the values just produced.
  for (u32 rh = a.rh0; rh < a.rh1; ++rh, ++seq) {
    const int hand = (int)(rh & 1);
    if (!sc_grid_barrier(gs, G, gen, a.timeout_ticks, a.two_level != 0)) return;  // QW of this evaluation is complete
    SC_LAP(0);
    {  // shrink: one workgroup per 1024 entries of the largest array; the others are done
      u32 big = nh > nW[0] ? nh : nW[0];
      big = big > nW[1] ? big : nW[1];
      u32 want = (big + a.per_wg - 1) / a.per_wg;
      want = want ? want : 1;
      if (want < G) G = want;
      if (g >= G) return;
      // tail: once everything fits (<= SC_TAIL entries) the last workgroup moves the whole state into LDS and the
      // remaining rounds never touch global memory for data: every phase is then a product plus an LDS round trip
      // instead of a product plus a global-memory round trip
      if (G == 1 && !in_lds && a.tail_lds && big <= SC_TAIL) {
        uint2* hcL0 = (uint2*)sc_dyn;                              // 2 x SC_TAIL corner pairs
        uint2* hcL1 = hcL0 + SC_TAIL;
        elt_t* vcL0 = (elt_t*)(hcL1 + SC_TAIL);                    // 2 x SC_TAIL values
        elt_t* vcL1 = vcL0 + SC_TAIL;
        elt_t* wL = vcL1 + SC_TAIL;                                // per hand: current (SC_TAIL) + two bind destinations (SC_TAIL / 2 each)
        u64* qwL = (u64*)(wL + 4 * SC_TAIL);                        // SC_TAIL targets x up to 4 words
        for (u32 i = tid; i < nh; i += SM_THREADS) {
          hcL0[i] = hc[i];
          st16(&vcL0[i], ld16(&vc[i]));
        }
        for (int h = 0; h < 2; ++h)
          for (u32 i = tid; i < nW[h]; i += SM_THREADS) st16(&wL[h * 2 * SC_TAIL + i], ld16(&W[h][i]));
        for (u32 i = tid; i < qwords * nW[hand]; i += SM_THREADS) qwL[i] = QW[i];  // the sums of this evaluation
        __syncthreads();
        hc = hcL0; vc = vcL0; hc_o = hcL1; vc_o = vcL1;
        for (int h = 0; h < 2; ++h) {
          W[h] = wL + h * 2 * SC_TAIL;
          Wdst[h][0] = wL + h * 2 * SC_TAIL + SC_TAIL;
          Wdst[h][1] = wL + h * 2 * SC_TAIL + SC_TAIL + SC_TAIL / 2;
          wsel[h] = 0;
        }
        QW = qwL;
        QWn = qwL;
        src = (u32*)(qwL + 4 * SC_TAIL);
        in_lds = true;
      }
    }
    // ---- the last rounds of a layer: at most 64 entries in every array.  A lone wave issues one instruction at a time, so
    // what a round-hand costs here is the number of instructions on its critical path plus a workgroup barrier (sixteen
    // waves to collect) per phase.  From here on wave 0 finishes the layer alone: both sums in ONE product (lanes 0-31
    // the a0 terms, lanes 32-63 the a2 terms), both binds in one product when they fit the wave together, the layout by
    // ballot, reductions by shuffles -- no workgroup barrier is left (the other waves have ended; a barrier of one wave
    // is a wait for its own memory operations).  Same field operations on the same operands as the phases below.
    if (G == 1 && in_lds && a.wave_tail && !wave_mode && nh <= SC_WAVE_TAIL && nW[0] <= SC_WAVE_TAIL && nW[1] <= SC_WAVE_TAIL) {
      if (wave != 0) return;  // the barrier at the top of the loop was their last one: everything they wrote is visible
      wave_mode = true;
    }
    if (wave_mode) {
      const u32 nq = nW[hand], nodd = nq / 2;
      const elt_t* Wh = W[hand];
      auto qw_at = [&](u32 j) -> elt_t {
        if (F == FIELD_GF2_128) return elt_t{QW[2 * (size_t)j], QW[2 * (size_t)j + 1]};
        const u64* q = QW + 4 * (size_t)j;
        return fp_reduce_limbs(q[0], q[1], q[2], q[3]);
      };
      elt_t x = elt_zero(), y = elt_zero();
      {
        const u32 i = lane & 31;
        if (i < nodd) {
          const elt_t q0 = qw_at(2 * i), w0 = ld16(&Wh[2 * i]);
          if (lane < 32) {
            x = q0;
            y = w0;
          } else {
            x = Fld<F>::sub(qw_at(2 * i + 1), q0);
            y = Fld<F>::sub(ld16(&Wh[2 * i + 1]), w0);
          }
        } else if (i == nodd && 2 * nodd < nq) {  // odd tail (prover_layers.h:381-388): in both sums
          x = qw_at(2 * nodd);
          y = ld16(&Wh[2 * nodd]);
        }
      }
      elt_t t = Fld<F>::mul(x, y);
      for (int off = 16; off > 0; off >>= 1) {  // sums inside each half of the wave
        elt_t o;
        o.lo = __shfl_down(t.lo, off, 32);
        o.hi = __shfl_down(t.hi, off, 32);
        t = Fld<F>::add(t, o);
      }
      const elt_t a2{__shfl(t.lo, 32, 64), __shfl(t.hi, 32, 64)};
      if (lane == 0) {
        u64* po = (u64*)a.post;
        __hip_atomic_store(&po[0], t.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[1], t.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[2], a2.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[3], a2.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[4], (u64)nh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[8], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lf_wait_stores_before_publish();  // payload acknowledged before the sequence word leaves
        __hip_atomic_store(&po[5], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      // layout of HQuad::bind_h (no challenge needed): lane i looks at entry i
      const bool head = lane < nh && !is_second(hc, lane, hand);
      const u64 hmask = __ballot(head);
      const u32 new_nh = (u32)__popcll(hmask);
      if (head) {
        const u32 off = (u32)__popcll(hmask & ((1ull << lane) - 1));
        uint2 h = hc[lane];
        const u32 hh = hand ? h.y : h.x;
        const u32 kind = (lane + 1 < nh && is_second(hc, lane + 1, hand)) ? 0u : ((hh & 1) == 0 ? 1u : 2u);
        if (hand) h.y = hh >> 1; else h.x = hh >> 1;
        hc_o[off] = h;
        src[off] = lane | (kind << 30);
      }
      // the challenge (see below for the protocol)
      u64 w = 0;
      {
        const u64 t0 = wall_clock64();
        const u64 tag = seq & 0xffffffffull;
        for (;;) {
          if (lane < 4) w = __hip_atomic_load((const u64*)&a.cmd[4 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (__all(lane >= 4 || (w >> 32) == tag)) break;
          int stop = 0;
          if (lane == 0 && wall_clock64() - t0 > a.timeout_ticks) {  // the host went away: report and leave
            a.post[8] = 1;
            __threadfence_system();
            __hip_atomic_store((u64*)&a.post[5], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            stop = 1;
          }
          if (__shfl(stop, 0, 64)) return;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      const u64 c0 = __shfl(w, 0, 64), c1 = __shfl(w, 1, 64), c2 = __shfl(w, 2, 64), c3 = __shfl(w, 3, 64);
      const elt_t r{(c0 & 0xffffffffull) | (c1 << 32), (c2 & 0xffffffffull) | (c3 << 32)};
      // this evaluation's sums are spent: clear for the next one (one buffer in LDS)
      for (u32 i = lane; i < qwords * nW[1 - hand]; i += 64) QW[i] = 0;
      __syncthreads();  // one wave: a wait for the stores above (layout, clear)
      // Dense::bind of W[hand] and the HQuad::bind_h values: out = base + d * r for both (hquad.h:94-118, dense.h bind)
      const elt_t* Wold = W[hand];
      const u32 n0 = nW[hand], nout = (n0 + 1) / 2;
      elt_t* const Wout = Wdst[hand][wsel[hand]];
      const bool merged = new_nh + nout <= 64;  // both binds in one pass: HQUAD values on the low lanes, the hand array on the high ones
      elt_t vbound = elt_zero();
      for (int pass = 0; pass < (merged ? 1 : 2); ++pass) {
        elt_t base = elt_zero(), d = elt_zero();
        int job = 0;  // 1: HQUAD value `lane`, 2: hand entry j
        u32 j = 0;
        if (pass == 0 && lane < new_nh) {
          job = 1;
          const u32 sidx = src[lane], i = sidx & 0x3fffffffu, kind = sidx >> 30;
          const elt_t v0 = ld16(&vc[i]);
          if (kind == 0) { base = v0; d = Fld<F>::sub(ld16(&vc[i + 1]), v0); }
          else if (kind == 1) { base = v0; d = Fld<F>::sub(elt_zero(), v0); }
          else { d = v0; }
        } else if (merged ? lane >= 64 - nout : (pass == 1 && lane < nout)) {
          job = 2;
          j = merged ? lane - (64 - nout) : lane;
          const elt_t f0 = ld16(&Wold[2 * j]);
          base = f0;
          d = 2 * j + 1 < n0 ? Fld<F>::sub(ld16(&Wold[2 * j + 1]), f0) : Fld<F>::sub(elt_zero(), f0);
        }
        const elt_t out = Fld<F>::add(base, Fld<F>::mul(d, r));
        if (job == 1) {
          vbound = out;
          st16(&vc_o[lane], out);
        } else if (job == 2) {
          st16(&Wout[j], out);
        }
      }
      __syncthreads();
      if (rh + 1 < a.rh1) {  // the next evaluation's sums (for the other hand)
        const bool valid = lane < new_nh;
        u32 key = 0xffffffffu;
        elt_t tt = elt_zero();
        if (valid) {
          const uint2 h = hc_o[lane];
          key = hand ? h.x : h.y;
          tt = Fld<F>::mul(vbound, ld16(&Wout[hand ? h.y : h.x]));
        }
        qw_add(QWn, key, tt, valid);
      }
      W[hand] = Wout;
      nW[hand] = nout;
      wsel[hand] ^= 1;
      {
        nh = new_nh;
        uint2* th = const_cast<uint2*>(hc);
        elt_t* tv = const_cast<elt_t*>(vc);
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
    const u32 GT = G * SM_THREADS, gtid = (wave * G + g) * 64 + lane;  // 64-entry chunks dealt round-robin: few entries = few waves on EVERY workgroup
    SC_LAP(1);
    // ---- ProverLayers::evaluations: a0, a2
    {
      const u32 nq = nW[hand], nodd = nq / 2;
      const elt_t* Wh = W[hand];
      auto qw_at = [&](u32 j) -> elt_t {
        if (F == FIELD_GF2_128) return elt_t{QW[2 * (size_t)j], QW[2 * (size_t)j + 1]};
        const u64* q = QW + 4 * (size_t)j;
        return fp_reduce_limbs(q[0], q[1], q[2], q[3]);
      };
      elt_t a0 = elt_zero(), a2 = elt_zero();
      if (G == 1 && split_waves) {  // the two products of a pair on different waves (see the bind phase): even waves sum a0, odd waves a2
        const u32 half = SM_THREADS / 2, ht = (wave >> 1) * 64 + lane;
        const bool second = (wave & 1) != 0;
        for (u32 i = ht; i < nodd; i += half) {
          const elt_t q0 = qw_at(2 * i), w0 = ld16(&Wh[2 * i]);
          if (!second) a0 = Fld<F>::add(a0, Fld<F>::mul(q0, w0));
          else a2 = Fld<F>::add(a2, Fld<F>::mul(Fld<F>::sub(qw_at(2 * i + 1), q0), Fld<F>::sub(ld16(&Wh[2 * i + 1]), w0)));
        }
        if (ht == 0 && 2 * nodd < nq) {  // odd tail (prover_layers.h:381-388): in both sums
          const elt_t t = Fld<F>::mul(qw_at(2 * nodd), ld16(&Wh[2 * nodd]));
          if (!second) a0 = Fld<F>::add(a0, t);
          else a2 = Fld<F>::add(a2, t);
        }
      } else {
        for (u32 i = gtid; i < nodd; i += GT) {
          const elt_t q0 = qw_at(2 * i), q1 = qw_at(2 * i + 1);
          const elt_t w0 = ld16(&Wh[2 * i]), w1 = ld16(&Wh[2 * i + 1]);
          a0 = Fld<F>::add(a0, Fld<F>::mul(q0, w0));
          a2 = Fld<F>::add(a2, Fld<F>::mul(Fld<F>::sub(q1, q0), Fld<F>::sub(w1, w0)));
        }
        if (gtid == 0 && 2 * nodd < nq) {  // odd tail (prover_layers.h:381-388)
          const elt_t t = Fld<F>::mul(qw_at(2 * nodd), ld16(&Wh[2 * nodd]));
          a0 = Fld<F>::add(a0, t);
          a2 = Fld<F>::add(a2, t);
        }
      }
      // workgroup sum (wave shuffles, then the 16 wave sums through LDS)
      auto wg_sum = [&](elt_t& x0, elt_t& x2) {
        for (int off = 32; off > 0; off >>= 1) {
          elt_t o0, o2;
          o0.lo = __shfl_down(x0.lo, off, 64); o0.hi = __shfl_down(x0.hi, off, 64);
          o2.lo = __shfl_down(x2.lo, off, 64); o2.hi = __shfl_down(x2.hi, off, 64);
          x0 = Fld<F>::add(x0, o0);
          x2 = Fld<F>::add(x2, o2);
        }
        if (lane == 0) {
          sh.red[0][wave] = x0;
          sh.red[1][wave] = x2;
        }
        __syncthreads();
        if (tid == 0) {
          x0 = sh.red[0][0];
          x2 = sh.red[1][0];
          for (u32 w = 1; w < SM_THREADS / 64; ++w) {
            x0 = Fld<F>::add(x0, sh.red[0][w]);
            x2 = Fld<F>::add(x2, sh.red[1][w]);
          }
        }
        __syncthreads();
      };
      wg_sum(a0, a2);
      bool poster = tid == 0;  // thread 0 of the workgroup that holds the device-wide sums
      if (G > 1) {  // slots + arrival ticket: the last workgroup to arrive folds all slots
        if (tid == 0) {
          u64* sl = &gs->slots[4 * g];
          sl[0] = a0.lo; sl[1] = a0.hi; sl[2] = a2.lo; sl[3] = a2.hi;
          s_last = sc_arrive(gs->l1_arr, &gs->arrive, G, g, a.two_level != 0) ? 1u : 0u;  // releases the slot
        }
        __syncthreads();
        poster = false;
        if (s_last) {  // uniform per workgroup
          a0 = elt_zero();
          a2 = elt_zero();
          if (tid < G) {
            const u64* sl = &gs->slots[4 * tid];
            a0 = elt_t{__hip_atomic_load(&sl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&sl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
            a2 = elt_t{__hip_atomic_load(&sl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&sl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
          }
          wg_sum(a0, a2);
          poster = tid == 0;
        }
      }
      if (poster) {
        // the post words live in uncached pinned host memory: write them with system-scope stores, wait until they
        // have been acknowledged (s_waitcnt vmcnt(0): no L2 write-back), then publish the sequence number
        u64* po = (u64*)a.post;
        __hip_atomic_store(&po[0], a0.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[1], a0.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[2], a2.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[3], a2.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[4], (u64)nh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&po[8], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        lf_wait_stores_before_publish();  // payload acknowledged before the sequence word leaves
        __hip_atomic_store(&po[5], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    SC_LAP(2);
    // ---- while the host works on the challenge: everything of HQuad::bind_h that does not depend on it -- which
    // entries merge (pair / lone even / lone odd), where each result goes, the halved corner indices
    u32 my_off, my_end, new_nh;
    {
      const u32 R = ((nh + G - 1) / G + 63) / 64 * 64;  // range per workgroup, whole waves
      const u32 lo = (u64)g * R < nh ? g * R : nh, hi = (u64)lo + R < nh ? lo + R : nh;
      if (tid == 0) {
        sh.carry = 0;
        s_off = 0;
        s_tot = 0;
      }
      __syncthreads();
      u32* const oc = a.off_cache + (size_t)rh * (LF_SC_GRID_WGS + 1);
      if (G > 1 && a.off_mode == 2) {  // recorded by an earlier proof (ScGridOffCache): no count, no barrier
        if (tid == 0) {
          s_off = oc[g];
          s_tot = oc[LF_SC_GRID_WGS];
        }
        __syncthreads();
      } else if (G > 1) {
        u32 mine = 0;
        for (u32 base = lo; base < hi; base += SM_THREADS) {
          const u32 i = base + tid;
          const bool head = i < hi && !is_second(hc, i, hand);
          mine += (u32)__popcll(__ballot(head));
        }
        if (lane == 0 && mine) atomicAdd(&sh.carry, mine);
        __syncthreads();
        if (tid == 0) a.counts[g] = sh.carry;
        SC_LAP(3);
        if (!sc_grid_barrier(gs, G, gen, a.timeout_ticks, a.two_level != 0)) return;
        SC_LAP(4);
        if (tid < G) {
          const u32 cnt = a.counts[tid];
          if (cnt) {
            atomicAdd(&s_tot, cnt);
            if (tid < g) atomicAdd(&s_off, cnt);
          }
        }
        __syncthreads();
        if (a.off_mode == 1 && tid == 0) {
          oc[g] = s_off;
          if (g == 0) oc[LF_SC_GRID_WGS] = s_tot;
        }
      }
      my_off = s_off;
      __syncthreads();
      if (tid == 0) sh.carry = my_off;
      __syncthreads();
      for (u32 base = lo; base < hi; base += SM_THREADS) {
        const u32 i = base + tid;
        const bool head = i < hi && !is_second(hc, i, hand);
        const u64 mask = __ballot(head);
        if (lane == 0) sh.wave[wave] = (u32)__popcll(mask);
        __syncthreads();
        u32 off = sh.carry;
        for (u32 w = 0; w < wave; ++w) off += sh.wave[w];
        off += (u32)__popcll(mask & ((1ull << lane) - 1));
        if (head) {
          uint2 h = hc[i];
          const u32 hh = hand ? h.y : h.x;
          const u32 kind = (i + 1 < nh && is_second(hc, i + 1, hand)) ? 0u : ((hh & 1) == 0 ? 1u : 2u);
          if (hand) h.y = hh >> 1; else h.x = hh >> 1;
          hc_o[off] = h;
          src[off] = i | (kind << 30);
        }
        __syncthreads();
        if (tid == 0) {
          u32 tot = 0;
          for (u32 w = 0; w < SM_THREADS / 64; ++w) tot += sh.wave[w];
          sh.carry += tot;
        }
        __syncthreads();
      }
      my_end = sh.carry;
      new_nh = G > 1 ? s_tot : my_end;
    }
    SC_LAP(5);
    // ---- the challenge.  The host writes it as four 64-bit words {tag = low half of the sequence number, 32 bits of
    // the challenge}; 64-bit stores are atomic, so a read that finds the tag in all four words has the whole challenge
    // and needs no second PCIe round trip for the payload after a flag.  Lanes 0-3 of wave 0 read one word each in ONE
    // vector load (one request for the 32 bytes).  Workgroup 0 polls the host and hands the challenge to the others
    // through a device slot (all_poll: every workgroup polls the host itself).
    if (wave == 0) {
      const u64 t0 = wall_clock64();
      const u64 tag = seq & 0xffffffffull;
      const bool from_host = g == 0 || a.all_poll;
      u64 got = 0, w = 0;
      for (;;) {
        if (from_host) {
          if (lane < 4) w = __hip_atomic_load((const u64*)&a.cmd[4 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else {
          if (lane < 4) w = __hip_atomic_load(&gs->chal4[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__all(lane >= 4 || (w >> 32) == tag)) {
          got = seq;
          break;
        }
        int stop = 0;
        if (lane == 0) {
          if (G > 1 && __hip_atomic_load(&gs->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) stop = 1;
          else if (wall_clock64() - t0 > (from_host ? 1 : 2) * a.timeout_ticks) {  // the host went away: release every workgroup and report
            __hip_atomic_store(&gs->abort, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (g == 0) {
              a.post[8] = 1;
              __threadfence_system();
              __hip_atomic_store((u64*)&a.post[5], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            stop = 1;
          }
        }
        if (__shfl(stop, 0, 64)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (got == seq && g == 0 && G > 1 && !a.all_poll && lane < 4)  // tagged words again: no flag, no fence
        __hip_atomic_store(&gs->chal4[lane], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 c0 = __shfl(w, 0, 64), c1 = __shfl(w, 1, 64), c2 = __shfl(w, 2, 64), c3 = __shfl(w, 3, 64);
      if (lane == 0) {
        sh.cmd[0] = (c0 & 0xffffffffull) | (c1 << 32);
        sh.cmd[1] = (c2 & 0xffffffffull) | (c3 << 32);
        sh.cmd[2] = got;
      }
    }
    __syncthreads();
    if (sh.cmd[2] != seq) return;  // uniform per workgroup; the barriers of the others see the abort flag
    const elt_t r{sh.cmd[0], sh.cmd[1]};
    SC_LAP(6);
    // ---- Dense::bind of W[hand] (out of place); HQuad::bind_h values for the layout above; and, from those values,
    // the next evaluation's sums QWn[h'[other]] += v' * bind(W[hand])[h'[hand]] -- the bound hand entry is recomputed
    // from the unbound array (one more product) so that no workgroup waits for another one's Dense::bind
    const elt_t* Wold = W[hand];
    const u32 n0 = nW[hand], nout = (n0 + 1) / 2;
    auto bind_at = [&](u32 j) -> elt_t {
      const elt_t f0 = ld16(&Wold[2 * j]);
      if (2 * j + 1 < n0) return Fld<F>::add(f0, Fld<F>::mul(Fld<F>::sub(ld16(&Wold[2 * j + 1]), f0), r));
      return Fld<F>::sub(f0, Fld<F>::mul(f0, r));
    };
    const bool more = rh + 1 < a.rh1;
    if (QWn == QW) {  // one buffer (LDS tail, a single workgroup): this evaluation's sums are spent, clear for the next
      for (u32 i = tid; i < qwords * nW[1 - hand]; i += SM_THREADS) QW[i] = 0;
      __syncthreads();
    } else {  // two buffers: this one is next written two round-hands from now, for this hand again
      for (u32 i = gtid; i < qwords * nout; i += GT) QW[i] = 0;
    }
    auto bind_value = [&](u32 o) -> elt_t {  // HQuad::bind_h value of output o (hquad.h:94-118)
      const u32 sidx = src[o], i = sidx & 0x3fffffffu, kind = sidx >> 30;
      const elt_t v0 = ld16(&vc[i]);
      if (kind == 0) return Fld<F>::add(v0, Fld<F>::mul(Fld<F>::sub(ld16(&vc[i + 1]), v0), r));
      if (kind == 1) return Fld<F>::sub(v0, Fld<F>::mul(v0, r));
      return Fld<F>::mul(v0, r);
    };
    elt_t* const Wout = Wdst[hand][wsel[hand]];
    if (G == 1 && split_waves) {
      // One workgroup left: what counts is the longest chain of products a single wave issues (a wave issues one
      // instruction at a time, ~1.5 us per GF(2^128) product), so the independent products go to different waves --
      // the even waves bind the HQUAD values while the odd waves bind the hand array -- and, a workgroup barrier being cheap,
      // the next evaluation's sums take the bound hand entries from memory instead of recomputing them.
      const u32 half = SM_THREADS / 2, ht = (wave >> 1) * 64 + lane;  // even / odd waves sit on different SIMDs
      if ((wave & 1) == 0) {
        for (u32 o = ht; o < my_end; o += half) st16(&vc_o[o], bind_value(o));
      } else {
        for (u32 i = ht; i < nout; i += half) st16(&Wout[i], bind_at(i));
      }
      __threadfence_block();
      __syncthreads();
      if (more) {
        for (u32 base = 0; base < my_end; base += SM_THREADS) {
          const u32 o = base + tid;
          const bool valid = o < my_end;
          u32 key = 0xffffffffu;
          elt_t t = elt_zero();
          if (valid) {
            const uint2 h = hc_o[o];
            key = hand ? h.x : h.y;  // the next evaluation is for the other hand
            t = Fld<F>::mul(ld16(&vc_o[o]), ld16(&Wout[hand ? h.y : h.x]));
          }
          qw_add(QWn, key, t, valid);
        }
      }
    } else {
      for (u32 i = gtid; i < nout; i += GT) st16(&Wout[i], bind_at(i));
      for (u32 base = my_off; base < my_end; base += SM_THREADS) {
        const u32 o = base + tid;
        const bool valid = o < my_end;
        u32 key = 0xffffffffu;
        elt_t t = elt_zero();
        if (valid) {
          const elt_t v = bind_value(o);
          st16(&vc_o[o], v);
          if (more) {
            const uint2 h = hc_o[o];
            key = hand ? h.x : h.y;  // the next evaluation is for the other hand
            t = Fld<F>::mul(v, bind_at(hand ? h.y : h.x));
          }
        }
        if (more) qw_add(QWn, key, t, valid);
      }
    }
    W[hand] = Wout;
    nW[hand] = nout;
    wsel[hand] ^= 1;
    {
      nh = new_nh;
      uint2* th = const_cast<uint2*>(hc);
      elt_t* tv = const_cast<elt_t*>(vc);
      hc = hc_o;
      vc = vc_o;
      hc_o = th;
      vc_o = tv;
      u64* tq = QW;
      QW = QWn;
      QWn = tq;
    }
    SC_LAP(7);
  }
#ifdef LF_SC_PROF
  if (g == 0 && tid == 0)
    for (int k = 0; k < 9; ++k) a.post[16 + k] = pt[k];
#endif
  if (g == 0 && tid == 0) {  // end of the layer: W[R,C], W[L,C] and HQUAD->scalar()
    const elt_t w0 = nW[0] ? ld16(&W[0][0]) : elt_zero();
    const elt_t w1 = nW[1] ? ld16(&W[1][0]) : elt_zero();
    const elt_t sc = nh ? ld16(&vc[0]) : elt_zero();
    a.post[0] = w0.lo; a.post[1] = w0.hi; a.post[2] = w1.lo; a.post[3] = w1.hi;
    a.post[4] = nh;
    a.post[6] = sc.lo; a.post[7] = sc.hi;
    a.post[8] = 0;
    __threadfence_system();
    __hip_atomic_store((u64*)&a.post[5], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// waits until the post carries `seq`; polls coherent memory and watches the stream so a dead kernel is noticed
static int sc_wait_post(lfgpu_ctx* c, u64 seq) {
  // The stream is only looked at after 50 ms without the post (a dead kernel must not hang the caller): in a healthy run no HIP
  // call is made while a resident kernel waits for this thread.  It matters with several provers on one device -- another
  // thread's hipFree / hipMalloc holds the runtime's lock while it waits for every stream of the device, hence for that kernel;
  // a hipStreamQuery here would then wait for the lock and the kernel for its challenge, until its timeout.
  u64 spins = 0;
  double t_first = 0;
  while (__atomic_load_n((const u64*)&c->poll_h[5], __ATOMIC_ACQUIRE) != seq) {
    // far longer than a round-hand without a post: this thread's device is busy with other provers' work (throughput mode with more
    // provers than host cores, or a large kernel ahead in the queue) -- let another thread have the core between polls
    if (spins > 0x8000 && (spins & 0xff) == 0) sched_yield();
    if ((++spins & 0xfff) == 0) {
      const double t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
      if (t_first == 0) t_first = t;
      if (t - t_first < 50.0) continue;
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) {  // the kernel is over: its post must be visible now
        if (__atomic_load_n((const u64*)&c->poll_h[5], __ATOMIC_ACQUIRE) == seq) break;
        return lf_fail(c, LFGPU_ERR_ASSERT, "sumcheck step: kernel finished without posting");
      }
      if (q != hipErrorNotReady) return lf_fail(c, LFGPU_ERR_HIP, "sumcheck step: %s", hipGetErrorString(q));
    }
  }
  if (c->poll_h[8] == 2) return lf_fail(c, LFGPU_ERR_BUSY, "sumcheck grid kernel: its workgroups were not placed together (another tenant holds CUs)");
  if (c->poll_h[8] != 0) return lf_fail(c, LFGPU_ERR_ASSERT, "sumcheck layer kernel: challenge wait timed out");
  return LFGPU_OK;
}

// launches one fused step and waits for its post
int lf_sc_small_step(lfgpu_ctx* c, const ScSmall& a, u64 out[8]) {
  if (a.nh > LF_SC_SMALL_MAX || a.nW[0] > LF_SC_SMALL_MAX || a.nW[1] > LF_SC_SMALL_MAX)
    return lf_fail(c, LFGPU_ERR_ARG, "sc_small_step: operands larger than the single-workgroup bound");
  const u64 seq = ++c->poll_seq;
  if (a.field == LFGPU_FIELD_GF2_128)
    hipLaunchKernelGGL(sc_small_step_kernel<FIELD_GF2_128>, dim3(1), dim3(SM_THREADS), 0, c->stream, a, seq, c->poll_h);
  else
    hipLaunchKernelGGL(sc_small_step_kernel<FIELD_FP128>, dim3(1), dim3(SM_THREADS), 0, c->stream, a, seq, c->poll_h);
  LF_HIP(c, hipGetLastError());
  LF_TRY(sc_wait_post(c, seq));
  for (int i = 0; i < 8; ++i) out[i] = c->poll_h[i];
  return LFGPU_OK;
}

// one round trip of the resident protocol on a 1-thread kernel with a short timeout: decides once per context
// whether coherent pinned memory lets a RUNNING kernel see host writes on this system (otherwise the per-step
// launches are used)
__global__ void sc_handshake_test_kernel(u64 seq, u64 timeout_ticks, volatile u64* post, const volatile u64* cmd) {
  if (threadIdx.x != 0) return;
  post[8] = 0;
  __threadfence_system();
  __hip_atomic_store((u64*)&post[5], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const u64 t0 = wall_clock64();
  u64 ok = 0;
  while (wall_clock64() - t0 <= timeout_ticks) {
    if (__hip_atomic_load((const u64*)&cmd[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == seq) {
      ok = 1;
      break;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  post[8] = ok ? 0 : 1;
  post[0] = ok ? __hip_atomic_load((const u64*)&cmd[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
  __threadfence_system();
  __hip_atomic_store((u64*)&post[5], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
bool lf_sc_resident_ok(lfgpu_ctx* c) {
  if (c->resident_state > 0) return true;
  if (c->resident_state <= -3) return false;  // three failed attempts: this system does not show a running kernel's posts
  --c->resident_state;  // (one failed attempt may be a launch held up by another thread's allocation)
  const u64 seq = c->poll_seq + 1;
  c->poll_seq += 2;
  volatile u64* cmd = c->poll_h + 64;
  // a tool that holds a dispatch back until it has finished makes the first post invisible while the kernel runs: the
  // test then fails cleanly and the per-launch driver is used
  hipLaunchKernelGGL(sc_handshake_test_kernel, dim3(1), dim3(64), 0, c->stream, seq, 200ull * c->wall_khz /*0.2 s*/, c->poll_h,
                     (const volatile u64*)cmd);
  if (hipGetLastError() != hipSuccess) return false;
  bool seen = false;  // the first post must arrive while the kernel is still waiting for us
  {  // no HIP call while the kernel waits for this thread (see sc_wait_post): the clock bounds the wait (the kernel gives up after 0.2 s)
    const auto t0 = std::chrono::steady_clock::now();
    for (u64 spins = 0;; ++spins) {
      if (__atomic_load_n((const u64*)&c->poll_h[5], __ATOMIC_ACQUIRE) == seq) {
        seen = true;
        break;
      }
      if ((spins & 0xfff) == 0xfff && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > 400.0) break;
    }
  }
  if (seen) {
    cmd[0] = 0x5eed5eed5eedull;
    cmd[1] = 0;
    __atomic_store_n((u64*)&cmd[2], seq, __ATOMIC_RELEASE);
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) return false;
  if (seen && c->poll_h[5] == seq + 1 && c->poll_h[8] == 0 && c->poll_h[0] == 0x5eed5eed5eedull) c->resident_state = 1;
  if (getenv("LFGPU_VERBOSE"))
    fprintf(stderr, "lfgpu: resident sumcheck handshake %s (first post seen while running: %d, status %llu)\n",
            c->resident_state > 0 ? "ok" : "unavailable", (int)seen, (unsigned long long)c->poll_h[8]);
  return c->resident_state > 0;
}

// resident variant: launch once for round-hands [rh0, rh1), then lf_sc_layer_next per round-hand and a last call
// for the end-of-layer read-out
int lf_sc_layer_begin(lfgpu_ctx* c, const ScSmall& a, u32 rh0, u32 rh1, void* d_W_shared, void* wtmp) {
  if (a.nh > LF_SC_SMALL_MAX || a.nW[0] > LF_SC_SMALL_MAX || a.nW[1] > LF_SC_SMALL_MAX || rh0 >= rh1)
    return lf_fail(c, LFGPU_ERR_ARG, "sc_layer_begin: bad operands");
  const u64 seq0 = c->poll_seq + 1;
  c->poll_seq += (u64)(rh1 - rh0) + 1;
  c->poll_next = seq0;
  const u64 timeout_ticks = 5000ull * c->wall_khz;  // 5 s of the constant-rate wall clock
  volatile u64* post = c->poll_h;
  const volatile u64* cmd = c->poll_h + 64;
  if (a.field == LFGPU_FIELD_GF2_128)
    hipLaunchKernelGGL(sc_small_layer_kernel<FIELD_GF2_128>, dim3(1), dim3(SM_THREADS), 0, c->stream, a, rh0, rh1, (elt_t*)d_W_shared,
                       (elt_t*)wtmp, seq0, timeout_ticks, post, cmd);
  else
    hipLaunchKernelGGL(sc_small_layer_kernel<FIELD_FP128>, dim3(1), dim3(SM_THREADS), 0, c->stream, a, rh0, rh1, (elt_t*)d_W_shared,
                       (elt_t*)wtmp, seq0, timeout_ticks, post, cmd);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}
// round-hands [rh0, 2*logw) of a layer as one launch on ceil(max size / per_wg) co-resident workgroups;
// d_state: LF_SC_GRID_STATE_BYTES of device scratch (ScGridSync + the per-workgroup counts)
int lf_sc_grid_begin(lfgpu_ctx* c, int field, void* hc_cur, void* vc_cur, void* hc_oth, void* vc_oth, size_t nh, const u32* d_nh, void* W0, size_t nW0,
                     void* W1, size_t nW1, void* Wb00, void* Wb01, void* Wb10, void* Wb11, void* qw, size_t rh0,
                     size_t logw, void* d_state, ScGridOffCache* oc, u32* G_out, u32* per_wg_out, bool state_clean) {
  const size_t big = std::max(nh, std::max(nW0, nW1));
  if (rh0 >= 2 * logw || big > LF_SC_GRID_MAX) return lf_fail(c, LFGPU_ERR_ARG, "sc_grid_begin: bad operands");
  static_assert(sizeof(ScGridSync) + 4 * LF_SC_GRID_WGS + 36 * LF_SC_GRID_MAX <= LF_SC_GRID_STATE_BYTES, "grid state size");
  // entries per active workgroup: a workgroup's share of the products runs on ONE CU, so fewer entries per workgroup
  // buy shorter phases until the barrier among more workgroups costs more (measured: GF(2^128) products are ~4x the
  // Fp128 ones, hence the smaller share)
  static const int per_wg_env = getenv("LFGPU_SC_PER_WG") ? std::max(64, atoi(getenv("LFGPU_SC_PER_WG"))) : 0;
  // Several provers on the device (>= 6 contexts with streams of their own: throughput mode): smaller grids -- 1024 entries
  // per workgroup, at most 24 workgroups -- give each prover a little more latency and the device 10 - 14 % more proofs per
  // second (K = 16 on flatsha-32: 333 -> 379 proofs/s; profiles/r03/README.md); a prover that has the device to itself keeps
  // the latency-optimal shape.  The environment overrides both.
  const bool shared_dev = lf_cu_sharers(c) >= 6;  // measured: at K = 2 and 4 the latency-optimal shape still gives more proofs per second (128 / 237 vs 113 / 216), from K = 8 on the small one (317 vs 301)
  const u32 per_wg = per_wg_env ? (u32)per_wg_env : (field == LFGPU_FIELD_GF2_128 && !shared_dev ? 512u : 1024u);
  u32 G = (u32)((big + per_wg - 1) / per_wg);
  G = G ? G : 1;
  static const u32 wgs_cap_env = getenv("LFGPU_SC_WGS") ? (u32)std::min(LF_SC_GRID_WGS, std::max(1, atoi(getenv("LFGPU_SC_WGS")))) : 0u;
  const u32 wgs_cap = wgs_cap_env ? wgs_cap_env : (shared_dev ? 24u : 64u);
  if (G > wgs_cap) G = wgs_cap;
  if ((int)G > c->num_cu) G = (u32)c->num_cu;
  int& tail_ok = c->sc_tail_ok;  // dynamic LDS for the tail (112 KiB) needs the attribute once per kernel and context
  if (tail_ok < 0) {
    const bool off = getenv("LFGPU_SC_TAIL") && atoi(getenv("LFGPU_SC_TAIL")) == 0;
    tail_ok = !off && hipFuncSetAttribute((const void*)sc_grid_layer_kernel<FIELD_GF2_128>, hipFuncAttributeMaxDynamicSharedMemorySize, SC_TAIL_LDS_BYTES) == hipSuccess &&
                      hipFuncSetAttribute((const void*)sc_grid_layer_kernel<FIELD_FP128>, hipFuncAttributeMaxDynamicSharedMemorySize, SC_TAIL_LDS_BYTES) == hipSuccess
                  ? 1 : 0;
    (void)hipGetLastError();
  }
  {  // all G workgroups must be resident together (they synchronise through device memory): one fits a CU ...
    const void* fn0 = field == LFGPU_FIELD_GF2_128 ? (const void*)sc_grid_layer_kernel<FIELD_GF2_128> : (const void*)sc_grid_layer_kernel<FIELD_FP128>;
    int pc = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn0, SM_THREADS, tail_ok ? SC_TAIL_LDS_BYTES : 0) != hipSuccess) pc = 0;
    if (pc < 1) return LFGPU_ERR_BUSY;  // the caller takes another driver
  }
  // ... and the device's budget has G CUs left for them (ctx.h: the sum over every live grid of the process stays within the
  // CU count, so each gets placed).  Short of that the caller runs this round-hand with per-launch kernels and asks again at
  // the next one, when the arrays -- and the grid -- are half the size.
  // With other provers on the device (throughput mode) a grid takes at most half of what is free, but not less than 8
  // workgroups while the arrays are large: K provers then share the CUs instead of the first few taking 64 each and the rest
  // falling back to per-launch kernels.  Alone on the device nothing changes (half of 256 is more than the cap of 64).
  {
    const u32 avail = (u32)lf_cu_available(c);
    const u32 share = std::max<u32>(8u, avail / 2);
    if (G > share) G = share;
  }
  if (!lf_cu_acquire(c, (int)G)) return LFGPU_ERR_BUSY;
  if (G_out) *G_out = G;
  if (per_wg_out) *per_wg_out = per_wg;
  ScGrid a{};
  a.field = field;
  a.hcA = (uint2*)hc_cur; a.vcA = (elt_t*)vc_cur; a.hcB = (uint2*)hc_oth; a.vcB = (elt_t*)vc_oth;
  a.nh = (u32)nh;  // with d_nh: an upper bound (grid size), the kernel takes the count from the device word
  a.d_nh = d_nh;
  a.W[0] = (elt_t*)W0; a.W[1] = (elt_t*)W1;
  a.nW[0] = (u32)nW0; a.nW[1] = (u32)nW1;
  a.Wb[0][0] = (elt_t*)Wb00; a.Wb[0][1] = (elt_t*)Wb01; a.Wb[1][0] = (elt_t*)Wb10; a.Wb[1][1] = (elt_t*)Wb11;
  a.QW = (u64*)qw;
  a.rh0 = (u32)rh0;
  a.rh1 = (u32)(2 * logw);
  a.seq0 = c->poll_seq + 1;
  c->poll_seq += (2 * logw - rh0) + 1;
  c->poll_next = a.seq0;
  a.timeout_ticks = 5000ull * c->wall_khz;
  static const u64 place_ms = getenv("LFGPU_SC_PLACE_MS") ? (u64)std::max(1, atoi(getenv("LFGPU_SC_PLACE_MS"))) : 250ull;
  a.place_ticks = place_ms * c->wall_khz;
  {  // test hook: the first LFGPU_SC_TEST_DROP grid launches of the process lose their last workgroup (a grid that is not placed whole)
    static std::atomic<int> drops{getenv("LFGPU_SC_TEST_DROP") ? atoi(getenv("LFGPU_SC_TEST_DROP")) : 0};
    if (G > 1 && drops.load() > 0 && drops.fetch_sub(1) > 0) a.test_drop = 1;
  }
  a.post = c->poll_h;
  a.cmd = c->poll_h + 64;
  a.gs = (ScGridSync*)d_state;
  a.counts = (u32*)((uint8_t*)d_state + sizeof(ScGridSync));
  a.src = (u32*)((uint8_t*)d_state + LF_SC_GRID_STATE_BYTES - 36 * LF_SC_GRID_MAX);
  a.QW2 = (u64*)((uint8_t*)d_state + LF_SC_GRID_STATE_BYTES - 32 * LF_SC_GRID_MAX);
  // counters (both levels), abort flag, challenge slot: zero at launch -- cleared on the side by the layer's bind_g emit kernel
  // (state_clean), else here
  if (!state_clean) LF_HIP(c, hipMemsetAsync(d_state, 0, LF_SC_GRID_SYNC_CLEAR_BYTES, c->stream));
  void* args[] = {&a};
  const void* fn = field == LFGPU_FIELD_GF2_128 ? (const void*)sc_grid_layer_kernel<FIELD_GF2_128> : (const void*)sc_grid_layer_kernel<FIELD_FP128>;
  a.tail_lds = (u32)tail_ok;
  a.per_wg = per_wg;
  static const int all_poll_env = getenv("LFGPU_SC_ALLPOLL") ? atoi(getenv("LFGPU_SC_ALLPOLL")) : 0;
  a.all_poll = (u32)all_poll_env;
  static const int two_level_env = getenv("LFGPU_SC_TWOLEVEL") ? atoi(getenv("LFGPU_SC_TWOLEVEL")) : 1;
  a.two_level = (u32)two_level_env;
  static const int split_env = getenv("LFGPU_SC_SPLIT") ? atoi(getenv("LFGPU_SC_SPLIT")) : -1;
  a.split_waves = split_env >= 0 ? (u32)split_env : 1u;
  static const int wave_tail_env = getenv("LFGPU_SC_WAVE_TAIL") ? atoi(getenv("LFGPU_SC_WAVE_TAIL")) : 1;
  a.wave_tail = (u32)wave_tail_env;
  static const bool off_cache_env = !(getenv("LFGPU_SC_OFFCACHE") && atoi(getenv("LFGPU_SC_OFFCACHE")) == 0);
  if (oc && off_cache_env && !d_nh && 2 * logw <= 64) {
    // the offsets of every later round-hand follow from how the grid shrinks, i.e. from all three array sizes
    const u32 key[7] = {(u32)rh0, G, (u32)nh, per_wg, (u32)nW0, (u32)nW1, (u32)logw};
    if (!oc->d && hipMalloc((void**)&oc->d, 64 * (LF_SC_GRID_WGS + 1) * sizeof(u32)) != hipSuccess) {
      oc->d = nullptr;
      (void)hipGetLastError();
    }
    if (oc->d) {
      const bool same = memcmp(oc->key, key, sizeof(key)) == 0;
      if (oc->state == 2 && same) {
        a.off_mode = 2;
      } else {  // first proof through this layer (or other knobs): record
        memcpy(oc->key, key, sizeof(key));
        oc->state = 1;
        a.off_mode = 1;
      }
      a.off_cache = oc->d;
    }
  }
  if (hipLaunchKernel(fn, dim3(G), dim3(SM_THREADS), args, tail_ok ? SC_TAIL_LDS_BYTES : 0, c->stream) != hipSuccess) {
    lf_cu_release(c, (int)G);
    return lf_fail(c, LFGPU_ERR_HIP, "sc_grid_begin: launch failed: %s", hipGetErrorString(hipGetLastError()));
  }
  return LFGPU_OK;
}

// waits for the next post of the resident kernel; r != nullptr answers the PREVIOUS post with its challenge first
int lf_sc_layer_next(lfgpu_ctx* c, const u64* r, u64 out[8]) {
  if (r) {
    volatile u64* cmd = c->poll_h + 64;
    const u64 seq = c->poll_next - 1, tag = (seq & 0xffffffffull) << 32;
    // the grid kernel's form: four tagged words, valid as soon as all four carry the tag (any order, no flag)
    __atomic_store_n((u64*)&cmd[4], tag | (r[0] & 0xffffffffull), __ATOMIC_RELAXED);
    __atomic_store_n((u64*)&cmd[5], tag | (r[0] >> 32), __ATOMIC_RELAXED);
    __atomic_store_n((u64*)&cmd[6], tag | (r[1] & 0xffffffffull), __ATOMIC_RELAXED);
    __atomic_store_n((u64*)&cmd[7], tag | (r[1] >> 32), __ATOMIC_RELAXED);
    // the single-workgroup kernels' form: payload, then the sequence number
    cmd[0] = r[0];
    cmd[1] = r[1];
    __atomic_store_n((u64*)&cmd[2], seq, __ATOMIC_RELEASE);
  }
  LF_TRY(sc_wait_post(c, c->poll_next));
  ++c->poll_next;
  for (int i = 0; i < 8; ++i) out[i] = c->poll_h[i];
#ifdef LF_SC_PROF
  {  // per round-hand arrival times against the HQUAD size (LFGPU_SC_TRACE=1)
    static std::vector<std::pair<double, u64>> tr;
    static const bool on = getenv("LFGPU_SC_TRACE") != nullptr;
    if (on) {
      tr.emplace_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(), out[4]);
      if (c->poll_next == c->poll_seq + 1) {
        fprintf(stderr, "sc_trace:");
        for (size_t i = 1; i < tr.size(); ++i) fprintf(stderr, " %llu:%.1f", (unsigned long long)tr[i - 1].second, tr[i].first - tr[i - 1].first);
        fprintf(stderr, "\n");
        tr.clear();
      }
    }
  }
  if (c->poll_next == c->poll_seq + 1) {  // final post of a layer: fold the phase clocks
    static u64 tot[9];
    static int layers = 0;
    for (int k = 0; k < 9; ++k) tot[k] += c->poll_h[16 + k];
    if (++layers % 13 == 0) {
      fprintf(stderr, "sc_grid phases (us, %d layers, %llu round-hands counted): bar1 %.0f shrink %.0f sums+post %.0f counts %.0f bar2 %.0f layout %.0f wait %.0f bind+scatter %.0f\n", layers,
              (unsigned long long)tot[8], tot[0] * 1e3 / c->wall_khz, tot[1] * 1e3 / c->wall_khz, tot[2] * 1e3 / c->wall_khz, tot[3] * 1e3 / c->wall_khz,
              tot[4] * 1e3 / c->wall_khz, tot[5] * 1e3 / c->wall_khz, tot[6] * 1e3 / c->wall_khz, tot[7] * 1e3 / c->wall_khz);
      for (int k = 0; k < 9; ++k) tot[k] = 0;
    }
  }
#endif
  return LFGPU_OK;
}

static int sc_partials(lfgpu_ctx* c, int field, size_t n, const void* d_QW, const void* d_W, uint64_t a0[2], uint64_t a2[2], bool clean);
extern "C" int lfgpu_sumcheck_partials(lfgpu_ctx* c, int field, size_t n, const void* d_QW, const void* d_W,
                                       uint64_t a0[2], uint64_t a2[2]) {
  return sc_partials(c, field, n, d_QW, d_W, a0, a2, false);
}
// the sums of one evaluation; `d_QW` is left all zero (the accumulators are spent: see sumcheck_partials_kernel)
int lf_sumcheck_partials_clean(lfgpu_ctx* c, int field, size_t n, void* d_QW, const void* d_W, uint64_t a0[2], uint64_t a2[2]) {
  return sc_partials(c, field, n, d_QW, d_W, a0, a2, true);
}
static int sc_partials(lfgpu_ctx* c, int field, size_t n, const void* d_QW, const void* d_W, uint64_t a0[2], uint64_t a2[2], bool clean) {
  if (!c || !a0 || !a2 || (n && (!d_QW || !d_W))) return lf_fail(c, LFGPU_ERR_ARG, "sumcheck_partials: null argument");
  LF_HIP(c, hipSetDevice(c->device));
  size_t nodd = n / 2;
  u32 nb = (u32)((nodd + SC_THREADS - 1) / SC_THREADS);
  if (nb == 0) nb = 1;
  if (nb > SC_MAX_BLOCKS) nb = SC_MAX_BLOCKS;
  elt_t* partial = (elt_t*)c->mailbox_d + 8;  // needs 2*nb*16 bytes: use scratch2 instead when large
  void* sc = nullptr;
  LF_TRY(lf_scratch2(c, (size_t)2 * SC_MAX_BLOCKS * 16 + 64, &sc));
  partial = (elt_t*)sc;
  elt_t* out = (elt_t*)c->mailbox_d;
  DISPATCH_FIELD(field, sumcheck_partials_kernel, dim3(nb), dim3(SC_THREADS), n, (const elt_t*)d_QW,
                 (const elt_t*)d_W, partial, clean ? (elt_t*)const_cast<void*>(d_QW) : (elt_t*)nullptr);
  if (lf_sc_resident_ok(c)) {  // a running kernel's stores to the pinned words reach the host on this system (self-test, once)
    const u64 seq = ++c->poll_seq;
    DISPATCH_FIELD(field, sumcheck_final_kernel, dim3(1), dim3(SC_THREADS), nb, (const elt_t*)partial, out, c->poll_h, seq);
    LF_HIP(c, hipGetLastError());
    LF_TRY(sc_wait_post(c, seq));
    a0[0] = c->poll_h[0]; a0[1] = c->poll_h[1]; a2[0] = c->poll_h[2]; a2[1] = c->poll_h[3];
    return LFGPU_OK;
  }
  DISPATCH_FIELD(field, sumcheck_final_kernel, dim3(1), dim3(SC_THREADS), nb, (const elt_t*)partial, out, (volatile u64*)nullptr, (u64)0);
  LF_HIP(c, hipGetLastError());
  LF_HIP(c, hipMemcpyAsync(c->mailbox_h, out, 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(a0, c->mailbox_h, 16);
  memcpy(a2, (const uint8_t*)c->mailbox_h + 16, 16);
  return LFGPU_OK;
}

int lf_qw_scatter_gf_into(lfgpu_ctx* c, size_t n, const void* d_hc, const void* d_vc, int hand, const void* d_Wother, void* d_QW);
extern "C" int lfgpu_qw_scatter(lfgpu_ctx* c, int field, size_t n, const void* d_hc, const void* d_vc, int hand,
                                const void* d_Wother, size_t nqw, void* d_QW) {
  if (!c || !d_QW || (n && (!d_hc || !d_vc || !d_Wother))) return lf_fail(c, LFGPU_ERR_ARG, "qw_scatter: null argument");
  LF_HIP(c, hipSetDevice(c->device));
  if (field == LFGPU_FIELD_FP128) {
    // integer limb accumulators + one reduction (see qw_scatter_fp_kernel)
    if (n >> 32) return lf_fail(c, LFGPU_ERR_ARG, "qw_scatter: more than 2^32 terms");
    void* sc = nullptr;
    LF_TRY(lf_scratch2(c, nqw * 32 + 5 * 16 + 64, &sc));
    u64* acc = (u64*)sc;
    LF_HIP(c, hipMemsetAsync(acc, 0, nqw * 32, c->stream));
    if (n) {
      u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
      hipLaunchKernelGGL(qw_scatter_fp_kernel, dim3(nb), dim3(SC_THREADS), 0, c->stream, n, (const uint2*)d_hc,
                         (const elt_t*)d_vc, hand ? 1 : 0, (const elt_t*)d_Wother, acc);
    }
    u32 nb2 = (u32)((nqw + SC_THREADS - 1) / SC_THREADS);
    hipLaunchKernelGGL(fp_limb_normalize_kernel, dim3(nb2), dim3(SC_THREADS), 0, c->stream, nqw, (const u64*)acc, (elt_t*)d_QW);
    LF_HIP(c, hipGetLastError());
    return LFGPU_OK;
  }
  LF_HIP(c, hipMemsetAsync(d_QW, 0, nqw * 16, c->stream));
  return lf_qw_scatter_gf_into(c, n, d_hc, d_vc, hand, d_Wother, d_QW);
}
// GF(2^128): QW ^= the scatter; the caller guarantees cleared targets (a memset once, lf_sumcheck_partials_clean thereafter)
int lf_qw_scatter_gf_into(lfgpu_ctx* c, size_t n, const void* d_hc, const void* d_vc, int hand, const void* d_Wother, void* d_QW) {
  if (n) {
    u32 nb = (u32)((n + SCAT_THREADS - 1) / SCAT_THREADS);
    hipLaunchKernelGGL(qw_scatter_gf_kernel, dim3(nb), dim3(SCAT_THREADS), 0, c->stream, n, (const uint2*)d_hc,
                       (const elt_t*)d_vc, hand ? 1 : 0, (const elt_t*)d_Wother, (u64*)d_QW);
    LF_HIP(c, hipGetLastError());
  }
  return LFGPU_OK;
}

extern "C" int lfgpu_dense_bind(lfgpu_ctx* c, int field, size_t n0, const uint64_t r[2], const void* d_in,
                                void* d_out) {
  if (!c || !r || (n0 && (!d_in || !d_out))) return lf_fail(c, LFGPU_ERR_ARG, "dense_bind: null argument");
  if (n0 == 0) return LFGPU_OK;
  LF_HIP(c, hipSetDevice(c->device));
  size_t nout = (n0 + 1) / 2;
  elt_t rr{r[0], r[1]};
  void* dst = d_out;
  if (d_out == d_in) LF_TRY(lf_scratch2(c, nout * 16, &dst));  // parallel in-place would race
  u32 nb = (u32)((nout + SC_THREADS - 1) / SC_THREADS);
  DISPATCH_FIELD(field, dense_bind_kernel, dim3(nb), dim3(SC_THREADS), n0, rr, (const elt_t*)d_in, (elt_t*)dst);
  LF_HIP(c, hipGetLastError());
  if (dst != d_out) LF_HIP(c, hipMemcpyAsync(d_out, dst, nout * 16, hipMemcpyDeviceToDevice, c->stream));
  return LFGPU_OK;
}

extern "C" int lfgpu_hquad_bind_h(lfgpu_ctx* c, int field, size_t n, const void* d_hc, const void* d_vc,
                                  const uint64_t r[2], int hand, void* d_hc_out, void* d_vc_out, size_t* n_out) {
  if (!c || !r || !n_out || (n && (!d_hc || !d_vc || !d_hc_out || !d_vc_out)))
    return lf_fail(c, LFGPU_ERR_ARG, "hquad_bind_h: null argument");
  if (d_hc_out == d_hc || d_vc_out == d_vc) return lf_fail(c, LFGPU_ERR_ARG, "hquad_bind_h: outputs must not alias inputs");
  *n_out = 0;
  if (n == 0) return LFGPU_OK;
  LF_HIP(c, hipSetDevice(c->device));
  u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
  void* sc = nullptr;
  LF_TRY(lf_scratch2(c, (size_t)nb * 4 + 64, &sc));
  u32* counts = (u32*)sc;
  u32* total = (u32*)((uint8_t*)c->mailbox_d + 64);
  elt_t rr{r[0], r[1]};
  hand = hand ? 1 : 0;
  hipLaunchKernelGGL(hquad_count_kernel, dim3(nb), dim3(SC_THREADS), 0, c->stream, n, (const uint2*)d_hc, hand, counts);
  hipLaunchKernelGGL(hquad_scan_kernel, dim3(1), dim3(1024), 0, c->stream, nb, counts, total);
  DISPATCH_FIELD(field, hquad_emit_kernel, dim3(nb), dim3(SC_THREADS), n, (const uint2*)d_hc, (const elt_t*)d_vc, rr,
                 hand, (const u32*)counts, (uint2*)d_hc_out, (elt_t*)d_vc_out);
  LF_HIP(c, hipGetLastError());
  LF_HIP(c, hipMemcpyAsync(c->mailbox_h, total, 4, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  *n_out = *(const u32*)c->mailbox_h;
  return LFGPU_OK;
}

// HQuad::bind_h for a caller that keeps the merge structure: which entries merge depends on the corner indices only
// (hquad.h:90-123), i.e. on the circuit and the position in the layer, never on the challenges.
//   d_off_cached == nullptr: count + scan + emit as lfgpu_hquad_bind_h; *d_off_keep (if asked for) receives a
//                            persistent copy of the per-block output offsets, *n_out the new size (one synchronisation);
//   d_off_cached != nullptr: emit only with those offsets -- no count, no scan, nothing read back.
int lf_hquad_bind_h_cached(lfgpu_ctx* c, int field, size_t n, const void* d_hc, const void* d_vc, const uint64_t r[2], int hand,
                           void* d_hc_out, void* d_vc_out, const u32* d_off_cached, u32** d_off_keep, size_t* n_out) {
  if (n == 0) return lf_fail(c, LFGPU_ERR_ARG, "hquad_bind_h_cached: empty");
  const u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
  const elt_t rr{r[0], r[1]};
  hand = hand ? 1 : 0;
  if (d_off_cached) {
    DISPATCH_FIELD(field, hquad_emit_kernel, dim3(nb), dim3(SC_THREADS), n, (const uint2*)d_hc, (const elt_t*)d_vc, rr,
                   hand, d_off_cached, (uint2*)d_hc_out, (elt_t*)d_vc_out);
    LF_HIP(c, hipGetLastError());
    return LFGPU_OK;
  }
  LF_TRY(lfgpu_hquad_bind_h(c, field, n, d_hc, d_vc, r, hand, d_hc_out, d_vc_out, n_out));
  if (d_off_keep) {  // the offsets still stand in the scratch the call above used (nothing ran since)
    void* sc = nullptr;
    LF_TRY(lf_scratch2(c, (size_t)nb * 4 + 64, &sc));
    u32* keep = nullptr;
    if (hipMalloc((void**)&keep, (size_t)nb * 4) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "hquad_bind_h_cached: offsets");
    LF_HIP(c, hipMemcpyAsync(keep, sc, (size_t)nb * 4, hipMemcpyDeviceToDevice, c->stream));
    *d_off_keep = keep;
  }
  return LFGPU_OK;
}

// Dense::bind (out of place, d_out != d_in) + HQuad::bind_h with recorded offsets, one launch (bind_both_kernel)
int lf_bind_both_cached(lfgpu_ctx* c, int field, size_t n0, const uint64_t r[2], const void* d_in, void* d_out, size_t n, const void* d_hc,
                        const void* d_vc, int hand, void* d_hc_out, void* d_vc_out, const u32* d_off_cached) {
  if (!c || !r || n0 == 0 || n == 0 || !d_in || !d_out || d_in == d_out || !d_hc || !d_vc || !d_hc_out || !d_vc_out || !d_off_cached)
    return lf_fail(c, LFGPU_ERR_ARG, "bind_both_cached: bad argument");
  const u32 nbD = (u32)(((n0 + 1) / 2 + SC_THREADS - 1) / SC_THREADS), nbH = (u32)((n + SC_THREADS - 1) / SC_THREADS);
  const elt_t rr{r[0], r[1]};
  DISPATCH_FIELD(field, bind_both_kernel, dim3(nbD + nbH), dim3(SC_THREADS), nbD, n0, rr, (const elt_t*)d_in, (elt_t*)d_out, n, (const uint2*)d_hc,
                 (const elt_t*)d_vc, hand ? 1 : 0, d_off_cached, (uint2*)d_hc_out, (elt_t*)d_vc_out);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

extern "C" int lfgpu_rows_axpy(lfgpu_ctx* c, int field, size_t nrows, size_t n, void* d_y, const uint64_t* h_u,
                               const void* d_T, size_t ld) {
  if (!c || (n && !d_y) || (nrows && (!h_u || !d_T))) return lf_fail(c, LFGPU_ERR_ARG, "rows_axpy: null argument");
  if (n == 0 || nrows == 0) return LFGPU_OK;
  LF_HIP(c, hipSetDevice(c->device));
  void* du = nullptr;
  LF_TRY(lf_scratch2(c, nrows * 16, &du));
  LF_HIP(c, hipMemcpyAsync(du, h_u, nrows * 16, hipMemcpyHostToDevice, c->stream));
  u32 nb = (u32)((n + 63) / 64);
  DISPATCH_FIELD(field, rows_axpy_kernel, dim3(nb), dim3(1024), (u32)nrows, n, (elt_t*)d_y, (const elt_t*)du,
                 (const elt_t*)d_T, ld);
  LF_HIP(c, hipGetLastError());
  LF_HIP(c, hipStreamSynchronize(c->stream));  // h_u / du are reused by the caller
  return LFGPU_OK;
}

extern "C" int lfgpu_gather_columns(lfgpu_ctx* c, size_t nrow, size_t ld, size_t col0, const void* d_T,
                                    const size_t* h_idx, size_t nreq, void* d_req) {
  if (!c || !d_T || !h_idx || !d_req) return lf_fail(c, LFGPU_ERR_ARG, "gather_columns: null argument");
  if (nrow == 0 || nreq == 0) return LFGPU_OK;
  for (size_t j = 0; j < nreq; ++j)
    if (col0 + h_idx[j] >= ld) return lf_fail(c, LFGPU_ERR_ARG, "gather_columns: index out of range");
  LF_HIP(c, hipSetDevice(c->device));
  void* di = nullptr;
  LF_TRY(lf_scratch2(c, nreq * 8, &di));
  LF_HIP(c, hipMemcpyAsync(di, h_idx, nreq * 8, hipMemcpyHostToDevice, c->stream));
  u32 tot = (u32)(nrow * nreq);
  hipLaunchKernelGGL(gather_columns_kernel, dim3((tot + 255) / 256), dim3(256), 0, c->stream, (u32)nrow, ld, col0,
                     (const elt_t*)d_T, (const u64*)di, (u32)nreq, (elt_t*)d_req);
  LF_HIP(c, hipGetLastError());
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}

extern "C" int lfgpu_field_binop(lfgpu_ctx* c, int field, int op, size_t n, const void* d_a, const void* d_b,
                                 void* d_out) {
  if (!c || (n && (!d_a || !d_b || !d_out)) || op < 0 || op > 2) return lf_fail(c, LFGPU_ERR_ARG, "field_binop: bad argument");
  if (n == 0) return LFGPU_OK;
  LF_HIP(c, hipSetDevice(c->device));
  if (field == LFGPU_FIELD_P256) return lf_p256_binop(c, op, n, d_a, d_b, d_out);
  u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
  DISPATCH_FIELD(field, field_binop_kernel, dim3(nb), dim3(SC_THREADS), op, n, (const elt_t*)d_a, (const elt_t*)d_b,
                 (elt_t*)d_out);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}
