// quad.hip -- K10 (Quad::bind_g) and K11 (ProverLayers::eval_quad) on an uploaded layer.
//
// Reference: Quad::bind_g + prep_v (lib/sumcheck/quad.h:152-185,213-220), Eqs::raw_eq2 /
// fill_recursive (lib/arrays/eqs.h:46-80), ProverLayers::eval_quad
// (lib/sumcheck/prover_layers.h:278-305), corner order of EQuad::canonicalize
// (lib/sumcheck/equad.h:79-106).
//
// A layer is uploaded ONCE per circuit in expanded form (the delta decode of quad.h:100-130
// is a host prefix sum done by the caller's adapter) and kept in HBM in two orders:
//   * canonical order (Morton(h0,h1), then g): equal hand pairs are adjacent, which is what
//     makes bind_g a run-length reduction (K10);
//   * grouped by output gate g (CSR): eval_quad becomes a per-gate gather/sum without atomics,
//     so Fp128 needs no 128-bit atomic and GF2_128 results do not depend on arrival order (K11).
#include <algorithm>
#include <numeric>

#include "ctx.h"
#include "quad.h"
#include "runfold.h"
#include "zkint.h"
#include <chrono>

#define QD_THREADS 256

// ---- K11: V[g] = sum_{terms of g} kvec[vi] * W[h1] * W[h0]; assert-zero terms must vanish
template <int F>
__global__ __launch_bounds__(QD_THREADS) void eval_quad_kernel(u32 nv, const u32* __restrict__ goff,
                                                               const corner4* __restrict__ terms,
                                                               const elt_t* __restrict__ kvec, const elt_t* __restrict__ W,
                                                               elt_t* __restrict__ V, int* __restrict__ fail) {
  u32 g = blockIdx.x * QD_THREADS + threadIdx.x;
  if (g >= nv) return;
  elt_t acc = elt_zero();
  bool bad = false;
  for (u32 t = goff[g]; t < goff[g + 1]; ++t) {
    corner4 cr = *reinterpret_cast<const corner4*>(&terms[t]);
    elt_t v = ld16(&kvec[cr.vi]);
    elt_t p = Fld<F>::mul(ld16(&W[cr.h1]), ld16(&W[cr.h0]));
    if ((v.lo | v.hi) == 0) {
      bad |= (p.lo | p.hi) != 0;
    } else {
      acc = Fld<F>::add(acc, Fld<F>::mul(v, p));
    }
  }
  st16(&V[g], acc);
  if (bad) atomicOr(fail, 1);
}

// ---- K10 step 1: eq[i] = EQ(G0,i) + alpha*EQ(G1,i), EQ(G,i) = prod_l (bit_l(i) ? G[l] : 1 - G[l])
// The binding points travel as a KERNEL ARGUMENT (4 * logn + 1 <= 161 elements = 2.6 KB of the 4 KB kernarg segment; dynamic
// indexing compiles to loads from that segment, no scratch): no staging copy, i.e. one dispatch less per EQ table -- a layer of a
// proof is a chain of dependent dispatches and every one costs 3 - 4 us alone and ten times that with 16 provers.
struct EqPoints {
  elt_t g[4 * 40 + 1];  // G0 | G1 | 1-G0 | 1-G1
};
// zero[0 .. nzero) (64-bit words) is cleared on the side by the kernels that span the table: the accumulators of the next kernel
// on the stream (Quad::bind_g's emit), instead of a hipMemsetAsync of their own
__device__ __forceinline__ void eq_side_clear(u64* __restrict__ zero, u32 nzero) {
  if (!zero) return;
  const u32 T = gridDim.x * QD_THREADS;
  for (u32 k = blockIdx.x * QD_THREADS + threadIdx.x; k < nzero; k += T) zero[k] = 0;
}
template <int F>
__global__ __launch_bounds__(QD_THREADS) void raw_eq2_kernel(u32 logn, u32 n, EqPoints gp, elt_t alpha, elt_t one, elt_t* __restrict__ eq,
                                                             u64* __restrict__ zero, u32 nzero) {
  const elt_t* G = gp.g;
  eq_side_clear(zero, nzero);
  u32 i = blockIdx.x * QD_THREADS + threadIdx.x;
  if (i >= n) return;
  elt_t e0 = one, e1 = alpha;
  for (u32 l = 0; l < logn; ++l) {
    u32 bit = (i >> l) & 1;
    e0 = Fld<F>::mul(e0, ld16(&G[(bit ? 0 : 2 * logn) + l]));
    e1 = Fld<F>::mul(e1, ld16(&G[(bit ? logn : 3 * logn) + l]));
  }
  st16(&eq[i], Fld<F>::add(e0, e1));
}

// The same vector with 2 products per entry instead of 2*logn: EQ(G, i) factors over the low and the high half of
// the index bits, EQ(G, i) = LO[i mod 2^lb] * HI[i >> lb].  eq_tables_kernel builds the four factor tables
// (LO0 | HI0 | LO1 | HI1, alpha folded into HI1; at most 2^ceil(logn/2) entries each, a product of <= 20 factors per
// entry), raw_eq2_split_kernel combines them.  Exact field arithmetic: the association does not matter.
template <int F>
__global__ __launch_bounds__(QD_THREADS) void eq_tables_kernel(u32 logn, u32 lb, EqPoints gp, elt_t alpha, elt_t one, elt_t* __restrict__ tab) {
  const elt_t* G = gp.g;
  // four lanes per entry, each multiplies every fourth factor, two shuffle steps combine them: the kernel is a chain of
  // dependent products (its tables are tiny), so the depth -- ceil(bits / 4) + 2 instead of bits -- is its run time
  const u32 hb = logn - lb, nlo = 1u << lb, nhi = 1u << hb;
  const u32 tid = blockIdx.x * QD_THREADS + threadIdx.x, t = tid >> 2, part = tid & 3;
  const bool live = t < 2 * (nlo + nhi);  // whole quads are live or idle together (the entry count is even)
  const u32 tt = live ? t : 0;
  const u32 which = tt < nlo ? 0 : tt < nlo + nhi ? 1 : tt < 2 * nlo + nhi ? 2 : 3;  // LO0, HI0, LO1, HI1
  const u32 j = which == 0 ? tt : which == 1 ? tt - nlo : which == 2 ? tt - nlo - nhi : tt - 2 * nlo - nhi;
  const u32 bits = (which & 1) ? hb : lb, shift = (which & 1) ? lb : 0, g = which >> 1;  // g: 0 -> G0, 1 -> G1
  elt_t e = (which == 3 && part == 0) ? alpha : one;
  for (u32 l = part; l < bits; l += 4) {
    const u32 bit = (j >> l) & 1;
    e = Fld<F>::mul(e, ld16(&G[(bit ? g * logn : (2 + g) * logn) + shift + l]));
  }
#pragma unroll
  for (int x = 1; x <= 2; x <<= 1) {
    elt_t o;
    o.lo = __shfl_xor(e.lo, x, 64);
    o.hi = __shfl_xor(e.hi, x, 64);
    e = Fld<F>::mul(e, o);
  }
  if (live && part == 0) st16(&tab[t], e);
}
template <int F>
__global__ __launch_bounds__(QD_THREADS) void raw_eq2_split_kernel(u32 logn, u32 lb, u32 n, const elt_t* __restrict__ tab, elt_t* __restrict__ eq,
                                                                   u64* __restrict__ zero, u32 nzero) {
  eq_side_clear(zero, nzero);
  const u32 i = blockIdx.x * QD_THREADS + threadIdx.x;
  if (i >= n) return;
  const u32 nlo = 1u << lb, nhi = 1u << (logn - lb);
  const u32 lo = i & (nlo - 1), hi = i >> lb;
  const elt_t e0 = Fld<F>::mul(ld16(&tab[lo]), ld16(&tab[nlo + hi]));
  const elt_t e1 = Fld<F>::mul(ld16(&tab[nlo + nhi + lo]), ld16(&tab[2 * nlo + nhi + hi]));
  st16(&eq[i], Fld<F>::add(e0, e1));
}

// Both steps in ONE launch for tables of up to 2^16 entries (one dispatch less per layer): the low factor covers
// 4 index bits, so a 256-thread block needs 16 + 16 low and 16 + 16 high factor entries -- 64 entries, four lanes each as above,
// kept in LDS -- and then combines them into its 256 entries of the vector.  The blocks repeat the 32 low entries; the work per
// entry stays within 3.5x of the split kernel's, which is why larger tables keep the two launches.  Same products in another
// association: exact field arithmetic, same bytes.
template <int F>
__global__ __launch_bounds__(QD_THREADS) void raw_eq2_fused_kernel(u32 logn, u32 n, EqPoints gp, elt_t alpha, elt_t one, elt_t* __restrict__ eq,
                                                                   u64* __restrict__ zero, u32 nzero) {
  static_assert(QD_THREADS == 256, "raw_eq2_fused_kernel: 64 factor entries x 4 lanes");
  __shared__ elt_t tab[64];  // LO0[16] | LO1[16] | HI0[16] | HI1[16] (alpha folded into HI1)
  const elt_t* G = gp.g;
  eq_side_clear(zero, nzero);
  constexpr u32 lb = 4;
  const u32 t = threadIdx.x, e = t >> 2, part = t & 3;
  const u32 which = e >> 4, j = e & 15, g = which & 1;
  const bool high = which >= 2;
  const u32 bits = high ? logn - lb : lb, shift = high ? lb : 0;
  const u32 idx = high ? blockIdx.x * 16 + j : j;  // (index bits beyond `bits` are never looked at)
  elt_t v = (which == 3 && part == 0) ? alpha : one;
  for (u32 l = part; l < bits; l += 4) {
    const u32 bit = (idx >> l) & 1;
    v = Fld<F>::mul(v, ld16(&G[(bit ? g * logn : (2 + g) * logn) + shift + l]));
  }
#pragma unroll
  for (int x = 1; x <= 2; x <<= 1) {
    elt_t o;
    o.lo = __shfl_xor(v.lo, x, 64);
    o.hi = __shfl_xor(v.hi, x, 64);
    v = Fld<F>::mul(v, o);
  }
  if (part == 0) tab[e] = v;
  __syncthreads();
  const u32 i = blockIdx.x * QD_THREADS + t;
  if (i >= n) return;
  const u32 lo = t & 15, h = t >> 4;
  const elt_t e0 = Fld<F>::mul(tab[lo], tab[32 + h]);
  const elt_t e1 = Fld<F>::mul(tab[16 + lo], tab[48 + h]);
  st16(&eq[i], Fld<F>::add(e0, e1));
}

// ---- K10 step 2: run heads (first term of each distinct hand pair)
__device__ __forceinline__ bool is_head(const corner4* t, size_t i) {
  if (i == 0) return true;
  return t[i].h0 != t[i - 1].h0 || t[i].h1 != t[i - 1].h1;
}
__global__ __launch_bounds__(BG_THREADS) void bindg_count_kernel(size_t n, const corner4* __restrict__ t,
                                                                 u32* __restrict__ block_counts) {
  __shared__ u32 cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  size_t i = (size_t)blockIdx.x * BG_THREADS + threadIdx.x;
  bool head = i < n && is_head(t, i);
  u64 mask = __ballot(head);
  if ((threadIdx.x & 63) == 0) atomicAdd(&cnt, (u32)__popcll(mask));
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = cnt;
}
__global__ __launch_bounds__(1024) void bindg_scan_kernel(u32 nblocks, u32* __restrict__ block_counts, u32* __restrict__ total) {
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
      u32 x = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += x;
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
// zero2[0 .. nzero2) (64-bit words, a few hundred at most) is cleared by the first block of the emit kernel: the synchronisation
// words of the resident grid that takes the layer over later (sumcheck.hip, ScGridSync) -- one hipMemsetAsync less per layer
// zero3[0 .. nzero3): a large region cleared by the whole grid -- the QW accumulators of the layer's first per-launch round-hand
__device__ __forceinline__ void emit_side_clear(u64* __restrict__ zero2, u32 nzero2, u64* __restrict__ zero3, u32 nzero3) {
  if (blockIdx.x == 0 && zero2)
    for (u32 k = threadIdx.x; k < nzero2; k += BG_THREADS) zero2[k] = 0;
  if (zero3) {
    const u32 T = gridDim.x * BG_THREADS;
    for (u32 k = blockIdx.x * BG_THREADS + threadIdx.x; k < nzero3; k += T) zero3[k] = 0;
  }
}
// ---- K10 step 3: every term computes v' = (v == 0 ? beta : v) * eq[g] (prep_v) and the run sums are formed in parallel.
// GF2_128: a hand pair shared by very many gates (constant wires: the
// 32-block flatsha256 layers hold runs of > 10^5 terms) would otherwise be summed by ONE lane.  Every
// term computes its own product, a wave folds equal-run neighbours with shuffles (runs are contiguous
// in canonical order), and the last lane of each run fragment issues one pair of 64-bit atomic XORs --
// exact and order-independent because addition in GF(2^128) is XOR.  d_vc_out must be zeroed first.
__global__ __launch_bounds__(BG_THREADS) void bindg_emit_gf_kernel(size_t n, const corner4* __restrict__ t,
                                                                   const elt_t* __restrict__ kvec,
                                                                   const elt_t* __restrict__ eq, elt_t beta,
                                                                   const u32* __restrict__ block_off,
                                                                   uint2* __restrict__ hc_out, u64* __restrict__ vc_out,
                                                                   u64* __restrict__ zero2, u32 nzero2, u64* __restrict__ zero3, u32 nzero3) {
  __shared__ u32 wave_off[BG_THREADS / 64];
  emit_side_clear(zero2, nzero2, zero3, nzero3);
  const size_t i = (size_t)blockIdx.x * BG_THREADS + threadIdx.x;
  const bool valid = i < n;
  const bool head = valid && is_head(t, i);
  const u64 mask = __ballot(head);
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_off[wave] = (u32)__popcll(mask);
  __syncthreads();
  u32 ri = block_off[blockIdx.x];
  for (u32 w = 0; w < wave; ++w) ri += wave_off[w];
  ri += (u32)__popcll(mask & ((2ull << lane) - 1));  // inclusive count of heads up to this lane
  ri -= 1;                                            // run index (a run continuing from an earlier block/wave keeps its index)
  elt_t pv = elt_zero();
  corner4 c0{0, 0, 0, 0};
  if (valid) {
    c0 = t[i];
    elt_t v = ld16(&kvec[c0.vi]);
    if ((v.lo | v.hi) == 0) v = beta;
    pv = gf_mul(v, ld16(&eq[c0.g]));
  }
  if (head) hc_out[ri] = make_uint2(c0.h0, c0.h1);
  gf_run_fold_commit<BG_THREADS>(valid ? ri : 0xffffffffu, pv, vc_out);  // one atomic pair per run and block
}

// Fp128 variant: same per-term parallelism; the run sums are accumulated as plain integers in four
// 64-bit limb accumulators per run (residues add as integers; < 2^32 terms per run) and reduced once.
__global__ __launch_bounds__(BG_THREADS) void bindg_emit_fp_kernel(size_t n, const corner4* __restrict__ t,
                                                                   const elt_t* __restrict__ kvec,
                                                                   const elt_t* __restrict__ eq, elt_t beta,
                                                                   const u32* __restrict__ block_off,
                                                                   uint2* __restrict__ hc_out, u64* __restrict__ acc,
                                                                   u64* __restrict__ zero2, u32 nzero2, u64* __restrict__ zero3, u32 nzero3) {
  __shared__ u32 wave_off[BG_THREADS / 64];
  emit_side_clear(zero2, nzero2, zero3, nzero3);
  const size_t i = (size_t)blockIdx.x * BG_THREADS + threadIdx.x;
  const bool valid = i < n;
  const bool head = valid && is_head(t, i);
  const u64 mask = __ballot(head);
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_off[wave] = (u32)__popcll(mask);
  __syncthreads();
  if (!valid) return;
  u32 ri = block_off[blockIdx.x];
  for (u32 w = 0; w < wave; ++w) ri += wave_off[w];
  ri += (u32)__popcll(mask & ((2ull << lane) - 1)) - 1;
  const corner4 c0 = t[i];
  elt_t v = ld16(&kvec[c0.vi]);
  if ((v.lo | v.hi) == 0) v = beta;
  const elt_t pv = fp_mul(v, ld16(&eq[c0.g]));
  if (head) hc_out[ri] = make_uint2(c0.h0, c0.h1);
  u64* a = acc + 4 * (size_t)ri;
  atomicAdd(&a[0], (u64)(u32)pv.lo);
  atomicAdd(&a[1], pv.lo >> 32);
  atomicAdd(&a[2], (u64)(u32)pv.hi);
  atomicAdd(&a[3], pv.hi >> 32);
}
__global__ __launch_bounds__(QD_THREADS) void fp_limb_normalize4_kernel(const u32* __restrict__ total, const u64* __restrict__ acc,
                                                                        elt_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * QD_THREADS + threadIdx.x;
  if (i >= *total) return;
  st16(&out[i], fp_reduce_limbs(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]));
}

#define QD_DISPATCH(field, KERNEL, grid, block, ...)                                    \
  do {                                                                                  \
    if ((field) == LFGPU_FIELD_GF2_128)                                                 \
      hipLaunchKernelGGL(KERNEL<FIELD_GF2_128>, grid, block, 0, c->stream, __VA_ARGS__); \
    else                                                                                \
      hipLaunchKernelGGL(KERNEL<FIELD_FP128>, grid, block, 0, c->stream, __VA_ARGS__);   \
  } while (0)

QuadArrays::~QuadArrays() {
  for (void* x : p)
    if (x) (void)hipFree(x);
}
extern "C" int lfgpu_quad_free(lfgpu_quad* q) {
  if (!q) return LFGPU_ERR_ARG;
  if (q->arrays) {
    q->arrays.reset();  // the last handle frees them
  } else {  // an upload that failed half-way
    if (q->d_morton) (void)hipFree(q->d_morton);
    if (q->d_bygate) (void)hipFree(q->d_bygate);
    if (q->d_goff) (void)hipFree(q->d_goff);
    if (q->d_kvec) (void)hipFree(q->d_kvec);
    if (q->d_runoff) (void)hipFree(q->d_runoff);
    if (q->d_nh) (void)hipFree(q->d_nh);
  }
  for (auto& b : q->bind_shape)
    if (b.d_off) (void)hipFree(b.d_off);
  if (q->grid_off.d) (void)hipFree(q->grid_off.d);
  delete q;
  return LFGPU_OK;
}

// ---- the by-gate order on the device: histogram of g, exclusive scan, scatter.  eval_quad adds a gate's terms exactly
// (field arithmetic), so their order inside a gate does not matter and the scatter may take its slots with atomics.
__global__ __launch_bounds__(256) void quad_gate_hist_kernel(size_t n, const corner4* __restrict__ t, u32* __restrict__ cnt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) atomicAdd(&cnt[t[i].g], 1u);
}
// exclusive scan of cnt[0..m) by ONE workgroup, 16 consecutive entries per thread and step (m <= a few million: well under a
// millisecond); writes off[0..m], and a second copy the scatter consumes
__global__ __launch_bounds__(1024) void quad_gate_scan_kernel(u32 m, const u32* __restrict__ cnt, u32* __restrict__ off, u32* __restrict__ cursor) {
  __shared__ u32 sh[1024];
  __shared__ u32 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (u32 base = 0; base < m; base += 16 * 1024) {
    const u32 i0 = base + threadIdx.x * 16;
    u32 v[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v[k] = i0 + k < m ? cnt[i0 + k] : 0;
      sum += v[k];
    }
    sh[threadIdx.x] = sum;
    __syncthreads();
    for (u32 o = 1; o < 1024; o <<= 1) {
      const u32 x = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += x;
      __syncthreads();
    }
    u32 run = carry + sh[threadIdx.x] - sum;  // exclusive prefix of this thread's first entry
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (i0 + k < m) {
        off[i0 + k] = run;
        cursor[i0 + k] = run;
      }
      run += v[k];
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) off[m] = carry;
}
__global__ __launch_bounds__(256) void quad_gate_scatter_kernel(size_t n, const corner4* __restrict__ t, u32* __restrict__ cursor, corner4* __restrict__ byg) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const corner4 cr = t[i];
  byg[atomicAdd(&cursor[cr.g], 1u)] = cr;
}

int lf_quad_upload_corners(lfgpu_ctx* c, int field, size_t n, const corner4* corners, size_t hmax, size_t nk, const void* h_kvec, size_t nv, lfgpu_quad** out) {
  if (n > 0xfffffff0u || nv > 0xfffffff0u) return lf_fail(c, LFGPU_ERR_ARG, "quad_upload: too large");
  const size_t esz = field == LFGPU_FIELD_P256 ? 32 : 16;  // h_kvec: nk elements of the field's in-memory size
  LF_HIP(c, hipSetDevice(c->device));
  lfgpu_quad* q = new lfgpu_quad();
  q->c = c;
  q->field = field;
  q->n = n;
  q->nk = nk;
  q->nv = nv;
  q->hmax = hmax;
  q->d_morton = q->d_bygate = nullptr;
  q->d_goff = nullptr;
  q->d_kvec = nullptr;
  q->d_runoff = q->d_nh = nullptr;
  q->nh0 = 0;
  static const bool verbose = getenv("LFGPU_VERBOSE") != nullptr;
  auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tq0 = clk();
  u32* d_tmp = nullptr;  // gate counts, then the scatter's cursors
  bool ok = hipMalloc((void**)&q->d_morton, n * 16) == hipSuccess && hipMalloc((void**)&q->d_bygate, n * 16) == hipSuccess &&
            hipMalloc((void**)&q->d_goff, (nv + 1) * 4) == hipSuccess && hipMalloc((void**)&q->d_kvec, nk * esz) == hipSuccess &&
            hipMalloc((void**)&d_tmp, 2 * (nv + 1) * 4) == hipSuccess;
  const double tq1 = clk();
  ok = ok && hipMemcpyAsync(q->d_morton, corners, n * 16, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
       hipMemcpyAsync(q->d_kvec, h_kvec, nk * esz, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
       hipMemsetAsync(d_tmp, 0, (nv + 1) * 4, c->stream) == hipSuccess;
  const u32 nb = (u32)((n + BG_THREADS - 1) / BG_THREADS);
  u32 total = 0;
  ok = ok && hipMalloc((void**)&q->d_runoff, (size_t)nb * 4) == hipSuccess && hipMalloc((void**)&q->d_nh, 4) == hipSuccess;
  if (ok) {
    const u32 nb256 = (u32)((n + 255) / 256);
    hipLaunchKernelGGL(quad_gate_hist_kernel, dim3(nb256), dim3(256), 0, c->stream, n, (const corner4*)q->d_morton, d_tmp);
    hipLaunchKernelGGL(quad_gate_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (u32)nv, (const u32*)d_tmp, q->d_goff, d_tmp + nv + 1);
    hipLaunchKernelGGL(quad_gate_scatter_kernel, dim3(nb256), dim3(256), 0, c->stream, n, (const corner4*)q->d_morton, d_tmp + nv + 1, q->d_bygate);
    // run heads of the canonical order: counts per block -> exclusive offsets + total
    hipLaunchKernelGGL(bindg_count_kernel, dim3(nb), dim3(BG_THREADS), 0, c->stream, n, (const corner4*)q->d_morton, q->d_runoff);
    hipLaunchKernelGGL(bindg_scan_kernel, dim3(1), dim3(1024), 0, c->stream, nb, q->d_runoff, q->d_nh);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&total, q->d_nh, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
         hipStreamSynchronize(c->stream) == hipSuccess;  // also: `corners` and `h_kvec` are the caller's
  }
  const double tq2 = clk();
  if (d_tmp) (void)hipFree(d_tmp);
  if (verbose) fprintf(stderr, "lfgpu quad_upload: %zu terms, %zu gates: alloc %.2f ms, copy + kernels %.2f ms, free %.2f ms\n", n, nv, tq1 - tq0, tq2 - tq1, clk() - tq2);
  if (!ok) {
    lfgpu_quad_free(q);
    return lf_fail(c, LFGPU_ERR_NOMEM, "quad_upload: device allocation / copy / run structure failed");
  }
  q->nh0 = total;
  q->arrays = std::make_shared<QuadArrays>();
  void* const owned[6] = {q->d_morton, q->d_bygate, q->d_goff, q->d_kvec, q->d_runoff, q->d_nh};
  for (int i = 0; i < 6; ++i) q->arrays->p[i] = owned[i];
  *out = q;
  return LFGPU_OK;
}
int lf_quad_share(lfgpu_ctx* c, const lfgpu_quad* src, lfgpu_quad** out) {
  if (!c || !src || !out || !src->arrays || c->device != src->c->device) return lf_fail(c, LFGPU_ERR_ARG, "quad_share: contexts must be on the same device");
  lfgpu_quad* q = new lfgpu_quad();
  q->c = c;
  q->arrays = src->arrays;
  q->field = src->field;
  q->n = src->n; q->nk = src->nk; q->nv = src->nv; q->hmax = src->hmax;
  q->d_morton = src->d_morton; q->d_bygate = src->d_bygate; q->d_goff = src->d_goff; q->d_kvec = src->d_kvec;
  q->d_runoff = src->d_runoff; q->d_nh = src->d_nh;
  q->nh0 = src->nh0;
  *out = q;
  return LFGPU_OK;
}

extern "C" int lfgpu_quad_upload(lfgpu_ctx* c, int field, size_t n, const uint32_t* g, const uint32_t* h0,
                                 const uint32_t* h1, const uint32_t* vi, size_t nk, const void* h_kvec, size_t nv,
                                 lfgpu_quad** out) {
  if (!c || !out || n == 0 || !g || !h0 || !h1 || !vi || !h_kvec || nk == 0 || nv == 0)
    return lf_fail(c, LFGPU_ERR_ARG, "quad_upload: bad argument (Quad n > 0, quad.h:86)");
  if (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128 && field != LFGPU_FIELD_P256) return lf_fail(c, LFGPU_ERR_ARG, "quad_upload: field");
  if (n > 0xfffffff0u || nv > 0xfffffff0u) return lf_fail(c, LFGPU_ERR_ARG, "quad_upload: too large");
  std::vector<corner4> mort(n);
  size_t hmax = 0;
  for (size_t i = 0; i < n; ++i) {
    if (g[i] >= nv || vi[i] >= nk) return lf_fail(c, LFGPU_ERR_ARG, "quad_upload: corner %zu out of range", i);
    hmax = std::max<size_t>(hmax, std::max(h0[i], h1[i]));
    mort[i] = corner4{g[i], h0[i], h1[i], vi[i]};
  }
  return lf_quad_upload_corners(c, field, n, mort.data(), hmax, nk, h_kvec, nv, out);
}

// one layer, asynchronously: assert-zero failures are OR-ed into *d_fail (device); the caller clears and reads it
int lf_eval_quad_async(lfgpu_quad* q, const void* d_W, void* d_V, int* d_fail) {
  lfgpu_ctx* c = q->c;
  if (q->field == LFGPU_FIELD_P256) return lf256_eval_quad_async(q, d_W, d_V, d_fail);
  u32 nb = (u32)((q->nv + QD_THREADS - 1) / QD_THREADS);
  QD_DISPATCH(q->field, eval_quad_kernel, dim3(nb), dim3(QD_THREADS), (u32)q->nv, (const u32*)q->d_goff,
              (const corner4*)q->d_bygate, (const elt_t*)q->d_kvec, (const elt_t*)d_W, (elt_t*)d_V, d_fail);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

extern "C" int lfgpu_eval_quad(lfgpu_quad* q, size_t nw, const void* d_W, void* d_V, int* ok_out) {
  if (!q || !d_W || !d_V || !ok_out) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = q->c;
  if (nw <= q->hmax) return lf_fail(c, LFGPU_ERR_ARG, "eval_quad: nw = %zu but a corner reads wire %zu", nw, q->hmax);
  LF_HIP(c, hipSetDevice(c->device));
  int* d_fail = (int*)((uint8_t*)c->mailbox_d + 128);
  LF_HIP(c, hipMemsetAsync(d_fail, 0, 4, c->stream));
  LF_TRY(lf_eval_quad_async(q, d_W, d_V, d_fail));
  LF_HIP(c, hipMemcpyAsync(c->mailbox_h, d_fail, 4, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  *ok_out = *(const int*)c->mailbox_h ? 0 : 1;
  return LFGPU_OK;
}

// d_zero / zero_bytes (a multiple of 8, < 32 GiB): device words cleared on the side (eq_side_clear), or nullptr
static int lf_raw_eq2_clear(lfgpu_ctx* c, int field, size_t logn, size_t n, const void* h_G0, const void* h_G1, const uint64_t alpha[2], void* d_eq,
                            void* d_zero, size_t zero_bytes) {
  if (!c || !alpha || !d_eq || (logn && (!h_G0 || !h_G1)) || (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128))
    return LFGPU_ERR_ARG;
  if (logn > 40 || ((size_t)1 << logn) < n || (n >> 32)) return lf_fail(c, LFGPU_ERR_ARG, "raw_eq2: n out of range");
  if (n == 0 || (zero_bytes >> 35)) {  // nothing to launch (or an accumulator region beyond the side clear's 32-bit word count)
    if (d_zero && zero_bytes) LF_HIP(c, hipMemsetAsync(d_zero, 0, zero_bytes, c->stream));
    if (n == 0) return LFGPU_OK;
    d_zero = nullptr;
  }
  u64* const zp = (u64*)d_zero;
  const u32 nz = d_zero ? (u32)(zero_bytes / 8) : 0u;
  LF_HIP(c, hipSetDevice(c->device));
  const elt_t one = field == LFGPU_FIELD_GF2_128 ? elt_t{1, 0} : h_fp_of_scalar(1);
  // table [G0 | G1 | 1-G0 | 1-G1]: a kernel argument (EqPoints)
  EqPoints gp;
  elt_t* const Gt = gp.g;
  const elt_t* G0 = (const elt_t*)h_G0;
  const elt_t* G1 = (const elt_t*)h_G1;
  for (size_t l = 0; l < logn; ++l) {
    Gt[l] = G0[l];
    Gt[logn + l] = G1[l];
    Gt[2 * logn + l] = field == LFGPU_FIELD_GF2_128 ? gf_add(one, G0[l]) : fp_sub(one, G0[l]);
    Gt[3 * logn + l] = field == LFGPU_FIELD_GF2_128 ? gf_add(one, G1[l]) : fp_sub(one, G1[l]);
  }
  const u32 lb = (u32)(logn / 2), hb = (u32)logn - lb;
  const size_t ntab = 2 * (((size_t)1 << lb) + ((size_t)1 << hb));
  void* d_tabv = nullptr;
  LF_TRY(lf_scratch2(c, ntab * 16 + 64, &d_tabv));
  elt_t* d_tab = (elt_t*)d_tabv;
  const elt_t al{alpha[0], alpha[1]};
  static const bool fuse_off = getenv("LFGPU_EQ_FUSED") && atoi(getenv("LFGPU_EQ_FUSED")) == 0;  // A/B
  static const bool fuse_on = getenv("LFGPU_EQ_FUSED") && atoi(getenv("LFGPU_EQ_FUSED")) == 1;  // (tests: the fused kernel on a device of one's own)
  if (logn < 6) {  // tiny: the direct product per entry
    QD_DISPATCH(field, raw_eq2_kernel, dim3((u32)((n + QD_THREADS - 1) / QD_THREADS)), dim3(QD_THREADS), (u32)logn, (u32)n, gp, al, one, (elt_t*)d_eq,
                zp, nz);
  } else if (logn <= 16 && !fuse_off && (fuse_on || lf_cu_sharers(c) >= 6)) {
    // factor tables and their combination in one launch -- in throughput mode only: with 16 provers on the device the dispatch
    // saved is worth +0.4 % proofs/s, a prover alone pays one product more on its critical path (flatsha-1: +0.05 ms)
    QD_DISPATCH(field, raw_eq2_fused_kernel, dim3((u32)((n + QD_THREADS - 1) / QD_THREADS)), dim3(QD_THREADS), (u32)logn, (u32)n, gp, al, one,
                (elt_t*)d_eq, zp, nz);
  } else {
    QD_DISPATCH(field, eq_tables_kernel, dim3((u32)((4 * ntab + QD_THREADS - 1) / QD_THREADS)), dim3(QD_THREADS), (u32)logn, lb, gp, al, one, d_tab);
    QD_DISPATCH(field, raw_eq2_split_kernel, dim3((u32)((n + QD_THREADS - 1) / QD_THREADS)), dim3(QD_THREADS), (u32)logn, lb, (u32)n,
                (const elt_t*)d_tab, (elt_t*)d_eq, zp, nz);
  }
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}
// Eqs::raw_eq2 (lib/arrays/eqs.h): eq[i] = EQ(G0, i) + alpha EQ(G1, i), i < n <= 2^logn
extern "C" int lfgpu_raw_eq2(lfgpu_ctx* c, int field, size_t logn, size_t n, const void* h_G0, const void* h_G1,
                             const uint64_t alpha[2], void* d_eq) {
  return lf_raw_eq2_clear(c, field, logn, n, h_G0, h_G1, alpha, d_eq, nullptr, 0);
}

// Enqueue only: the outputs are ordered on the context's stream; the HQUAD size is a property of the circuit (nh0,
// computed at upload), so nothing is read back.
int lf_quad_bind_g(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                   const uint64_t beta[2], void* d_hc_out, void* d_vc_out, size_t* n_out, void* d_zero2, size_t zero2_bytes, void* d_zero3,
                   size_t zero3_bytes) {
  if (!q || !alpha || !beta || !d_hc_out || !d_vc_out || (logv && (!h_G0 || !h_G1))) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = q->c;
  if (q->field == LFGPU_FIELD_P256) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "quad_bind_g: Fp256Base layers are bound inside the ZK driver (zk256.hip)");
  if (logv > 40 || ((size_t)1 << logv) < q->nv) return lf_fail(c, LFGPU_ERR_ARG, "quad_bind_g: 2^logv < nv");
  LF_HIP(c, hipSetDevice(c->device));
  const int field = q->field;
  const size_t n = q->n;
  const u32 nb = (u32)((n + BG_THREADS - 1) / BG_THREADS);
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, q->nv * 16 + 256, &sc));
  elt_t* d_eq = (elt_t*)sc;
  elt_t be{beta[0], beta[1]};
  if (field == LFGPU_FIELD_GF2_128) {
    // (the emit kernel XORs into d_vc_out: cleared on the side by the EQ kernel in front of it)
    LF_TRY(lf_raw_eq2_clear(c, field, logv, q->nv, h_G0, h_G1, alpha, d_eq, d_vc_out, q->nh0 * 16));
    hipLaunchKernelGGL(bindg_emit_gf_kernel, dim3(nb), dim3(BG_THREADS), 0, c->stream, n, (const corner4*)q->d_morton,
                       (const elt_t*)q->d_kvec, (const elt_t*)d_eq, be, (const u32*)q->d_runoff, (uint2*)d_hc_out, (u64*)d_vc_out,
                       (u64*)d_zero2, (u32)(zero2_bytes / 8), (u64*)d_zero3, (u32)(zero3_bytes / 8));
  } else {
    // Fp128 has no 128-bit atomic add: integer limb accumulators + one reduction per run
    if (n >> 32) return lf_fail(c, LFGPU_ERR_ARG, "quad_bind_g: more than 2^32 terms");
    void* accv = nullptr;
    LF_TRY(lf_scratch4(c, q->nh0 * 32 + 64, &accv));  // (scratch2 holds the EQ factor tables of lf_raw_eq2_clear)
    LF_TRY(lf_raw_eq2_clear(c, field, logv, q->nv, h_G0, h_G1, alpha, d_eq, accv, q->nh0 * 32));
    hipLaunchKernelGGL(bindg_emit_fp_kernel, dim3(nb), dim3(BG_THREADS), 0, c->stream, n, (const corner4*)q->d_morton,
                       (const elt_t*)q->d_kvec, (const elt_t*)d_eq, be, (const u32*)q->d_runoff, (uint2*)d_hc_out, (u64*)accv,
                       (u64*)d_zero2, (u32)(zero2_bytes / 8), (u64*)d_zero3, (u32)(zero3_bytes / 8));
    hipLaunchKernelGGL(fp_limb_normalize4_kernel, dim3((u32)((q->nh0 + QD_THREADS - 1) / QD_THREADS)), dim3(QD_THREADS), 0, c->stream,
                       (const u32*)q->d_nh, (const u64*)accv, (elt_t*)d_vc_out);
  }
  LF_HIP(c, hipGetLastError());
  if (n_out) *n_out = q->nh0;
  return LFGPU_OK;
}
extern "C" int lfgpu_quad_bind_g(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                                 const uint64_t beta[2], void* d_hc_out, void* d_vc_out, size_t* n_out) {
  if (!n_out) return LFGPU_ERR_ARG;
  return lf_quad_bind_g(q, logv, h_G0, h_G1, alpha, beta, d_hc_out, d_vc_out, n_out, nullptr, 0, nullptr, 0);
}


// ---- Quad::bind_gh_all (lib/sumcheck/quad.h:188-210): the verifier's combined bind, no expansion:
//   sum over the corners of prep_v(v, beta) * (EQ(G0,g) + alpha EQ(G1,g)) * EQ(H0,h0) * EQ(H1,h1)
// Three EQ tables by raw_eq2_kernel, then one pass over the corners with a block-reduced sum; the device-wide fold
// of the block sums uses XOR words (GF2_128) / 32-bit-limb integer accumulators (Fp128), so it is exact and
// independent of arrival order.
#define BGH_THREADS 1024  // few, large blocks: every block ends with one atomic pair on the same accumulator words
template <int F>
__global__ __launch_bounds__(BGH_THREADS) void bind_gh_all_kernel(size_t n, const corner4* __restrict__ t, const elt_t* __restrict__ kvec,
                                                                 const elt_t* __restrict__ eqg, const elt_t* __restrict__ eqh0,
                                                                 const elt_t* __restrict__ eqh1, elt_t beta, u64* __restrict__ acc) {
  __shared__ elt_t sh[BGH_THREADS / 64];
  elt_t s = elt_zero();
  for (size_t i = (size_t)blockIdx.x * BGH_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * BGH_THREADS) {
    const corner4 cr = t[i];
    elt_t v = ld16(&kvec[cr.vi]);
    if ((v.lo | v.hi) == 0) v = beta;  // prep_v: assert-zero terms carry beta (quad.h:213-220)
    elt_t qv = Fld<F>::mul(v, ld16(&eqg[cr.g]));
    qv = Fld<F>::mul(qv, ld16(&eqh0[cr.h0]));
    s = Fld<F>::add(s, Fld<F>::mul(qv, ld16(&eqh1[cr.h1])));
  }
  for (int off = 32; off > 0; off >>= 1) {
    elt_t o;
    o.lo = __shfl_down(s.lo, off, 64);
    o.hi = __shfl_down(s.hi, off, 64);
    s = Fld<F>::add(s, o);
  }
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (u32 w = 1; w < BGH_THREADS / 64; ++w) s = Fld<F>::add(s, sh[w]);
    if (F == FIELD_GF2_128) {
      atomicXor(&acc[0], s.lo);
      atomicXor(&acc[1], s.hi);
    } else {
      atomicAdd(&acc[0], (u64)(u32)s.lo);
      atomicAdd(&acc[1], s.lo >> 32);
      atomicAdd(&acc[2], (u64)(u32)s.hi);
      atomicAdd(&acc[3], s.hi >> 32);
    }
  }
}

// enqueues the tables and the sum on the context's stream; d_acc: 4 words (zeroed here) that hold the XOR words / limb
// sums afterwards.  No synchronisation: the host operands are staged before this returns.
int lf_quad_bind_gh_all_enqueue(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                                const uint64_t beta[2], size_t logw, size_t nw, const void* h_H0, const void* h_H1, u64* d_acc) {
  if (!q || !alpha || !beta || !d_acc || (logv && (!h_G0 || !h_G1)) || (logw && (!h_H0 || !h_H1))) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = q->c;
  if (q->field == LFGPU_FIELD_P256) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "quad_bind_gh_all: not built for Fp256Base");
  if (logv > 40 || logw > 40 || ((size_t)1 << logv) < q->nv || nw == 0 || ((size_t)1 << logw) < nw || nw <= q->hmax)
    return lf_fail(c, LFGPU_ERR_ARG, "quad_bind_gh_all: table sizes (nw must exceed the largest hand index %zu)", q->hmax);
  LF_HIP(c, hipSetDevice(c->device));
  const int field = q->field;
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, (q->nv + 2 * nw) * 16 + 64, &sc));
  elt_t* d_eqg = (elt_t*)sc;
  elt_t* d_eqh0 = d_eqg + q->nv;
  elt_t* d_eqh1 = d_eqh0 + nw;
  const uint64_t zero[2] = {0, 0};
  LF_TRY(lfgpu_raw_eq2(c, field, logv, q->nv, h_G0, h_G1, alpha, d_eqg));
  LF_TRY(lfgpu_raw_eq2(c, field, logw, nw, h_H0, h_H0, zero, d_eqh0));  // EQ(H0, i) + 0 * (...)
  LF_TRY(lfgpu_raw_eq2(c, field, logw, nw, h_H1, h_H1, zero, d_eqh1));
  LF_HIP(c, hipMemsetAsync(d_acc, 0, 32, c->stream));
  const elt_t be{beta[0], beta[1]};
  u32 nb = (u32)((q->n + BGH_THREADS - 1) / BGH_THREADS);
  if (nb > 512) nb = 512;
  if (nb == 0) nb = 1;
  QD_DISPATCH(field, bind_gh_all_kernel, dim3(nb), dim3(BGH_THREADS), q->n, (const corner4*)q->d_morton, (const elt_t*)q->d_kvec,
              (const elt_t*)d_eqg, (const elt_t*)d_eqh0, (const elt_t*)d_eqh1, be, d_acc);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}
// the 4 words of d_acc (on the host) -> the field element
void lf_quad_bind_gh_all_fold(int field, const u64 w[4], uint64_t out[2]) {
  if (field == LFGPU_FIELD_GF2_128) {
    out[0] = w[0];
    out[1] = w[1];
  } else {  // recombine the limbs: sum_k w[k] 2^(32k) mod p (a sum of Montgomery images is the image of the sum)
    const elt_t sum = fp_reduce_limbs(w[0], w[1], w[2], w[3]);
    out[0] = sum.lo;
    out[1] = sum.hi;
  }
}

extern "C" int lfgpu_quad_bind_gh_all(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                                      const uint64_t beta[2], size_t logw, size_t nw, const void* h_H0, const void* h_H1,
                                      uint64_t out[2]) {
  if (!q || !out) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = q->c;
  u64* d_acc = (u64*)((uint8_t*)c->mailbox_d + 256);
  LF_TRY(lf_quad_bind_gh_all_enqueue(q, logv, h_G0, h_G1, alpha, beta, logw, nw, h_H0, h_H1, d_acc));
  u64 w[4];
  LF_HIP(c, hipMemcpyAsync(w, d_acc, 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  lf_quad_bind_gh_all_fold(q->field, w, out);
  return LFGPU_OK;
}

// ------------------------------------------------------------------ one sumcheck layer (host loop)
extern "C" int lfgpu_sumcheck_partials(lfgpu_ctx*, int, size_t, const void*, const void*, uint64_t*, uint64_t*);
extern "C" int lfgpu_qw_scatter(lfgpu_ctx*, int, size_t, const void*, const void*, int, const void*, size_t, void*);
int lf_qw_scatter_gf_into(lfgpu_ctx* c, size_t n, const void* d_hc, const void* d_vc, int hand, const void* d_Wother, void* d_QW);
int lf_sumcheck_partials_clean(lfgpu_ctx* c, int field, size_t n, void* d_QW, const void* d_W, uint64_t a0[2], uint64_t a2[2]);
extern "C" int lfgpu_dense_bind(lfgpu_ctx*, int, size_t, const uint64_t*, const void*, void*);
extern "C" int lfgpu_hquad_bind_h(lfgpu_ctx*, int, size_t, const void*, const void*, const uint64_t*, int, void*, void*, size_t*);
int lf_bind_both_cached(lfgpu_ctx* c, int field, size_t n0, const uint64_t r[2], const void* d_in, void* d_out, size_t n, const void* d_hc,
                        const void* d_vc, int hand, void* d_hc_out, void* d_vc_out, const u32* d_off_cached);

#include "hostfield.h"

extern "C" int lfgpu_sumcheck_layer(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                                    const uint64_t beta[2], size_t logw, size_t nw, void* d_W, const uint64_t wc_in[2][2],
                                    lfgpu_sc_round_fn round, void* user, uint64_t wc_out[2][2], uint64_t* g_out,
                                    uint64_t bound_quad[2]) {
  if (!q || !alpha || !beta || !d_W || !wc_in || !round || !wc_out || !g_out || nw == 0 || logw > 40 || nw > ((size_t)1 << logw) || nw <= q->hmax)
    return q ? lf_fail(q->c, LFGPU_ERR_ARG, "sumcheck_layer: bad argument (nw must exceed the largest hand index)") : LFGPU_ERR_ARG;
  lfgpu_ctx* c = q->c;
  const int field = q->field;
  if (field == LFGPU_FIELD_P256) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "sumcheck_layer: Fp256Base layers run inside the ZK driver (zk256.hip)");
  LF_HIP(c, hipSetDevice(c->device));
  const HostField F(c, field);
  // device state: HQUAD ping-pong, QW, out-of-place buffer for the first bind of hand 0 (context scratch: no
  // allocation per layer)
  void* sc = nullptr;
  const size_t nt = q->n;
  const size_t qw_bytes = nw * (field == LFGPU_FIELD_FP128 ? 32 : 16);  // Fp128 fused steps keep 4 limb words per target
  const size_t half = ((nw + 1) / 2) * 16;
  const size_t bytes = 2 * (nt * 8 + nt * 16) + qw_bytes + 4 * half + LF_SC_GRID_STATE_BYTES + 256;
  LF_TRY(lf_scratch(c, bytes, &sc));
  uint8_t* base = (uint8_t*)sc;
  void* hc[2] = {base, base + nt * 8};
  void* vc[2] = {base + 2 * nt * 8, base + 2 * nt * 8 + nt * 16};
  void* qw = base + 2 * nt * 8 + 2 * nt * 16;
  void* wtmp = (uint8_t*)qw + qw_bytes;  // 4 half-size hand buffers; the first is the detach buffer of the step paths
  void* grid_state = (uint8_t*)wtmp + 4 * half;
  size_t nh = 0;
  static const bool verbose = getenv("LFGPU_VERBOSE") != nullptr;
  auto clk = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_large = 0, t_small_first = 0, t_cb = 0, t_think = 0, t_seen = 0;
  size_t n_large = 0;
  int cur = 0;
  const elt_t al{alpha[0], alpha[1]};
  elt_t sum = F.add(elt_t{wc_in[0][0], wc_in[0][1]}, F.mul(al, elt_t{wc_in[1][0], wc_in[1][1]}));
  void* WH[2] = {d_W, d_W};
  size_t nW[2] = {nw, nw};
  // LFGPU_SC_MODE: how the rounds are driven once the HQUAD and both hand arrays are small enough
  //   grid (default)  one launch of co-resident workgroups on a grid that shrinks with the data (<= LF_SC_GRID_MAX entries)
  //   resident        one single-workgroup launch for the rest of the layer (<= LF_SC_SMALL_MAX entries)
  //   launch          one fused single-workgroup kernel per round-hand (<= LF_SC_SMALL_MAX entries)
  //   off             the multi-kernel path throughout
  // Larger rounds always take the multi-kernel path (whole-GPU kernels, 2 stream synchronisations per round-hand).
  static const int sc_mode = [] {
    const char* e = getenv("LFGPU_SC_MODE");
    if (!e) return 3;
    return !strcmp(e, "off") ? 0 : !strcmp(e, "launch") ? 1 : !strcmp(e, "resident") ? 2 : 3;
  }();
  const bool no_fuse = sc_mode == 0;
  const bool use_resident = sc_mode >= 2 && lf_sc_resident_ok(c);
  static const size_t grid_max = [] {  // hand-off point to the shrinking grid (entries of the largest array)
    const char* e = getenv("LFGPU_SC_GRID_MAX");
    const size_t v = e ? (size_t)atol(e) : (size_t)128 * 1024;
    return std::min<size_t>(std::max<size_t>(v, 1024), LF_SC_GRID_MAX);
  }();
  static const bool grid_max_from_env = getenv("LFGPU_SC_GRID_MAX") != nullptr;
  // throughput mode (several provers share the device): the per-launch path costs more there than alone (its many small
  // launches queue behind everybody else's), so the grid takes over as early as its state allows
  const size_t grid_max_eff = (!grid_max_from_env && lf_cu_sharers(c) >= 6) ? (size_t)LF_SC_GRID_MAX : grid_max;
  const size_t resident_max = sc_mode == 3 ? grid_max_eff : LF_SC_SMALL_MAX;  // largest array a resident kernel takes over at
  // Quad::bind_g: enqueued on the stream, nothing read back (the HQUAD size is a circuit constant)
  const double tv0 = verbose ? clk() : 0;
  // (the emit kernel also clears the head of the grid state: the resident grid then starts without a memset of its own)
  bool gs_clean = sc_mode == 3 && nt > 0;
  // ... and, for a GF(2^128) layer that starts on per-launch kernels, the QW accumulators of its first round-hand
  const bool qw_side = field == LFGPU_FIELD_GF2_128 && nt > 0 && (qw_bytes >> 35) == 0 && std::max<size_t>(q->nh0, nw) > std::max<size_t>(resident_max, LF_SC_SMALL_MAX);
  LF_TRY(lf_quad_bind_g(q, logv, h_G0, h_G1, alpha, beta, hc[0], vc[0], &nh, gs_clean ? grid_state : nullptr, gs_clean ? LF_SC_GRID_SYNC_CLEAR_BYTES : 0,
                        qw_side ? qw : nullptr, qw_side ? qw_bytes : 0));
  const double tv1 = verbose ? clk() : 0;
  const size_t nh0 = nh;
  bool resident = false, have_r = false;
  // A resident kernel holds CUs of the device's budget (ctx.h); whatever way this function is left, they go back -- after
  // the kernel has ended, so that the budget never undercounts workgroups that still spin.
  struct CuGuard {
    lfgpu_ctx* c;
    ~CuGuard() {
      if (c->cu_held) {
        (void)hipStreamSynchronize(c->stream);  // bounded: every wait of the kernel has a timeout
        lf_cu_release(c, -1);
      }
    }
  } cu_guard{c};
  bool resident_off = !use_resident || no_fuse || c->grid_strikes >= 2;  // no (further) hand-off to a resident kernel in this layer
  u32 grid_G = 0, grid_per_wg = 1;  // the grid's workgroups as the kernel shrinks them (mirrored here to return CUs early)
  u64 last_r[2] = {0, 0};
  bool qw_clean = qw_side;  // GF(2^128), per-launch path: `qw` is all zero (cleared beside bind_g or by a memset, kept by the self-cleaning sums)
  bool pending = false;  // the binds of (phand, pr) ride in the next fused step
  int phand = 0;
  elt_t pr{0, 0};
  // the fused step with the current host-side state; applies the pending bind's bookkeeping after it ran
  auto small_step = [&](int do_eval, int eval_hand, u64 out[8]) -> int {
    ScSmall a{};
    a.field = field;
    a.do_bind = pending ? 1 : 0;
    a.bind_hand = phand;
    a.r = pr;
    a.do_eval = do_eval;
    a.eval_hand = eval_hand;
    a.hc_in = (uint2*)hc[cur];
    a.vc_in = (elt_t*)vc[cur];
    a.hc_out = (uint2*)hc[1 - cur];
    a.vc_out = (elt_t*)vc[1 - cur];
    a.nh = (u32)nh;
    a.W[0] = (elt_t*)WH[0];
    a.W[1] = (elt_t*)WH[1];
    a.nW[0] = (u32)nW[0];
    a.nW[1] = (u32)nW[1];
    a.Wdst = (elt_t*)((pending && phand == 0 && WH[0] == d_W) ? wtmp : WH[phand]);  // hand 0 detaches from the shared input
    a.QW = (u64*)qw;
    LF_TRY(lf_sc_small_step(c, a, out));
    if (pending) {
      WH[phand] = a.Wdst;
      nW[phand] = (nW[phand] + 1) / 2;
      nh = (size_t)out[4];
      cur = 1 - cur;
      pending = false;
    }
    return LFGPU_OK;
  };
  for (size_t rnd = 0; rnd < logw; ++rnd) {
    for (int hand = 0; hand < 2; ++hand) {
      uint64_t a0[2], a2[2];
      bool fused = false;  // this round-hand ran as a fused single-workgroup step: its binds ride in the next one
      const double tr0 = verbose ? clk() : 0;
      // ---- hand the rest of the layer to a resident kernel once the arrays are small enough AND the device's CU budget has
      // room for its workgroups; without room this round-hand runs on per-launch kernels and the next one asks again
      if (!resident && !resident_off && std::max(nh, std::max(nW[0], nW[1])) <= resident_max) {
        // (with a bind still pending the sizes above are the ones before it: the bind only halves them)
        if (pending) {
          u64 out[8];
          LF_TRY(small_step(0, 0, out));
        }
        int rc;
        if (sc_mode == 3) {  // the shrinking grid
          uint8_t* wb = (uint8_t*)wtmp;
          rc = lf_sc_grid_begin(c, field, hc[cur], vc[cur], hc[1 - cur], vc[1 - cur], nh, nullptr, WH[0], nW[0], WH[1], nW[1], wb, wb + half, wb + 2 * half,
                                wb + 3 * half, qw, 2 * rnd + hand, logw, grid_state, &q->grid_off, &grid_G, &grid_per_wg, gs_clean);
          if (rc == LFGPU_OK) gs_clean = false;  // (a refusal by the CU budget launches nothing: the state stays clean)
        } else if (lf_cu_acquire(c, 1)) {  // the resident single workgroup
          ScSmall a{};
          a.field = field;
          a.hc_in = (uint2*)hc[cur];
          a.vc_in = (elt_t*)vc[cur];
          a.hc_out = (uint2*)hc[1 - cur];
          a.vc_out = (elt_t*)vc[1 - cur];
          a.nh = (u32)nh;
          a.W[0] = (elt_t*)WH[0];
          a.W[1] = (elt_t*)WH[1];
          a.nW[0] = (u32)nW[0];
          a.nW[1] = (u32)nW[1];
          a.QW = (u64*)qw;
          grid_G = 1;
          rc = lf_sc_layer_begin(c, a, (u32)(2 * rnd + hand), (u32)(2 * logw), d_W, wtmp);
        } else {
          rc = LFGPU_ERR_BUSY;
        }
        qw_clean = false;  // a resident kernel (even one that is not placed and leaves) uses `qw` as its own scratch
        if (rc == LFGPU_OK) {
          resident = true;
          have_r = false;
          if (verbose && t_small_first == 0) t_small_first = clk();
        } else if (rc != LFGPU_ERR_BUSY) {
          return rc;
        }
      }
      if (resident) {
        u64 out[8];
        if (verbose && t_seen != 0) t_think += clk() - t_seen;  // post seen -> answer written: the host's share of the round trip
        const int rc = lf_sc_layer_next(c, have_r ? last_r : nullptr, out);
        if (rc == LFGPU_ERR_BUSY && !have_r) {
          // The grid's workgroups were not placed together (its first barrier is that check, and nothing but scratch is
          // written before it).  The budget makes this impossible within one process; another process on the device can
          // still hold CUs.  The kernel has left: drive this layer with per-launch kernels from here, on the same state.
          LF_HIP(c, hipStreamSynchronize(c->stream));
          lf_cu_release(c, -1);
          resident = false;
          resident_off = true;
          ++c->grid_strikes;
          if (q->grid_off.state == 1) q->grid_off.state = 0;  // a record that never completed
          if (verbose) fprintf(stderr, "lfgpu sumcheck_layer: resident grid not placed (strike %d): per-launch kernels for this layer\n", c->grid_strikes);
        } else if (rc != LFGPU_OK) {
          return rc;
        } else {
          if (verbose) t_seen = clk();
          a0[0] = out[0]; a0[1] = out[1]; a2[0] = out[2]; a2[1] = out[3];
          if (sc_mode == 3 && grid_G > 1) {  // the kernel has shrunk its grid to ceil(largest array / per_wg) before this post: those CUs are free
            nh = (size_t)out[4];
            const size_t big = std::max(nh, std::max(nW[0], nW[1]));
            const u32 want = (u32)std::max<size_t>(1, (big + grid_per_wg - 1) / grid_per_wg);
            if (want < grid_G) {
              lf_cu_release(c, (int)(grid_G - want));
              grid_G = want;
            }
          }
        }
      }
      if (!resident) {
        if (!no_fuse && std::max(nh, std::max(nW[0], nW[1])) <= LF_SC_SMALL_MAX) {  // one fused single-workgroup launch per round-hand
          if (verbose && t_small_first == 0) t_small_first = clk();
          u64 out[8];
          LF_TRY(small_step(1, hand, out));
          a0[0] = out[0]; a0[1] = out[1]; a2[0] = out[2]; a2[1] = out[3];
          fused = true;
        } else {
          if (field == LFGPU_FIELD_GF2_128) {
            // the accumulators clean themselves (the sums' kernel zeroes what it has read): one memset per layer, not per round-hand
            if (!qw_clean) {
              LF_HIP(c, hipMemsetAsync(qw, 0, qw_bytes, c->stream));
              qw_clean = true;
            }
            LF_TRY(lf_qw_scatter_gf_into(c, nh, hc[cur], vc[cur], hand, WH[1 - hand], qw));
            LF_TRY(lf_sumcheck_partials_clean(c, field, nW[hand], qw, WH[hand], a0, a2));
          } else {
            LF_TRY(lfgpu_qw_scatter(c, field, nh, hc[cur], vc[cur], hand, WH[1 - hand], nW[hand], qw));
            LF_TRY(lfgpu_sumcheck_partials(c, field, nW[hand], qw, WH[hand], a0, a2));
          }
        }
      }
      // coef[0] = eq0*a0, coef[2] = eq0*a2 with eq0 = 1 (logc = 0); coef[1] from sum (prover_layers.h:390-396)
      elt_t coef[3];
      coef[0] = elt_t{a0[0], a0[1]};
      coef[2] = elt_t{a2[0], a2[1]};
      coef[1] = F.sub(F.sub(F.sub(sum, coef[0]), coef[0]), coef[2]);
      elt_t ev[3];
      uint64_t evw[3][2], r[2];
      for (int k = 0; k < 3; ++k) {
        ev[k] = F.eval_monomial(coef, F.pts[k]);
        evw[k][0] = ev[k].lo;
        evw[k][1] = ev[k].hi;
      }
      const double tcb0 = verbose ? clk() : 0;
      round(user, (size_t)hand, rnd, evw, r);
      if (verbose) t_cb += clk() - tcb0;
      g_out[(hand * logw + rnd) * 2] = r[0];
      g_out[(hand * logw + rnd) * 2 + 1] = r[1];
      sum = F.eval_lagrange(ev, elt_t{r[0], r[1]});
      if (resident) {  // the challenge goes to the resident kernel with the next wait
        last_r[0] = r[0];
        last_r[1] = r[1];
        have_r = true;
        nW[hand] = (nW[hand] + 1) / 2;  // what the kernel does with it (the host only mirrors the sizes)
        continue;
      }
      if (fused) {  // the binds run at the head of the next fused step
        pending = true;
        phand = hand;
        pr = elt_t{r[0], r[1]};
        continue;
      }
      bool both = false;
      static const bool bind_split = getenv("LFGPU_SC_BIND_SPLIT") && atoi(getenv("LFGPU_SC_BIND_SPLIT")) != 0;  // A/B: two launches
      {
        // Dense::bind out of place, ping-pong between the hand's two half-size buffers (the ones the grid kernel binds into after
        // the hand-off): an in-place bind needs a temporary and a device-to-device copy back -- one more dispatch in a chain where
        // every dispatch counts.  Hand 0's first bind leaves the shared input as before (prover_layers.h:222-226,
        // 255-257); hand 1 no longer overwrites it.
        uint8_t* const wb = (uint8_t*)wtmp;
        void* dst = wb + (size_t)(2 * hand) * half;
        if (dst == WH[hand]) dst = wb + (size_t)(2 * hand + 1) * half;
        const size_t rh = 2 * rnd + hand;
        if (q->bind_shape.size() < 2 * logw) q->bind_shape.resize(2 * logw, lfgpu_quad::BindShape{nullptr, 0, 0});
        const lfgpu_quad::BindShape& bs0 = q->bind_shape[rh];
        both = bs0.d_off && bs0.n_in == nh && nh > 0 && !bind_split;  // recorded merge structure: both binds in one launch
        if (both) LF_TRY(lf_bind_both_cached(c, field, nW[hand], r, WH[hand], dst, nh, hc[cur], vc[cur], hand, hc[1 - cur], vc[1 - cur], bs0.d_off));
        else LF_TRY(lfgpu_dense_bind(c, field, nW[hand], r, WH[hand], dst));
        WH[hand] = dst;
      }
      nW[hand] = (nW[hand] + 1) / 2;
      {  // HQuad::bind_h: the merge structure of this round-hand is a circuit constant, kept from the first proof on
        const size_t rh = 2 * rnd + hand;
        lfgpu_quad::BindShape& bs = q->bind_shape[rh];
        if (both) {
          nh = bs.n_out;
        } else if (bs.d_off && bs.n_in == nh) {
          LF_TRY(lf_hquad_bind_h_cached(c, field, nh, hc[cur], vc[cur], r, hand, hc[1 - cur], vc[1 - cur], bs.d_off, nullptr, nullptr));
          nh = bs.n_out;
        } else {
          if (bs.d_off) (void)hipFree(bs.d_off);
          bs = lfgpu_quad::BindShape{nullptr, nh, 0};
          LF_TRY(lf_hquad_bind_h_cached(c, field, nh, hc[cur], vc[cur], r, hand, hc[1 - cur], vc[1 - cur], nullptr, &bs.d_off, &nh));
          bs.n_out = nh;
        }
      }
      cur = 1 - cur;
      if (verbose) {
        t_large += clk() - tr0;
        ++n_large;
      }
    }
  }
  uint64_t tmp[6];
  if (resident) {  // last challenge in, W[0][0], W[1][0] and the HQUAD scalar out
    u64 out[8];
    LF_TRY(lf_sc_layer_next(c, last_r, out));
    if (q->grid_off.state == 1) q->grid_off.state = 2;  // the layer ran to its end: the recorded bind offsets are good
    lf_cu_release(c, -1);  // the final post is the kernel's last action
    tmp[0] = out[0]; tmp[1] = out[1]; tmp[2] = out[2]; tmp[3] = out[3]; tmp[4] = out[6]; tmp[5] = out[7];
  } else if (pending) {  // last binds + read-out of W[0][0], W[1][0], HQUAD scalar in one step
    u64 out[8];
    LF_TRY(small_step(0, 0, out));
    tmp[0] = out[0]; tmp[1] = out[1]; tmp[2] = out[2]; tmp[3] = out[3]; tmp[4] = out[6]; tmp[5] = out[7];
  } else {
    LF_HIP(c, hipMemcpyAsync(tmp, WH[0], 16, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(c, hipMemcpyAsync(tmp + 2, WH[1], 16, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(c, hipMemcpyAsync(tmp + 4, vc[cur], 16, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(c, hipStreamSynchronize(c->stream));
  }
  if (verbose)
    fprintf(stderr, "lfgpu sumcheck_layer: nterms %zu nh0 %zu nw %zu logw %zu | bind_g %.0f us | %zu large round-hands %.0f us | %zu small %.0f us | caller's round callback %.1f us in all, host between post and answer %.1f us\n",
            nt, nh0, nw, logw, tv1 - tv0, n_large, t_large, 2 * logw - n_large, t_small_first ? clk() - t_small_first : 0.0, t_cb, t_think);
  wc_out[0][0] = tmp[0];
  wc_out[0][1] = tmp[1];
  wc_out[1][0] = tmp[2];
  wc_out[1][1] = tmp[3];
  if (bound_quad) {
    bound_quad[0] = tmp[4];
    bound_quad[1] = tmp[5];
  }
  return LFGPU_OK;
}
