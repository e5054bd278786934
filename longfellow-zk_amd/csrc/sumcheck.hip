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
template <int F>
__global__ __launch_bounds__(SC_THREADS) void sumcheck_partials_kernel(size_t n, const elt_t* __restrict__ QW,
                                                                       const elt_t* __restrict__ W,
                                                                       elt_t* __restrict__ partial) {
  __shared__ elt_t sh[SC_THREADS / 64];
  const size_t nodd = n / 2;
  elt_t a0 = elt_zero(), a2 = elt_zero();
  for (size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x; i < nodd; i += (size_t)gridDim.x * SC_THREADS) {
    elt_t q0 = ld16(&QW[2 * i]), q1 = ld16(&QW[2 * i + 1]);
    elt_t w0 = ld16(&W[2 * i]), w1 = ld16(&W[2 * i + 1]);
    a0 = Fld<F>::add(a0, Fld<F>::mul(q0, w0));
    a2 = Fld<F>::add(a2, Fld<F>::mul(Fld<F>::sub(q1, q0), Fld<F>::sub(w1, w0)));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && 2 * nodd < n) {  // odd tail (prover_layers.h:381-388)
    elt_t t = Fld<F>::mul(ld16(&QW[2 * nodd]), ld16(&W[2 * nodd]));
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
template <int F>
__global__ __launch_bounds__(SC_THREADS) void sumcheck_final_kernel(u32 nblocks, const elt_t* __restrict__ partial,
                                                                    elt_t* __restrict__ out) {
  __shared__ elt_t sh[SC_THREADS / 64];
  elt_t a0 = elt_zero(), a2 = elt_zero();
  for (u32 b = threadIdx.x; b < nblocks; b += SC_THREADS) {
    a0 = Fld<F>::add(a0, ld16(&partial[2 * b]));
    a2 = Fld<F>::add(a2, ld16(&partial[2 * b + 1]));
  }
  a0 = block_reduce<F>(a0, sh);
  a2 = block_reduce<F>(a2, sh);
  if (threadIdx.x == 0) {
    st16(&out[0], a0);
    st16(&out[1], a2);
  }
}

// QW[h[hand]] ^= v * Wother[h[1-hand]]  -- GF(2^128): addition is XOR, so two 64-bit atomic XORs per
// term are exact and order-independent.  One wire (the constant 1) is the target of up to 2*10^5 terms of a
// flatsha256 layer and its terms are mostly adjacent in canonical order, so equal-target neighbours are first
// folded inside the wave with shuffles; only the first lane of each run of equal targets issues atomics.
__global__ __launch_bounds__(SC_THREADS) void qw_scatter_gf_kernel(size_t n, const uint2* __restrict__ hc,
                                                                   const elt_t* __restrict__ vc, int hand,
                                                                   const elt_t* __restrict__ Wo, u64* __restrict__ QW) {
  const size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  const u32 lane = threadIdx.x & 63;
  const bool valid = i < n;
  u32 key = 0xffffffffu;
  elt_t t = elt_zero();
  if (valid) {
    uint2 h = hc[i];
    key = hand ? h.y : h.x;
    t = gf_mul(ld16(&vc[i]), ld16(&Wo[hand ? h.x : h.y]));
  }
  // runs of CONTIGUOUS equal targets: run id = number of run heads at or before the lane (monotone), so
  // "same run id" implies every lane in between has the same target and the suffix fold below is exact
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
// S = sum_k acc[k] * 2^(32k) mod p on PLAIN integers (S is then again a Montgomery image, because images add).
// acc[k] = lo32 + hi32 * 2^32; piece * 2^(32j) mod p = fp_mul(piece, c[j]) with c[j] = image of 2^(32j).
__global__ __launch_bounds__(SC_THREADS) void fp_limb_normalize_kernel(size_t n, const u64* __restrict__ acc,
                                                                       const elt_t* __restrict__ c /*5 constants*/,
                                                                       elt_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (i >= n) return;
  elt_t sum = elt_zero();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const u64 a = acc[4 * i + k];
    const elt_t lo = fp_mul(elt_t{(u64)(u32)a, 0}, ld16(&c[k]));
    const elt_t hi = fp_mul(elt_t{a >> 32, 0}, ld16(&c[k + 1]));
    sum = fp_add(sum, fp_add(lo, hi));
  }
  st16(&out[i], sum);
}

// out[i] = in[2i] + r*(in[2i+1]-in[2i]);  tail: in*(1-r)   (dense.h:70-87, affine.h:26-52)
template <int F>
__global__ __launch_bounds__(SC_THREADS) void dense_bind_kernel(size_t n0, elt_t r, const elt_t* __restrict__ in,
                                                                elt_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
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
__global__ __launch_bounds__(SC_THREADS) void hquad_emit_kernel(size_t n, const uint2* __restrict__ hc,
                                                                const elt_t* __restrict__ vc, elt_t r, int hand,
                                                                const u32* __restrict__ block_off,
                                                                uint2* __restrict__ hc_out, elt_t* __restrict__ vc_out) {
  __shared__ u32 wave_off[SC_THREADS / 64];
  size_t i = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  bool head = i < n && !is_second(hc, i, hand);
  u64 mask = __ballot(head);
  const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wave_off[wave] = (u32)__popcll(mask);
  __syncthreads();
  u32 off = block_off[blockIdx.x];
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

// y[j] += sum_i u[i]*T[i][j]
template <int F>
__global__ __launch_bounds__(SC_THREADS) void rows_axpy_kernel(u32 nrows, size_t n, elt_t* __restrict__ y,
                                                               const elt_t* __restrict__ u,
                                                               const elt_t* __restrict__ T, size_t ld) {
  size_t j = (size_t)blockIdx.x * SC_THREADS + threadIdx.x;
  if (j >= n) return;
  elt_t acc = ld16(&y[j]);
  for (u32 i = 0; i < nrows; ++i) acc = Fld<F>::add(acc, Fld<F>::mul(ld16(&T[(size_t)i * ld + j]), ld16(&u[i])));
  st16(&y[j], acc);
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

extern "C" int lfgpu_sumcheck_partials(lfgpu_ctx* c, int field, size_t n, const void* d_QW, const void* d_W,
                                       uint64_t a0[2], uint64_t a2[2]) {
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
                 (const elt_t*)d_W, partial);
  DISPATCH_FIELD(field, sumcheck_final_kernel, dim3(1), dim3(SC_THREADS), nb, (const elt_t*)partial, out);
  LF_HIP(c, hipGetLastError());
  LF_HIP(c, hipMemcpyAsync(c->mailbox_h, out, 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(a0, c->mailbox_h, 16);
  memcpy(a2, (const uint8_t*)c->mailbox_h + 16, 16);
  return LFGPU_OK;
}

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
    void* dconst = nullptr;
    if (!lf_table_lookup(c, "fp:pow2_32j", &dconst)) {
      elt_t cs[5];  // Montgomery images of 2^(32j), j = 0..4
      cs[0] = h_fp_of_scalar(1);
      const elt_t two32 = h_fp_of_scalar(1ull << 32);
      for (int j = 1; j < 5; ++j) cs[j] = fp_mul(cs[j - 1], two32);
      LF_TRY(lf_table(c, "fp:pow2_32j", cs, sizeof(cs), &dconst));
    }
    LF_HIP(c, hipMemsetAsync(acc, 0, nqw * 32, c->stream));
    if (n) {
      u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
      hipLaunchKernelGGL(qw_scatter_fp_kernel, dim3(nb), dim3(SC_THREADS), 0, c->stream, n, (const uint2*)d_hc,
                         (const elt_t*)d_vc, hand ? 1 : 0, (const elt_t*)d_Wother, acc);
    }
    u32 nb2 = (u32)((nqw + SC_THREADS - 1) / SC_THREADS);
    hipLaunchKernelGGL(fp_limb_normalize_kernel, dim3(nb2), dim3(SC_THREADS), 0, c->stream, nqw, (const u64*)acc,
                       (const elt_t*)dconst, (elt_t*)d_QW);
    LF_HIP(c, hipGetLastError());
    return LFGPU_OK;
  }
  LF_HIP(c, hipMemsetAsync(d_QW, 0, nqw * 16, c->stream));
  if (n) {
    u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
    hipLaunchKernelGGL(qw_scatter_gf_kernel, dim3(nb), dim3(SC_THREADS), 0, c->stream, n, (const uint2*)d_hc,
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

extern "C" int lfgpu_rows_axpy(lfgpu_ctx* c, int field, size_t nrows, size_t n, void* d_y, const uint64_t* h_u,
                               const void* d_T, size_t ld) {
  if (!c || (n && !d_y) || (nrows && (!h_u || !d_T))) return lf_fail(c, LFGPU_ERR_ARG, "rows_axpy: null argument");
  if (n == 0 || nrows == 0) return LFGPU_OK;
  LF_HIP(c, hipSetDevice(c->device));
  void* du = nullptr;
  LF_TRY(lf_scratch2(c, nrows * 16, &du));
  LF_HIP(c, hipMemcpyAsync(du, h_u, nrows * 16, hipMemcpyHostToDevice, c->stream));
  u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
  DISPATCH_FIELD(field, rows_axpy_kernel, dim3(nb), dim3(SC_THREADS), (u32)nrows, n, (elt_t*)d_y, (const elt_t*)du,
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
  u32 nb = (u32)((n + SC_THREADS - 1) / SC_THREADS);
  DISPATCH_FIELD(field, field_binop_kernel, dim3(nb), dim3(SC_THREADS), op, n, (const elt_t*)d_a, (const elt_t*)d_b,
                 (elt_t*)d_out);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}
