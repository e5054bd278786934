// p256.hip -- the P-256 base-field leg of the path (field id 1, 32-byte elements): BASELINE config 5's signature
// tableau, ZkProver<Fp256Base, ReedSolomonFactory<Fp256Base, FFTExtConvolutionFactory>> (reference
// lib/circuits/mdoc/mdoc_zk.cc:70-88,485-500).
//
//   K4' lfgpu_fp256_rs_encode_rows : ReedSolomon<Fp256Base, FFTExtConvolution>::interpolate over the rows of a tableau
//        (lib/algebra/reed_solomon.h:51-110, convolution.h:129-191, rfft.h:282-376)
//   K5' column_leaves32_kernel     : LigeroCommon<Fp256Base>::column_hash (lib/ligero/ligero_param.h:432-439)
//   field binops                   : FpGeneric::addf / subf / mulf for the parity tests
//
// The reference computes the convolution with a REAL fft in half-complex storage (RFFT over Fp2, because Fp256 itself
// has no 2^k-th roots of unity: p - 1 = 2 * odd).  The convolution is a well-defined element of the field, so any exact
// method returns the same 32 bytes; the device uses what maps to a GPU: two tableau rows a, b travel as ONE complex
// sequence a + i b through a radix-2 FFT over Fp2 = Fp256[i] (decimation in frequency forward, decimation in time
// backward, so no bit-reversal pass), the kernel y (1/k) is real, hence conv(a + i b, y) = conv(a, y) + i conv(b, y).
// Twiddles: powers of the reference's root of order 2^31 on the unit circle (mdoc_zk.cc:82-88): w^-1 = conj(w).
#include <string>

#include "ctx.h"
#include "fp256.h"

#define P256_LOCAL_LOG 9  // points per LDS tile of the local FFT kernel: 512 x 64 B = 32 KiB

LF_HD fp2_t fp2_conj(const fp2_t& a) { return fp2_t{a.re, fp256_neg(a.im)}; }
__device__ inline fp2_t ldc(const fp2_t* p) { return fp2_t{ld32(&p->re), ld32(&p->im)}; }
__device__ inline void stc(fp2_t* p, const fp2_t& v) {
  st32(&p->re, v.re);
  st32(&p->im, v.im);
}

// ------------------------------------------------------------------ FFT over Fp2
// one radix-2 stage on global memory, half-distance h; W[j] = w_P^j (j < P/2)
//   forward (DIF): (u, v) -> (u + v, (u - v) conj(W[k P/2h]))      backward (DIT): (u, v W[k P/2h]) -> (u + v, u - v)
template <bool FWD>
__global__ void fp2_fft_stage_kernel(fp2_t* __restrict__ Z, u32 P, u32 h, const fp2_t* __restrict__ W) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P / 2) return;
  const u32 k = t & (h - 1), i = ((t - k) << 1) | k;
  fp2_t* z = Z + (size_t)blockIdx.y * P;
  const fp2_t w = ldc(&W[(size_t)k * (P / (2 * h))]);
  fp2_t u = ldc(&z[i]), v = ldc(&z[i + h]);
  if (FWD) {
    stc(&z[i], fp2_add(u, v));
    stc(&z[i + h], fp2_mul(fp2_sub(u, v), fp2_conj(w)));
  } else {
    v = fp2_mul(v, w);
    stc(&z[i], fp2_add(u, v));
    stc(&z[i + h], fp2_sub(u, v));
  }
}
// the stages with half-distance < 2^tl on a contiguous tile of 2^tl points in LDS (forward: the last tl stages,
// backward: the first tl)
template <bool FWD>
__global__ __launch_bounds__(256) void fp2_fft_local_kernel(fp2_t* __restrict__ Z, u32 P, u32 tl, const fp2_t* __restrict__ W) {
  extern __shared__ uint4 lds4[];
  fp2_t* s = reinterpret_cast<fp2_t*>(lds4);
  const u32 L = 1u << tl, tid = threadIdx.x;
  fp2_t* z = Z + (size_t)blockIdx.y * P + (size_t)blockIdx.x * L;
  for (u32 i = tid; i < L; i += 256) s[i] = ldc(&z[i]);
  __syncthreads();
  for (u32 st = 0; st < tl; ++st) {
    const u32 h = FWD ? (L >> (st + 1)) : (1u << st);
    for (u32 t = tid; t < L / 2; t += 256) {
      const u32 k = t & (h - 1), i = ((t - k) << 1) | k;
      const fp2_t w = ldc(&W[(size_t)k * (P / (2 * h))]);
      fp2_t u = s[i], v = s[i + h];
      if (FWD) {
        s[i] = fp2_add(u, v);
        s[i + h] = fp2_mul(fp2_sub(u, v), fp2_conj(w));
      } else {
        v = fp2_mul(v, w);
        s[i] = fp2_add(u, v);
        s[i + h] = fp2_sub(u, v);
      }
    }
    __syncthreads();
  }
  for (u32 i = tid; i < L; i += 256) stc(&z[i], s[i]);
}

static int fp2_fft(lfgpu_ctx* c, bool fwd, size_t batch, u32 P, fp2_t* Z, const fp2_t* W) {
  if (P < 2 || batch == 0) return LFGPU_OK;
  const u32 lp = lf_log2(P), tl = lp < P256_LOCAL_LOG ? lp : P256_LOCAL_LOG;
  const dim3 gs((P / 2 + 255) / 256, (u32)batch), gl(P >> tl, (u32)batch);
  const size_t lds = (size_t)64 << tl;
  if (fwd) {
    for (u32 h = P / 2; h >= (1u << tl); h >>= 1) hipLaunchKernelGGL(fp2_fft_stage_kernel<true>, gs, dim3(256), 0, c->stream, Z, P, h, W);
    hipLaunchKernelGGL(fp2_fft_local_kernel<true>, gl, dim3(256), lds, c->stream, Z, P, tl, W);
  } else {
    hipLaunchKernelGGL(fp2_fft_local_kernel<false>, gl, dim3(256), lds, c->stream, Z, P, tl, W);
    for (u32 h = 1u << tl; h <= P / 2; h <<= 1) hipLaunchKernelGGL(fp2_fft_stage_kernel<false>, gs, dim3(256), 0, c->stream, Z, P, h, W);
  }
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

// ------------------------------------------------------------------ Reed-Solomon rows
// Z[pair][i] = binom[i] * (T[2 pair][i] + i T[2 pair + 1][i]) for i < n, 0 up to P  (reed_solomon.h:101-104)
__global__ void p256_rs_pre_kernel(u32 n, u32 P, u32 nrow, const elt32_t* __restrict__ binom, const elt32_t* __restrict__ T, size_t ld,
                                   fp2_t* __restrict__ Z) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const u32 r0 = 2 * blockIdx.y, r1 = r0 + 1;
  fp2_t v{e32_zero(), e32_zero()};
  if (i < n) {
    const elt32_t b = ld32(&binom[i]);
    v.re = fp256_mul(b, ld32(&T[(size_t)r0 * ld + i]));
    if (r1 < nrow) v.im = fp256_mul(b, ld32(&T[(size_t)r1 * ld + i]));
  }
  stc(&Z[(size_t)blockIdx.y * P + i], v);
}
// Z[k] *= yhat[k] (both in the forward transform's bit-reversed order)
__global__ void p256_rs_pointwise_kernel(u32 P, const fp2_t* __restrict__ yhat, fp2_t* __restrict__ Z) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  fp2_t* z = &Z[(size_t)blockIdx.y * P + i];
  stc(z, fp2_mul(ldc(z), ldc(&yhat[i])));
}
// y[k] = lead[k - (n-1)] * conv[k] for n <= k < m  (reed_solomon.h:106-109)
__global__ void p256_rs_post_kernel(u32 n, u32 m, u32 P, u32 nrow, const elt32_t* __restrict__ lead, const fp2_t* __restrict__ Z,
                                    elt32_t* __restrict__ T, size_t ld) {
  const u32 i = n + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const u32 r0 = 2 * blockIdx.y, r1 = r0 + 1;
  const elt32_t l = ld32(&lead[i - (n - 1)]);
  const fp2_t z = ldc(&Z[(size_t)blockIdx.y * P + i]);
  st32(&T[(size_t)r0 * ld + i], fp256_mul(l, z.re));
  if (r1 < nrow) st32(&T[(size_t)r1 * ld + i], fp256_mul(l, z.im));
}

// host-side field helpers: fp256.h
static elt32_t h256_dec(const char* s) {
  elt32_t r = e32_zero();
  const elt32_t ten = h256_of_scalar(10);
  for (; *s; ++s) r = fp256_add(fp256_mul(r, ten), h256_of_scalar((u64)(*s - '0')));
  return r;
}

// W[j] = w_P^j, j < P/2, w_P = the reference's root of order 2^31 re-rooted (lib/algebra/twiddle.h:36-55)
static int p256_twiddles(lfgpu_ctx* c, u32 P, const fp2_t** out) {
  char kb[48];
  snprintf(kb, sizeof(kb), "p256tw:%u", P);
  void* d = nullptr;
  if (!lf_table_lookup(c, kb, &d)) {
    fp2_t w{h256_dec("112649224146410281873500457609690258373018840430489408729223714171582664680802"),
            h256_dec("84087994358540907695740461427818660560182168997182378749313018254450460212908")};
    for (u64 r = P; r < ((u64)1 << 31); r += r) w = fp2_mul(w, w);
    std::vector<fp2_t> t(P / 2 ? P / 2 : 1);
    fp2_t x{h256_of_scalar(1), e32_zero()};
    for (u32 i = 0; 2 * i < P; ++i) {
      t[i] = x;
      x = fp2_mul(x, w);
    }
    LF_TRY(lf_table(c, kb, t.data(), t.size() * sizeof(fp2_t), &d));
  }
  *out = (const fp2_t*)d;
  return LFGPU_OK;
}

extern "C" int lfgpu_fp256_rs_encode_rows(lfgpu_ctx* c, size_t nrow, size_t n, size_t m, void* d_T, size_t ld) {
  if (!c || (!d_T && nrow)) return lf_fail(c, LFGPU_ERR_ARG, "fp256_rs_encode_rows: null argument");
  if (n == 0 || m < n || ld < m) return lf_fail(c, LFGPU_ERR_ARG, "fp256_rs_encode_rows: need 0 < n <= m <= ld");
  if (nrow == 0 || m == n) return LFGPU_OK;
  size_t P = 1;
  while (P < m) P <<= 1;
  if (P > ((size_t)1 << 24)) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "fp256_rs_encode_rows: m > 2^24");
  LF_HIP(c, hipSetDevice(c->device));
  const fp2_t* W = nullptr;
  LF_TRY(p256_twiddles(c, (u32)P, &W));
  const size_t d = n - 1;
  char kb[96];
  snprintf(kb, sizeof(kb), "p256rs:%zu:%zu", n, m);
  const std::string key(kb);
  void *dbinom = nullptr, *dlead = nullptr, *dyhat = nullptr;
  if (!lf_table_lookup(c, key + ":yhat", &dyhat)) {
    // constants of the ReedSolomon ctor (reed_solomon.h:51-88), batch inverse as AlgebraUtil::batch_inverse_arithmetic
    std::vector<elt32_t> inv(m), lead(m - n + 1), binom(n);
    const elt32_t one = h256_of_scalar(1), zero = e32_zero();
    {
      std::vector<elt32_t> pre(m);
      elt32_t acc = one;
      for (size_t i = 1; i < m; ++i) {
        pre[i] = acc;
        acc = fp256_mul(acc, h256_of_scalar(i));
      }
      elt32_t ia = m > 1 ? h256_inv(acc) : acc;
      inv[0] = zero;
      for (size_t i = m; i-- > 1;) {
        inv[i] = fp256_mul(ia, pre[i]);
        ia = fp256_mul(ia, h256_of_scalar(i));
      }
    }
    lead[0] = one;
    binom[0] = one;
    for (size_t i = 1; i + d < m; ++i) lead[i] = fp256_mul(lead[i - 1], fp256_mul(h256_of_scalar(d + i), inv[i]));
    for (size_t kk = d; kk < m; ++kk) {
      lead[kk - d] = fp256_mul(lead[kk - d], h256_of_scalar(kk - d));
      if (d % 2 == 1) lead[kk - d] = fp256_neg(lead[kk - d]);
    }
    for (size_t i = 1; i < n; ++i) binom[i] = fp256_mul(binom[i - 1], fp256_mul(h256_of_scalar(n - i), inv[i]));
    for (size_t i = 1; i < n; i += 2) binom[i] = fp256_neg(binom[i]);
    // yhat = forward transform of the padded inverses, divided by P (convolution.h:136-152), in bit-reversed order
    std::vector<fp2_t> yh(P, fp2_t{zero, zero});
    for (size_t i = 0; i < m; ++i) yh[i].re = inv[i];
    void* tmp = nullptr;
    LF_TRY(lf_scratch2(c, P * sizeof(fp2_t), &tmp));
    LF_HIP(c, hipMemcpy(tmp, yh.data(), P * sizeof(fp2_t), hipMemcpyHostToDevice));
    LF_TRY(fp2_fft(c, true, 1, (u32)P, (fp2_t*)tmp, W));
    LF_HIP(c, hipStreamSynchronize(c->stream));
    LF_HIP(c, hipMemcpy(yh.data(), tmp, P * sizeof(fp2_t), hipMemcpyDeviceToHost));
    const elt32_t sc = h256_inv(h256_of_scalar(P));
    for (size_t i = 0; i < P; ++i) yh[i] = fp2_t{fp256_mul(yh[i].re, sc), fp256_mul(yh[i].im, sc)};
    LF_TRY(lf_table(c, key + ":binom", binom.data(), binom.size() * 32, &dbinom));
    LF_TRY(lf_table(c, key + ":lead", lead.data(), lead.size() * 32, &dlead));
    LF_TRY(lf_table(c, key + ":yhat", yh.data(), yh.size() * sizeof(fp2_t), &dyhat));
  } else {
    lf_table_lookup(c, key + ":binom", &dbinom);
    lf_table_lookup(c, key + ":lead", &dlead);
  }
  const size_t npair = (nrow + 1) / 2;
  void* Zv = nullptr;
  LF_TRY(lf_scratch2(c, npair * P * sizeof(fp2_t), &Zv));
  fp2_t* Z = (fp2_t*)Zv;
  const dim3 gp((u32)((P + 255) / 256), (u32)npair);
  hipLaunchKernelGGL(p256_rs_pre_kernel, gp, dim3(256), 0, c->stream, (u32)n, (u32)P, (u32)nrow, (const elt32_t*)dbinom, (const elt32_t*)d_T, ld, Z);
  LF_TRY(fp2_fft(c, true, npair, (u32)P, Z, W));
  hipLaunchKernelGGL(p256_rs_pointwise_kernel, gp, dim3(256), 0, c->stream, (u32)P, (const fp2_t*)dyhat, Z);
  LF_TRY(fp2_fft(c, false, npair, (u32)P, Z, W));
  const dim3 go((u32)((m - n + 255) / 256), (u32)npair);
  hipLaunchKernelGGL(p256_rs_post_kernel, go, dim3(256), 0, c->stream, (u32)n, (u32)m, (u32)P, (u32)nrow, (const elt32_t*)dlead, (const fp2_t*)Z,
                     (elt32_t*)d_T, ld);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

// Host-buffer form (what GpuReedSolomon<Fp256Base>::interpolate calls from inside the reference's LigeroProver): stage in
// scratch4 -- the encode itself works through scratch2.
extern "C" int lfgpu_fp256_rs_encode_rows_host(lfgpu_ctx* c, size_t nrow, size_t n, size_t m, void* h_T, size_t ld) {
  if (!c || (!h_T && nrow)) return lf_fail(c, LFGPU_ERR_ARG, "fp256_rs_encode_rows_host: null argument");
  if (n == 0 || m < n || ld < m) return lf_fail(c, LFGPU_ERR_ARG, "fp256_rs_encode_rows_host: need 0 < n <= m <= ld");
  if (nrow == 0 || m == n) return LFGPU_OK;
  LF_HIP(c, hipSetDevice(c->device));
  void* d = nullptr;
  const size_t bytes = ((nrow - 1) * ld + m) * sizeof(elt32_t);
  LF_TRY(lf_scratch4(c, bytes, &d));
  LF_HIP(c, hipMemcpyAsync(d, h_T, bytes, hipMemcpyHostToDevice, c->stream));
  LF_TRY(lfgpu_fp256_rs_encode_rows(c, nrow, n, m, d, ld));
  LF_HIP(c, hipMemcpyAsync(h_T, d, bytes, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}

// ------------------------------------------------------------------ column hash, 32-byte elements
// leaf_j = SHA256(nonce_j[32] || canon(T[0][col0+j]) || ... ), canon = 32 little-endian bytes of the value out of
// Montgomery form.  One lane per column; a row read is two coalesced 16 B/lane loads per lane.
__global__ __launch_bounds__(256) void column_leaves32_kernel(u32 nrow, size_t ld, size_t col0, u32 ncols, const elt32_t* __restrict__ T,
                                                              const uint4* __restrict__ nonces, uint4* __restrict__ out, size_t out0) {
  const u32 j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ncols) return;
  const elt32_t* col = T + col0 + j;
  sha_state st;
  sha_init(st);
  const u32 nch = 2 + 2 * nrow;                  // 16-byte chunks of message
  const u32 nblk = (nch * 16 + 9 + 63) / 64;
  const u64 bits = (u64)nch * 128;
  elt32_t cur = e32_zero();
  for (u32 bi = 0; bi < nblk; ++bi) {
    u32 w[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32 ch = 4 * bi + q;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ch < 2) {
        const uint4 nb = nonces[2 * (size_t)j + ch];
        v = make_uint4(bswap32(nb.x), bswap32(nb.y), bswap32(nb.z), bswap32(nb.w));
      } else if (ch < nch) {
        const u32 e = ch - 2;
        if ((e & 1) == 0) cur = fp256_canon(ld32(col + (size_t)(e >> 1) * ld));
        const u64 lo = cur.l[2 * (e & 1)], hi = cur.l[2 * (e & 1) + 1];
        v = make_uint4(bswap32((u32)lo), bswap32((u32)(lo >> 32)), bswap32((u32)hi), bswap32((u32)(hi >> 32)));
      } else if (ch == nch) {
        v.x = 0x80000000u;
      }
      w[4 * q + 0] = v.x;
      w[4 * q + 1] = v.y;
      w[4 * q + 2] = v.z;
      w[4 * q + 3] = v.w;
    }
    if (bi == nblk - 1) {
      w[14] = (u32)(bits >> 32);
      w[15] = (u32)bits;
    }
    sha_compress(st, w);
  }
  out[2 * (out0 + j)] = make_uint4(bswap32(st.h[0]), bswap32(st.h[1]), bswap32(st.h[2]), bswap32(st.h[3]));
  out[2 * (out0 + j) + 1] = make_uint4(bswap32(st.h[4]), bswap32(st.h[5]), bswap32(st.h[6]), bswap32(st.h[7]));
}
int lf_column_leaves32(lfgpu_ctx* c, size_t nrow, size_t ld, size_t col0, size_t ncols, const void* d_T, const void* d_nonces, void* d_out,
                       size_t out0) {
  hipLaunchKernelGGL(column_leaves32_kernel, dim3((u32)((ncols + 255) / 256)), dim3(256), 0, c->stream, (u32)nrow, ld, col0, (u32)ncols,
                     (const elt32_t*)d_T, (const uint4*)d_nonces, (uint4*)d_out, out0);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

// ------------------------------------------------------------------ element-wise ops
__global__ void p256_binop_kernel(int op, size_t n, const elt32_t* __restrict__ a, const elt32_t* __restrict__ b, elt32_t* __restrict__ o) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const elt32_t x = ld32(&a[i]), y = ld32(&b[i]);
  st32(&o[i], op == 0 ? fp256_add(x, y) : op == 1 ? fp256_sub(x, y) : fp256_mul(x, y));
}
int lf_p256_binop(lfgpu_ctx* c, int op, size_t n, const void* d_a, const void* d_b, void* d_out) {
  hipLaunchKernelGGL(p256_binop_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, c->stream, op, n, (const elt32_t*)d_a, (const elt32_t*)d_b,
                     (elt32_t*)d_out);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}
