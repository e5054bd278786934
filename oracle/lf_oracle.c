/*
 * lf_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see lf_oracle.h).
 *
 * Plain-C restatement of the reference's hot-path algorithms.  Each function
 * cites the reference file:line it follows (paths relative to /root/reference).
 * All arithmetic is exact (GF(2) polynomials / integers mod p), so any correct
 * evaluation order is bit-identical to the reference.
 */
#include "lf_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ====================================================================== */
/* GF(2^128) = GF(2)[x]/(x^128 + x^7 + x^2 + x + 1)   lib/gf2k/sysdep.h:23-24 */
/* ====================================================================== */

static inline lfo_elt gf_add(lfo_elt a, lfo_elt b) {
  lfo_elt r = {{a.l[0] ^ b.l[0], a.l[1] ^ b.l[1]}};
  return r;
}

/* reduce a 256-bit carry-less product t[0..3] modulo the field polynomial.
 * Same algebra as gf2_128_reduce applied twice (lib/gf2k/sysdep.h:377-389):
 * x^128 = x^7 + x^2 + x + 1. */
static inline lfo_elt gf_reduce256(const uint64_t t[4]) {
  uint64_t t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3];
  /* fold t3 (weights x^192..) into t1,t2 */
  t1 ^= t3 ^ (t3 << 1) ^ (t3 << 2) ^ (t3 << 7);
  t2 ^= (t3 >> 63) ^ (t3 >> 62) ^ (t3 >> 57);
  /* fold t2 (weights x^128..) into t0,t1 */
  t0 ^= t2 ^ (t2 << 1) ^ (t2 << 2) ^ (t2 << 7);
  t1 ^= (t2 >> 63) ^ (t2 >> 62) ^ (t2 >> 57);
  lfo_elt r = {{t0, t1}};
  return r;
}

/* Bit-serial reference multiply: the slow cross-check the reference's own test
 * uses (refmul, lib/gf2k/gf2_128_test.cc:61-76). */
lfo_elt lfo_gf_mul_bitserial(lfo_elt a, lfo_elt b) {
  uint64_t t[4] = {0, 0, 0, 0};
  for (int i = 0; i < 128; ++i) {
    if ((b.l[i >> 6] >> (i & 63)) & 1) {
      int w = i >> 6, s = i & 63;
      t[w] ^= a.l[0] << s;
      t[w + 1] ^= a.l[1] << s;
      if (s) {
        t[w + 1] ^= a.l[0] >> (64 - s);
        t[w + 2] ^= a.l[1] >> (64 - s);
      }
    }
  }
  return gf_reduce256(t);
}

/* 64x64 -> 128 carry-less product with a 4-bit window (portable). */
static inline void clmul64_window(uint64_t x, uint64_t y, uint64_t* lo, uint64_t* hi) {
  uint64_t tl[16], th[16];
  tl[0] = th[0] = 0;
  tl[1] = y;
  th[1] = 0;
  for (int i = 2; i < 16; i += 2) {
    tl[i] = tl[i >> 1] << 1;
    th[i] = (th[i >> 1] << 1) | (tl[i >> 1] >> 63);
    tl[i + 1] = tl[i] ^ y;
    th[i + 1] = th[i];
  }
  uint64_t rl = 0, rh = 0;
  for (int i = 60; i >= 0; i -= 4) {
    rh = (rh << 4) | (rl >> 60);
    rl <<= 4;
    unsigned n = (unsigned)(x >> i) & 15u;
    rl ^= tl[n];
    rh ^= th[n];
  }
  *lo = rl;
  *hi = rh;
}

static lfo_elt gf_mul_portable(lfo_elt a, lfo_elt b) {
  /* Karatsuba over 64-bit halves, as gf2_128_mul lib/gf2k/sysdep.h:391-400 */
  uint64_t t[4], m0, m1, z0l, z0h, z2l, z2h;
  clmul64_window(a.l[0], b.l[0], &z0l, &z0h);
  clmul64_window(a.l[1], b.l[1], &z2l, &z2h);
  clmul64_window(a.l[0] ^ a.l[1], b.l[0] ^ b.l[1], &m0, &m1);
  m0 ^= z0l ^ z2l;
  m1 ^= z0h ^ z2h;
  t[0] = z0l;
  t[1] = z0h ^ m0;
  t[2] = z2l ^ m1;
  t[3] = z2h;
  return gf_reduce256(t);
}

#if defined(__x86_64__)
#include <immintrin.h>
/* PCLMULQDQ path, same schoolbook as gf2_128_mul lib/gf2k/sysdep.h:57-66 */
__attribute__((target("pclmul,sse2"))) static lfo_elt gf_mul_pclmul(lfo_elt a, lfo_elt b) {
  __m128i x = _mm_set_epi64x((long long)a.l[1], (long long)a.l[0]);
  __m128i y = _mm_set_epi64x((long long)b.l[1], (long long)b.l[0]);
  __m128i t0 = _mm_clmulepi64_si128(x, y, 0x00);
  __m128i t1 = _mm_xor_si128(_mm_clmulepi64_si128(x, y, 0x01), _mm_clmulepi64_si128(x, y, 0x10));
  __m128i t2 = _mm_clmulepi64_si128(x, y, 0x11);
  uint64_t t[4];
  t[0] = (uint64_t)_mm_cvtsi128_si64(t0);
  t[1] = (uint64_t)_mm_cvtsi128_si64(_mm_srli_si128(t0, 8)) ^ (uint64_t)_mm_cvtsi128_si64(t1);
  t[2] = (uint64_t)_mm_cvtsi128_si64(t2) ^ (uint64_t)_mm_cvtsi128_si64(_mm_srli_si128(t1, 8));
  t[3] = (uint64_t)_mm_cvtsi128_si64(_mm_srli_si128(t2, 8));
  return gf_reduce256(t);
}
static int g_have_pclmul = -1;
#endif

lfo_elt lfo_gf_mul(lfo_elt a, lfo_elt b) {
#if defined(__x86_64__)
  if (g_have_pclmul < 0) g_have_pclmul = __builtin_cpu_supports("pclmul") ? 1 : 0;
  if (g_have_pclmul) return gf_mul_pclmul(a, b);
#endif
  return gf_mul_portable(a, b);
}


/* a^(2^128-2) -- any correct inverse equals GF2_128::invertf (lib/gf2k/gf2_128.h:272-309) */
lfo_elt lfo_gf_inv(lfo_elt a) {
  /* a^(2^128 - 2) = prod_{i=1}^{127} a^(2^i) */
  lfo_elt r = {{1, 0}}, s = a;
  for (int i = 1; i < 128; ++i) {
    s = lfo_gf_mul(s, s);
    r = lfo_gf_mul(r, s);
  }
  return r;
}

/* GF2_128 ctor: subfield generator, beta basis  (lib/gf2k/gf2_128.h:97-116, 369-391) */
void lfo_gf_ctx_init(lfo_gf_ctx* c, unsigned k) {
  memset(c, 0, sizeof(*c));
  c->k = k;
  c->sub_bits = 1u << k;
  /* r = x^((2^128-1)/(2^(2^k)-1)) via r <- r^(2^(2^i)+1), i = k..6 */
  lfo_elt r = {{2, 0}};
  for (unsigned i = k; i < 7; ++i) {
    lfo_elt s = r;
    for (unsigned j = 0; j < (1u << i); ++j) s = lfo_gf_mul(s, s);
    r = lfo_gf_mul(r, s);
  }
  c->g = r;
  c->beta[0].l[0] = 1;
  c->beta[0].l[1] = 0;
  for (unsigned i = 1; i < c->sub_bits; ++i) c->beta[i] = lfo_gf_mul(c->beta[i - 1], r);

  /* LCH14 ctor (lib/gf2k/lch14.h:45-77): W_0(X)=X, W_{i+1}(X)=W_i(X)(W_i(X)+W_i(beta_i)),
   * then normalise row i by 1/W_i(beta_i). */
  unsigned sb = c->sub_bits;
  for (unsigned j = 0; j < sb; ++j) c->w_hat[0][j] = c->beta[j];
  for (unsigned i = 0; i + 1 < sb; ++i)
    for (unsigned j = 0; j < sb; ++j)
      c->w_hat[i + 1][j] = lfo_gf_mul(c->w_hat[i][j], gf_add(c->w_hat[i][j], c->w_hat[i][i]));
  for (unsigned i = 0; i < sb; ++i) {
    lfo_elt scale = lfo_gf_inv(c->w_hat[i][i]);
    for (unsigned j = 0; j < sb; ++j) c->w_hat[i][j] = lfo_gf_mul(scale, c->w_hat[i][j]);
  }
}

/* GF2_128::of_scalar (lib/gf2k/gf2_128.h:151-160) */
lfo_elt lfo_gf_of_scalar(const lfo_gf_ctx* c, uint64_t u) {
  lfo_elt t = {{0, 0}};
  for (unsigned k = 0; k < c->sub_bits && u; ++k, u >>= 1)
    if (u & 1) t = gf_add(t, c->beta[k]);
  return t;
}

/* poly_evaluation_points_: 0, 1, g, g^2, ... (lib/gf2k/gf2_128.h:121-127) */
lfo_elt lfo_gf_poly_evaluation_point(const lfo_gf_ctx* c, unsigned i) {
  lfo_elt z = {{0, 0}};
  if (i == 0) return z;
  lfo_elt gi = {{1, 0}};
  for (unsigned j = 1; j < i; ++j) gi = lfo_gf_mul(gi, c->g);
  return gi;
}

/* LCH14::twiddle (lib/gf2k/lch14.h:81-89) */
lfo_elt lfo_lch14_twiddle(const lfo_gf_ctx* c, unsigned i, uint64_t u) {
  lfo_elt t = {{0, 0}};
  for (unsigned k = 0; u != 0; ++k, u >>= 1)
    if (u & 1) t = gf_add(t, c->w_hat[i][k]);
  return t;
}

/* LCH14::twiddles (lib/gf2k/lch14.h:92-100): tw[u] = twiddle(i, coset ^ (u << (i+1))) */
static void lch14_twiddles(const lfo_gf_ctx* c, unsigned i, unsigned l, uint64_t coset, lfo_elt* tw) {
  tw[0] = lfo_lch14_twiddle(c, i, coset);
  for (unsigned k = 0; (i + 1) + k < l; ++k) {
    lfo_elt shift = c->w_hat[i][(i + 1) + k];
    for (size_t u = 0; u < ((size_t)1 << k); ++u) tw[u + ((size_t)1 << k)] = gf_add(tw[u], shift);
  }
}

static inline void bfly_fwd(lfo_elt* B, size_t uv, size_t s, lfo_elt tw) {
  B[uv] = gf_add(B[uv], lfo_gf_mul(tw, B[uv + s])); /* lch14.h:219-223 */
  B[uv + s] = gf_add(B[uv + s], B[uv]);
}
static inline void bfly_bwd(lfo_elt* B, size_t uv, size_t s, lfo_elt tw) {
  B[uv + s] = gf_add(B[uv + s], B[uv]); /* lch14.h:225-229 */
  B[uv] = gf_add(B[uv], lfo_gf_mul(tw, B[uv + s]));
}
static inline void bfly_diag(lfo_elt* B, size_t uv, size_t s, lfo_elt tw) {
  lfo_elt b1 = B[uv + s]; /* lch14.h:232-237 */
  B[uv + s] = gf_add(B[uv + s], B[uv]);
  B[uv] = gf_add(B[uv], lfo_gf_mul(tw, b1));
}

/* LCH14::FFT (lib/gf2k/lch14.h:106-124) */
void lfo_lch14_fft(const lfo_gf_ctx* c, unsigned l, uint64_t coset, lfo_elt* B) {
  if (l == 0) return;
  lfo_elt* tw = (lfo_elt*)malloc(sizeof(lfo_elt) << (l - 1));
  for (unsigned i = l; i-- > 0;) {
    size_t s = (size_t)1 << i;
    lch14_twiddles(c, i, l, coset, tw);
    for (size_t u = 0; (u << (i + 1)) < ((size_t)1 << l); ++u)
      for (size_t v = 0; v < s; ++v) bfly_fwd(B, (u << (i + 1)) + v, s, tw[u]);
  }
  free(tw);
}

/* LCH14::IFFT (lib/gf2k/lch14.h:126-144) */
void lfo_lch14_ifft(const lfo_gf_ctx* c, unsigned l, uint64_t coset, lfo_elt* B) {
  if (l == 0) return;
  lfo_elt* tw = (lfo_elt*)malloc(sizeof(lfo_elt) << (l - 1));
  for (unsigned i = 0; i < l; ++i) {
    size_t s = (size_t)1 << i;
    lch14_twiddles(c, i, l, coset, tw);
    for (size_t u = 0; (u << (i + 1)) < ((size_t)1 << l); ++u)
      for (size_t v = 0; v < s; ++v) bfly_bwd(B, (u << (i + 1)) + v, s, tw[u]);
  }
  free(tw);
}

/* LCH14::bidir_recur (lib/gf2k/lch14.h:185-217) */
static void bidir_recur(const lfo_gf_ctx* c, unsigned i, uint64_t coset, size_t k, lfo_elt* B) {
  if (i-- > 0) {
    size_t s = (size_t)1 << i;
    lfo_elt twu = lfo_lch14_twiddle(c, i, coset);
    if (k < s) {
      for (size_t uv = k; uv < s; ++uv) bfly_fwd(B, uv, s, twu);
      bidir_recur(c, i, coset, k, B);
      for (size_t uv = 0; uv < k; ++uv) bfly_diag(B, uv, s, twu);
      lfo_lch14_fft(c, i, coset + s, B + s);
    } else {
      lfo_lch14_ifft(c, i, coset, B);
      for (size_t uv = k - s; uv < s; ++uv) bfly_diag(B, uv, s, twu);
      bidir_recur(c, i, coset + s, k - s, B + s);
      for (size_t uv = 0; uv < k - s; ++uv) bfly_bwd(B, uv, s, twu);
    }
  }
}

void lfo_lch14_bidirectional_fft(const lfo_gf_ctx* c, unsigned l, uint64_t k, lfo_elt* B) {
  bidir_recur(c, l, 0, (size_t)k, B);
}

/* LCH14ReedSolomon::interpolate (lib/gf2k/lch14_reed_solomon.h:49-103) */
void lfo_lch14_rs_interpolate(const lfo_gf_ctx* c, size_t n, size_t m, lfo_elt* y) {
  unsigned l = 0;
  size_t fftn = 1;
  while (fftn < n) {
    fftn <<= 1;
    ++l;
  }
  lfo_elt* C = (lfo_elt*)calloc(fftn, sizeof(lfo_elt));
  memcpy(C, y, n * sizeof(lfo_elt));
  lfo_lch14_bidirectional_fft(c, l, n, C);
  for (size_t i = n; i < (m < fftn ? m : fftn); ++i) y[i] = C[i];
  for (size_t i = n; i < fftn; ++i) C[i].l[0] = C[i].l[1] = 0;
  for (size_t coset = 1; (coset << l) < m; ++coset) {
    size_t b = coset << l;
    if (b + fftn <= m) {
      memcpy(y + b, C, fftn * sizeof(lfo_elt));
      lfo_lch14_fft(c, l, b, y + b);
    } else {
      lfo_lch14_fft(c, l, b, C);
      for (size_t i = 0; i + b < m; ++i) y[i + b] = C[i];
    }
  }
  free(C);
}

/* ====================================================================== */
/* Fp128, p = 2^128 - 2^108 + 1, Montgomery R = 2^128                     */
/* (lib/algebra/fp_p128.h:61-88, lib/algebra/fp_generic.h:161-201,484-519) */
/* ====================================================================== */
#define P_LO 0x0000000000000001ull
#define P_HI 0xFFFFF00000000000ull

static inline u128 fp_u(lfo_elt a) { return ((u128)a.l[1] << 64) | a.l[0]; }
static inline lfo_elt fp_e(u128 v) {
  lfo_elt r = {{(uint64_t)v, (uint64_t)(v >> 64)}};
  return r;
}
static const u128 FP_P = ((u128)P_HI << 64) | P_LO;

lfo_elt lfo_fp_add(lfo_elt a, lfo_elt b) { /* FpGeneric::add fp_generic.h:161-170 */
  u128 x = fp_u(a), y = fp_u(b), s = x + y;
  int carry = s < x;
  if (carry || s >= FP_P) s -= FP_P;
  return fp_e(s);
}
lfo_elt lfo_fp_sub(lfo_elt a, lfo_elt b) { /* FpGeneric::sub fp_generic.h:176-183 */
  u128 x = fp_u(a), y = fp_u(b), d = x - y;
  if (x < y) d += FP_P;
  return fp_e(d);
}

/* Montgomery product a*b/R mod p.  -p^-1 mod 2^64 = 2^64-1 because p = 1 mod 2^64
 * (the special reduction step of Fp128Reduce, fp_p128.h:68-75, is this REDC with
 * the multiply by p done with shifts). */
lfo_elt lfo_fp_mul(lfo_elt a, lfo_elt b) {
  uint64_t t[5] = {0, 0, 0, 0, 0};
  /* schoolbook 2x2 */
  for (int i = 0; i < 2; ++i) {
    u128 carry = 0;
    for (int j = 0; j < 2; ++j) {
      u128 cur = (u128)a.l[i] * b.l[j] + t[i + j] + carry;
      t[i + j] = (uint64_t)cur;
      carry = cur >> 64;
    }
    u128 cur = (u128)t[i + 2] + carry;
    t[i + 2] = (uint64_t)cur;
    if (i + 3 < 5) t[i + 3] += (uint64_t)(cur >> 64);
  }
  /* REDC, two 64-bit steps */
  for (int i = 0; i < 2; ++i) {
    uint64_t m = (uint64_t)(0 - t[i]); /* t[i] * (-p^-1) */
    u128 cur = (u128)m * P_LO + t[i];
    u128 carry = cur >> 64;
    cur = (u128)m * P_HI + t[i + 1] + carry;
    t[i + 1] = (uint64_t)cur;
    carry = cur >> 64;
    for (int j = i + 2; j < 5 && carry; ++j) {
      cur = (u128)t[j] + carry;
      t[j] = (uint64_t)cur;
      carry = cur >> 64;
    }
  }
  u128 r = ((u128)t[3] << 64) | t[2];
  if (t[4] || r >= FP_P) r -= FP_P;
  return fp_e(r);
}

static lfo_elt fp_rsquare(void) {
  /* R^2 mod p: start from 1 and double 256 times (fp_generic.h:104-107) */
  static int init = 0;
  static lfo_elt rsq;
  if (!init) {
    lfo_elt r = {{1, 0}};
    for (int i = 0; i < 256; ++i) r = lfo_fp_add(r, r);
    rsq = r;
    init = 1;
  }
  return rsq;
}
lfo_elt lfo_fp_to_mont(lfo_elt raw) { return lfo_fp_mul(raw, fp_rsquare()); }
lfo_elt lfo_fp_from_mont(lfo_elt x) {
  lfo_elt one = {{1, 0}};
  return lfo_fp_mul(x, one);
}
lfo_elt lfo_fp_of_scalar(uint64_t u) {
  lfo_elt r = {{u, 0}};
  return lfo_fp_to_mont(r);
}
/* x^(p-2); equals FpGeneric::invertf (fp_generic.h:232-251) for x != 0 */
lfo_elt lfo_fp_inv(lfo_elt x) {
  u128 e = FP_P - 2;
  lfo_elt r = lfo_fp_of_scalar(1), b = x;
  while (e) {
    if (e & 1) r = lfo_fp_mul(r, b);
    b = lfo_fp_mul(b, b);
    e >>= 1;
  }
  return r;
}
/* 164956748514267535023998284330560247862 (lib/algebra/fp_p128.h:48-56) */
lfo_elt lfo_fp_omega32(void) {
  static int init = 0;
  static lfo_elt w;
  if (!init) {
    const char* s = "164956748514267535023998284330560247862";
    u128 v = 0;
    for (; *s; ++s) v = v * 10 + (unsigned)(*s - '0');
    w = lfo_fp_to_mont(fp_e(v));
    init = 1;
  }
  return w;
}

static lfo_elt fp_reroot(lfo_elt w, uint64_t n, uint64_t r) { /* twiddle.h:47-55 */
  while (r < n) {
    w = lfo_fp_mul(w, w);
    r += r;
  }
  return w;
}

/* FFT<Field>::fftb (lib/algebra/fft.h:185-195).  The reference's recursion is a
 * cache-oblivious schedule of the same DFT; the result A[j] = sum_k A[k] w_n^{jk}
 * is unique, so the oracle uses the iterative basecase (fft.h:70-89) at every n:
 * bit-reversal, then log2(n) decimation-in-time stages with w_{2m}^j. */
void lfo_fp_fftb(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j) {
  if (n <= 1) return;
  lfo_elt omega_n = fp_reroot(omega_j, j, n);
  /* bitrev, permutations.h:27-36 */
  unsigned lg = 0;
  while (((size_t)1 << lg) < n) ++lg;
  for (size_t i = 0; i < n; ++i) {
    size_t r = 0;
    for (unsigned b = 0; b < lg; ++b)
      if (i & ((size_t)1 << b)) r |= (size_t)1 << (lg - 1 - b);
    if (i < r) {
      lfo_elt t = A[i];
      A[i] = A[r];
      A[r] = t;
    }
  }
  lfo_elt* w = (lfo_elt*)malloc(sizeof(lfo_elt) * (n / 2));
  w[0] = lfo_fp_of_scalar(1);
  for (size_t i = 1; i < n / 2; ++i) w[i] = lfo_fp_mul(w[i - 1], omega_n);
  for (size_t m = 1; m < n; m <<= 1) {
    size_t ws = n / (2 * m);
    for (size_t k = 0; k < n; k += 2 * m)
      for (size_t jj = 0; jj < m; ++jj) {
        lfo_elt t = jj ? lfo_fp_mul(A[k + jj + m], w[jj * ws]) : A[k + jj + m];
        lfo_elt a0 = A[k + jj];
        A[k + jj] = lfo_fp_add(a0, t);
        A[k + jj + m] = lfo_fp_sub(a0, t);
      }
  }
  free(w);
}
void lfo_fp_fftf(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j) { /* fft.h:198-201 */
  lfo_fp_fftb(A, n, lfo_fp_inv(omega_j), j);
}

/* ------------------------------------------------------------------ F64, F64_2 */
/* Fp<1> over p = 2^64 - 2^32 + 1: one-limb Montgomery arithmetic, R = 2^64 (lib/algebra/fp_generic.h:161-184 add / sub,
 * :484-513 mul0 / mulstep with one reduction_step: a += (a[0] * mprime mod 2^64) * m, drop the low limb, subtract m once). */
#define F64_P 0xFFFFFFFF00000001ull
uint64_t lfo_f64_add(uint64_t a, uint64_t b) {
  u128 s = (u128)a + b;
  return (uint64_t)(s >= F64_P ? s - F64_P : s);
}
uint64_t lfo_f64_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (F64_P - b); }
uint64_t lfo_f64_mul(uint64_t a, uint64_t b) {
  static uint64_t mprime = 0; /* -p^-1 mod 2^64 by Newton iteration */
  if (!mprime) {
    uint64_t inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - F64_P * inv;
    mprime = 0 - inv;
  }
  u128 x = (u128)a * b;
  uint64_t q = (uint64_t)x * mprime;
  u128 qm = (u128)q * F64_P;
  /* (x + q m) / 2^64 without losing the carry out of 128 bits */
  u128 lo = (u128)(uint64_t)x + (uint64_t)qm; /* low limbs: sum is 0 mod 2^64 */
  u128 t = (x >> 64) + (qm >> 64) + (lo >> 64);
  return (uint64_t)(t >= F64_P ? t - F64_P : t);
}
uint64_t lfo_f64_of_scalar(uint64_t u) { /* to_montgomery (fp_generic.h:278-287): u * R^2 * R^-1 */
  static uint64_t r2 = 0;
  if (!r2) {
    u128 r = ((u128)1 << 64) % F64_P;
    r2 = (uint64_t)((r * r) % F64_P);
  }
  return lfo_f64_mul(u, r2);
}
uint64_t lfo_f64_from_mont(uint64_t x) { return lfo_f64_mul(x, 1); }
uint64_t lfo_f64_inv(uint64_t x) { /* x^(p-2) */
  uint64_t r = lfo_f64_of_scalar(1), e = F64_P - 2;
  for (; e; e >>= 1) {
    if (e & 1) r = lfo_f64_mul(r, x);
    x = lfo_f64_mul(x, x);
  }
  return r;
}
uint64_t lfo_f64_omega32(void) { return lfo_f64_of_scalar(2752994695033296049ull); } /* fft_test.cc:212 */
/* Fp2 with i^2 = -1 (lib/algebra/fp2.h:79-123) */
lfo_elt lfo_f64_2_add(lfo_elt a, lfo_elt b) { return (lfo_elt){{lfo_f64_add(a.l[0], b.l[0]), lfo_f64_add(a.l[1], b.l[1])}}; }
lfo_elt lfo_f64_2_sub(lfo_elt a, lfo_elt b) { return (lfo_elt){{lfo_f64_sub(a.l[0], b.l[0]), lfo_f64_sub(a.l[1], b.l[1])}}; }
lfo_elt lfo_f64_2_mul(lfo_elt a, lfo_elt b) {
  uint64_t p0 = lfo_f64_mul(a.l[0], b.l[0]), p1 = lfo_f64_mul(a.l[1], b.l[1]);
  uint64_t x = lfo_f64_mul(lfo_f64_add(a.l[0], a.l[1]), lfo_f64_add(b.l[0], b.l[1]));
  return (lfo_elt){{lfo_f64_sub(p0, p1), lfo_f64_sub(lfo_f64_sub(x, p0), p1)}};
}
lfo_elt lfo_f64_2_inv(lfo_elt a) {
  uint64_t d = lfo_f64_inv(lfo_f64_add(lfo_f64_mul(a.l[0], a.l[0]), lfo_f64_mul(a.l[1], a.l[1])));
  return (lfo_elt){{lfo_f64_mul(a.l[0], d), lfo_f64_mul(lfo_f64_sub(0, a.l[1]), d)}};
}
/* FFT<Fp2<Fp<1>>>::fftb: the same iterative DFT as lfo_fp_fftb above, over F64_2 */
void lfo_f64_2_fftb(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j) {
  if (n <= 1) return;
  lfo_elt omega_n = omega_j;
  for (uint64_t r = n; r < j; r += r) omega_n = lfo_f64_2_mul(omega_n, omega_n); /* twiddle.h:47-55 */
  unsigned lg = 0;
  while (((size_t)1 << lg) < n) ++lg;
  for (size_t i = 0; i < n; ++i) {
    size_t r = 0;
    for (unsigned b = 0; b < lg; ++b)
      if (i & ((size_t)1 << b)) r |= (size_t)1 << (lg - 1 - b);
    if (i < r) {
      lfo_elt t = A[i];
      A[i] = A[r];
      A[r] = t;
    }
  }
  lfo_elt* w = (lfo_elt*)malloc(sizeof(lfo_elt) * (n / 2 ? n / 2 : 1));
  w[0] = (lfo_elt){{lfo_f64_of_scalar(1), 0}};
  for (size_t i = 1; i < n / 2; ++i) w[i] = lfo_f64_2_mul(w[i - 1], omega_n);
  for (size_t m = 1; m < n; m <<= 1) {
    size_t ws = n / (2 * m);
    for (size_t k = 0; k < n; k += 2 * m)
      for (size_t jj = 0; jj < m; ++jj) {
        lfo_elt t = lfo_f64_2_mul(A[k + jj + m], w[jj * ws]);
        lfo_elt a0 = A[k + jj];
        A[k + jj] = lfo_f64_2_add(a0, t);
        A[k + jj + m] = lfo_f64_2_sub(a0, t);
      }
  }
  free(w);
}
void lfo_f64_2_fftf(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j) { lfo_f64_2_fftb(A, n, lfo_f64_2_inv(omega_j), j); }

/* ReedSolomon::interpolate + ctor (lib/algebra/reed_solomon.h:51-110) with
 * FFTConvolution (lib/algebra/convolution.h:56-106). */
void lfo_fp_rs_interpolate(size_t n, size_t m, lfo_elt* y) {
  size_t d = n - 1;
  lfo_elt one = lfo_fp_of_scalar(1);
  lfo_elt* inv = (lfo_elt*)calloc(m, sizeof(lfo_elt)); /* inv[i] = 1/i, inv[0] = 0 */
  for (size_t i = 1; i < m; ++i) inv[i] = lfo_fp_inv(lfo_fp_of_scalar(i));
  lfo_elt* lead = (lfo_elt*)calloc(m - n + 1, sizeof(lfo_elt));
  lfo_elt* binom = (lfo_elt*)calloc(n, sizeof(lfo_elt));
  lead[0] = one;
  binom[0] = one;
  for (size_t i = 1; i + d < m; ++i)
    lead[i] = lfo_fp_mul(lead[i - 1], lfo_fp_mul(lfo_fp_of_scalar(d + i), inv[i]));
  for (size_t k = d; k < m; ++k) {
    lead[k - d] = lfo_fp_mul(lead[k - d], lfo_fp_of_scalar(k - d));
    if (d % 2 == 1) lead[k - d] = lfo_fp_sub(lfo_fp_of_scalar(0), lead[k - d]);
  }
  for (size_t i = 1; i < n; ++i)
    binom[i] = lfo_fp_mul(binom[i - 1], lfo_fp_mul(lfo_fp_of_scalar(n - i), inv[i]));
  for (size_t i = 1; i < n; i += 2) binom[i] = lfo_fp_sub(lfo_fp_of_scalar(0), binom[i]);

  size_t pad = 1;
  while (pad < m) pad <<= 1;
  lfo_elt* yf = (lfo_elt*)calloc(pad, sizeof(lfo_elt));
  lfo_elt* xf = (lfo_elt*)calloc(pad, sizeof(lfo_elt));
  memcpy(yf, inv, m * sizeof(lfo_elt));
  lfo_elt w = lfo_fp_omega32();
  lfo_fp_fftf(yf, pad, w, (uint64_t)1 << 32);
  lfo_elt scale = lfo_fp_inv(lfo_fp_of_scalar(pad));
  for (size_t i = 0; i < pad; ++i) yf[i] = lfo_fp_mul(yf[i], scale);
  for (size_t i = 0; i < n; ++i) xf[i] = lfo_fp_mul(binom[i], y[i]);
  lfo_fp_fftf(xf, pad, w, (uint64_t)1 << 32);
  for (size_t i = 0; i < pad; ++i) xf[i] = lfo_fp_mul(xf[i], yf[i]);
  lfo_fp_fftb(xf, pad, w, (uint64_t)1 << 32);
  for (size_t i = n; i < m; ++i) y[i] = lfo_fp_mul(lead[i - d], xf[i]);
  free(inv);
  free(lead);
  free(binom);
  free(yf);
  free(xf);
}

/* Bogorng (lib/algebra/bogorng.h:43-51) */
void lfo_fp_bogorng_fill(uint64_t seed, size_t n, lfo_elt* out) {
  lfo_elt x = lfo_fp_of_scalar(seed), mul = lfo_fp_of_scalar(7300988u);
  for (size_t i = 0; i < n; ++i) {
    x = lfo_fp_mul(x, mul);
    out[i] = x;
  }
}
/* Bogorng<Fp<1>> (bogorng.h:39-51) lifted to F64_2 as in BM_FFT_F64_2 (fft_test.cc:216-220): real parts from one
 * generator; imag != 0 fills the imaginary parts from a second one seeded seed + 17 */
void lfo_f64_2_bogorng_fill(uint64_t seed, int imag, size_t n, lfo_elt* out) {
  uint64_t re = lfo_f64_of_scalar(seed), im = lfo_f64_of_scalar(seed + 17), mul = lfo_f64_of_scalar(7300988u);
  for (size_t i = 0; i < n; ++i) {
    re = lfo_f64_mul(re, mul);
    im = lfo_f64_mul(im, mul);
    out[i].l[0] = re;
    out[i].l[1] = imag ? im : 0;
  }
}
void lfo_gf_fill(uint64_t seed, size_t n, lfo_elt* out) {
  uint64_t s = seed;
  for (size_t i = 0; i < 2 * n; ++i) {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    out[i >> 1].l[i & 1] = z ^ (z >> 31);
  }
}

/* ====================================================================== */
/* field-generic                                                          */
/* ====================================================================== */
lfo_elt lfo_add(int f, lfo_elt a, lfo_elt b) { return f == LFO_FIELD_GF2_128 ? gf_add(a, b) : lfo_fp_add(a, b); }
lfo_elt lfo_sub(int f, lfo_elt a, lfo_elt b) { return f == LFO_FIELD_GF2_128 ? gf_add(a, b) : lfo_fp_sub(a, b); }
lfo_elt lfo_mul(int f, lfo_elt a, lfo_elt b) { return f == LFO_FIELD_GF2_128 ? lfo_gf_mul(a, b) : lfo_fp_mul(a, b); }
/* to_bytes_field: GF2_128 raw LE (gf2_128.h:178-180); Fp from_montgomery LE (fp_generic.h:378-380) */
void lfo_to_bytes(int f, uint8_t out[16], lfo_elt x) {
  if (f != LFO_FIELD_GF2_128) x = lfo_fp_from_mont(x);
  for (int i = 0; i < 8; ++i) {
    out[i] = (uint8_t)(x.l[0] >> (8 * i));
    out[8 + i] = (uint8_t)(x.l[1] >> (8 * i));
  }
}

/* ====================================================================== */
/* SHA-256 (FIPS 180-4; the reference calls OpenSSL, lib/util/crypto.h:41-70) */
/* ====================================================================== */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t* p) {
  uint32_t w[64];
  for (int i = 0; i < 16; ++i)
    w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
  for (int i = 16; i < 64; ++i) {
    uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
    uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
    w[i] = w[i - 16] + s0 + w[i - 7] + s1;
  }
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
  for (int i = 0; i < 64; ++i) {
    uint32_t S1 = ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25);
    uint32_t ch = (e & f) ^ (~e & g);
    uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
    uint32_t S0 = ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22);
    uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
    uint32_t t2 = S0 + mj;
    hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void lfo_sha256_init(lfo_sha256* s) {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(s->h, iv, sizeof(iv));
  s->len = 0;
}
void lfo_sha256_update(lfo_sha256* s, const uint8_t* p, size_t n) {
  size_t fill = (size_t)(s->len & 63);
  s->len += n;
  if (fill) {
    size_t take = 64 - fill < n ? 64 - fill : n;
    memcpy(s->buf + fill, p, take);
    p += take;
    n -= take;
    if (fill + take < 64) return;
    sha256_block(s->h, s->buf);
  }
  for (; n >= 64; p += 64, n -= 64) sha256_block(s->h, p);
  if (n) memcpy(s->buf, p, n);
}
void lfo_sha256_final(lfo_sha256* s, uint8_t out[32]) {
  uint64_t bits = s->len * 8;
  uint8_t pad[72];
  size_t fill = (size_t)(s->len & 63);
  size_t padlen = (fill < 56 ? 56 - fill : 120 - fill);
  memset(pad, 0, sizeof(pad));
  pad[0] = 0x80;
  for (int i = 0; i < 8; ++i) pad[padlen + i] = (uint8_t)(bits >> (56 - 8 * i));
  lfo_sha256_update(s, pad, padlen + 8);
  for (int i = 0; i < 8; ++i) {
    out[4 * i] = (uint8_t)(s->h[i] >> 24);
    out[4 * i + 1] = (uint8_t)(s->h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(s->h[i] >> 8);
    out[4 * i + 3] = (uint8_t)(s->h[i]);
  }
}

/* MerkleTree::build_tree (lib/merkle/merkle_tree.h:109-114), Digest::hash2 :51-58 */
void lfo_merkle_build_tree(size_t n, const uint8_t* leaves, uint8_t* layers) {
  memset(layers, 0, 32 * n);
  memcpy(layers + 32 * n, leaves, 32 * n);
  for (size_t i = n; i-- > 1;) {
    lfo_sha256 s;
    lfo_sha256_init(&s);
    lfo_sha256_update(&s, layers + 32 * (2 * i), 32);
    lfo_sha256_update(&s, layers + 32 * (2 * i + 1), 32);
    lfo_sha256_final(&s, layers + 32 * i);
  }
}

/* MerkleCommitment::commit leaf loop (lib/merkle/merkle_commitment.h:52-61) with
 * LigeroCommon::column_hash (lib/ligero/ligero_param.h:432-439) */
void lfo_column_leaves(int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                       const lfo_elt* T, const uint8_t* nonces, uint8_t* leaves) {
  for (size_t j = 0; j < ncols; ++j) {
    lfo_sha256 s;
    lfo_sha256_init(&s);
    lfo_sha256_update(&s, nonces + 32 * j, 32);
    for (size_t i = 0; i < nrow; ++i) {
      uint8_t buf[16];
      lfo_to_bytes(field, buf, T[i * ld + col0 + j]);
      lfo_sha256_update(&s, buf, 16);
    }
    lfo_sha256_final(&s, leaves + 32 * j);
  }
}

void lfo_column_commit(int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                       const lfo_elt* T, const uint8_t* nonces, uint8_t root_out[32], uint8_t* layers) {
  uint8_t* leaves = (uint8_t*)malloc(32 * ncols);
  uint8_t* lay = layers ? layers : (uint8_t*)malloc(64 * ncols);
  lfo_column_leaves(field, nrow, ld, col0, ncols, T, nonces, leaves);
  lfo_merkle_build_tree(ncols, leaves, lay);
  memcpy(root_out, lay + 32, 32);
  free(leaves);
  if (!layers) free(lay);
}

/* ====================================================================== */
/* sumcheck round pieces                                                  */
/* ====================================================================== */
/* loop body of ProverLayers::evaluations (lib/sumcheck/prover_layers.h:365-388) */
void lfo_sumcheck_partials(int f, size_t n, const lfo_elt* QW, const lfo_elt* W, lfo_elt* a0, lfo_elt* a2) {
  lfo_elt s0 = {{0, 0}}, s2 = {{0, 0}};
  size_t nodd = n / 2;
  for (size_t i = 0; i < nodd; ++i) {
    s0 = lfo_add(f, s0, lfo_mul(f, QW[2 * i], W[2 * i]));
    lfo_elt dqw = lfo_sub(f, QW[2 * i + 1], QW[2 * i]);
    lfo_elt dw = lfo_sub(f, W[2 * i + 1], W[2 * i]);
    s2 = lfo_add(f, s2, lfo_mul(f, dqw, dw));
  }
  if (2 * nodd < n) {
    lfo_elt t = lfo_mul(f, QW[2 * nodd], W[2 * nodd]);
    s0 = lfo_add(f, s0, t);
    s2 = lfo_add(f, s2, t);
  }
  *a0 = s0;
  *a2 = s2;
}

/* ProverLayers::evaluations tail (prover_layers.h:390-402): coef[1] from sum,
 * then Horner at poly_evaluation_point(0..2) (Poly::eval_monomial poly.h:100-108) */
void lfo_sumcheck_evaluations(int f, const lfo_gf_ctx* c, size_t n, lfo_elt eq0, const lfo_elt* QW,
                              const lfo_elt* W, lfo_elt sum, lfo_elt evals[3]) {
  lfo_elt a0, a2, coef[3];
  lfo_sumcheck_partials(f, n, QW, W, &a0, &a2);
  coef[0] = lfo_mul(f, eq0, a0);
  coef[2] = lfo_mul(f, eq0, a2);
  coef[1] = lfo_sub(f, lfo_sub(f, lfo_sub(f, sum, coef[0]), coef[0]), coef[2]);
  for (unsigned k = 0; k < 3; ++k) {
    lfo_elt x = f == LFO_FIELD_GF2_128 ? lfo_gf_poly_evaluation_point(c, k) : lfo_fp_of_scalar(k);
    lfo_elt e = coef[2];
    e = lfo_add(f, lfo_mul(f, e, x), coef[1]);
    e = lfo_add(f, lfo_mul(f, e, x), coef[0]);
    evals[k] = e;
  }
}

/* Dense::bind (lib/arrays/dense.h:70-87), n1 = 1 */
size_t lfo_dense_bind(int f, size_t n0, lfo_elt r, const lfo_elt* in, lfo_elt* out) {
  size_t i0 = 0, rd = 0, wr = 0;
  while (2 * i0 + 1 < n0) {
    lfo_elt f0 = in[rd], f1 = in[rd + 1];
    out[wr] = lfo_add(f, f0, lfo_mul(f, lfo_sub(f, f1, f0), r)); /* affine_interpolation affine.h:26-34 */
    ++i0, rd += 2, ++wr;
  }
  if (2 * i0 < n0) {
    lfo_elt f0 = in[rd];
    out[wr] = lfo_sub(f, f0, lfo_mul(f, f0, r)); /* affine_interpolation_nz_z affine.h:47-52 */
    ++wr;
  }
  return (n0 + 1) / 2;
}

/* HQuad::bind_h (lib/sumcheck/hquad.h:90-123) */
size_t lfo_hquad_bind_h(int f, size_t n, uint32_t* hc, lfo_elt* vc, lfo_elt r, int hand) {
  size_t rd = 0, wr = 0;
  int o = 1 - hand;
  while (rd < n) {
    uint32_t hh = hc[2 * rd + hand] >> 1, ho = hc[2 * rd + o];
    lfo_elt v;
    size_t rd1 = rd + 1;
    if (rd1 < n && hc[2 * rd + o] == hc[2 * rd1 + o] && (hc[2 * rd + hand] >> 1) == (hc[2 * rd1 + hand] >> 1) &&
        hc[2 * rd1 + hand] == hc[2 * rd + hand] + 1) {
      v = lfo_add(f, vc[rd], lfo_mul(f, lfo_sub(f, vc[rd1], vc[rd]), r));
      rd += 2;
    } else {
      if ((hc[2 * rd + hand] & 1) == 0)
        v = lfo_sub(f, vc[rd], lfo_mul(f, vc[rd], r));
      else
        v = lfo_mul(f, vc[rd], r); /* affine_interpolation_z_nz affine.h:37-44 */
      rd = rd1;
    }
    hc[2 * wr + hand] = hh;
    hc[2 * wr + o] = ho;
    vc[wr] = v;
    ++wr;
  }
  return wr;
}

/* QW loop in ProverLayers::layer (lib/sumcheck/prover_layers.h:239-243) */
void lfo_qw_scatter(int f, size_t n, const uint32_t* hc, const lfo_elt* vc, int hand, const lfo_elt* Wother,
                    size_t nqw, lfo_elt* QW) {
  memset(QW, 0, nqw * sizeof(lfo_elt));
  for (size_t i = 0; i < n; ++i) {
    uint32_t p0 = hc[2 * i + hand], p1 = hc[2 * i + 1 - hand];
    QW[p0] = lfo_add(f, QW[p0], lfo_mul(f, vc[i], Wother[p1]));
  }
}

/* Eqs::raw_eq2 / fill_recursive (lib/arrays/eqs.h:46-80) */
static void eq2_fill(int f, lfo_elt* eq, size_t l, size_t n, const lfo_elt* G0, const lfo_elt* G1, lfo_elt w0, lfo_elt w1) {
  if (l > 0) {
    size_t nl = l - 1, s = (size_t)1 << nl;
    lfo_elt w0hi = lfo_mul(f, w0, G0[nl]), w1hi = lfo_mul(f, w1, G1[nl]);
    lfo_elt w0lo = lfo_sub(f, w0, w0hi), w1lo = lfo_sub(f, w1, w1hi);
    if (n <= s) {
      eq2_fill(f, eq, nl, n, G0, G1, w0lo, w1lo);
    } else {
      eq2_fill(f, eq, nl, s, G0, G1, w0lo, w1lo);
      eq2_fill(f, eq + s, nl, n - s, G0, G1, w0hi, w1hi);
    }
  } else {
    eq[0] = lfo_add(f, w0, w1);
  }
}
void lfo_raw_eq2(int f, size_t logn, size_t n, const lfo_elt* G0, const lfo_elt* G1, lfo_elt alpha, lfo_elt* eq) {
  lfo_elt one = f == LFO_FIELD_GF2_128 ? (lfo_elt){{1, 0}} : lfo_fp_of_scalar(1);
  eq2_fill(f, eq, logn, n, G0, G1, one, alpha);
}

static int elt_is_zero(lfo_elt a) { return (a.l[0] | a.l[1]) == 0; }

/* ProverLayers::eval_quad (lib/sumcheck/prover_layers.h:278-305), n0 = 1; r = h[0], l = h[1] */
int lfo_eval_quad(int f, size_t nterms, const uint32_t* g, const uint32_t* h0, const uint32_t* h1, const uint32_t* vi,
                  const lfo_elt* kvec, size_t nv, const lfo_elt* W, lfo_elt* V) {
  memset(V, 0, nv * sizeof(lfo_elt));
  for (size_t i = 0; i < nterms; ++i) {
    lfo_elt v = kvec[vi[i]];
    if (elt_is_zero(v)) {
      lfo_elt y = lfo_mul(f, W[h1[i]], W[h0[i]]);
      if (!elt_is_zero(y)) return 0;
    } else {
      lfo_elt x = lfo_mul(f, lfo_mul(f, v, W[h1[i]]), W[h0[i]]);
      V[g[i]] = lfo_add(f, V[g[i]], x);
    }
  }
  return 1;
}

/* Quad::bind_g (lib/sumcheck/quad.h:152-185) with prep_v (:213-220) */
size_t lfo_quad_bind_g(int f, size_t nterms, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                       const uint32_t* vi, const lfo_elt* kvec, size_t logv, const lfo_elt* G0, const lfo_elt* G1,
                       lfo_elt alpha, lfo_elt beta, uint32_t* hc_out, lfo_elt* vc_out) {
  size_t nv = (size_t)1 << logv;
  lfo_elt* dot = (lfo_elt*)malloc(nv * sizeof(lfo_elt));
  lfo_raw_eq2(f, logv, nv, G0, G1, alpha, dot);
  size_t wr = 0;
  for (size_t i = 0; i < nterms; ++i) {
    lfo_elt v = kvec[vi[i]];
    lfo_elt pv = lfo_mul(f, elt_is_zero(v) ? beta : v, dot[g[i]]);
    if (wr > 0 && hc_out[2 * (wr - 1)] == h0[i] && hc_out[2 * (wr - 1) + 1] == h1[i]) {
      vc_out[wr - 1] = lfo_add(f, vc_out[wr - 1], pv);
    } else {
      hc_out[2 * wr] = h0[i];
      hc_out[2 * wr + 1] = h1[i];
      vc_out[wr] = pv;
      ++wr;
    }
  }
  free(dot);
  return wr;
}

/* Quad::bind_gh_all (lib/sumcheck/quad.h:188-210) */
lfo_elt lfo_quad_bind_gh_all(int f, size_t nterms, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                             const uint32_t* vi, const lfo_elt* kvec, size_t logv, size_t nv, const lfo_elt* G0,
                             const lfo_elt* G1, lfo_elt alpha, lfo_elt beta, size_t logw, size_t nw, const lfo_elt* H0,
                             const lfo_elt* H1) {
  lfo_elt zero = {{0, 0}};
  lfo_elt* eqg = (lfo_elt*)malloc((nv ? nv : 1) * sizeof(lfo_elt));
  lfo_elt* eqh0 = (lfo_elt*)malloc((nw ? nw : 1) * sizeof(lfo_elt));
  lfo_elt* eqh1 = (lfo_elt*)malloc((nw ? nw : 1) * sizeof(lfo_elt));
  lfo_raw_eq2(f, logv, nv, G0, G1, alpha, eqg);
  lfo_raw_eq2(f, logw, nw, H0, H0, zero, eqh0); /* Eqs(logw, nw, H0): EQ(H0, i) */
  lfo_raw_eq2(f, logw, nw, H1, H1, zero, eqh1);
  lfo_elt s = zero;
  for (size_t i = 0; i < nterms; ++i) {
    lfo_elt v = kvec[vi[i]];
    lfo_elt q = lfo_mul(f, elt_is_zero(v) ? beta : v, eqg[g[i]]);
    q = lfo_mul(f, q, eqh0[h0[i]]);
    s = lfo_add(f, s, lfo_mul(f, q, eqh1[h1[i]]));
  }
  free(eqg);
  free(eqh0);
  free(eqh1);
  return s;
}

/* Blas::axpy / vaxpy (lib/algebra/blas.h:62-78) */
void lfo_axpy(int f, size_t n, lfo_elt* y, lfo_elt a, const lfo_elt* x) {
  for (size_t i = 0; i < n; ++i) y[i] = lfo_add(f, y[i], lfo_mul(f, x[i], a));
}
void lfo_vaxpy(int f, size_t n, lfo_elt* y, const lfo_elt* a, const lfo_elt* x) {
  for (size_t i = 0; i < n; ++i) y[i] = lfo_add(f, y[i], lfo_mul(f, x[i], a[i]));
}
