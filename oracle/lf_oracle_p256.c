/* lf_oracle_p256.c -- CPU restatement of the reference's P-256 base-field path (TEST INFRASTRUCTURE, see lf_oracle.h):
 * Fp256Base arithmetic, Fp2, the real FFT over Fp2 (half-complex format), the FFTExt convolution, Reed-Solomon
 * interpolation and the column hash for 32-byte elements.  Each function cites the reference lines it follows.
 * Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product. */
#include <stdlib.h>
#include <string.h>

#include "lf_oracle.h"

typedef unsigned __int128 u128;

/* p = 2^256 - 2^224 + 2^192 + 2^96 - 1 (lib/algebra/fp_p256.h:34-39), little-endian u64 limbs */
static const uint64_t P[4] = {0xFFFFFFFFFFFFFFFFull, 0x00000000FFFFFFFFull, 0, 0xFFFFFFFF00000001ull};

static int geq_p(const uint64_t a[4]) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] > P[i]) return 1;
    if (a[i] < P[i]) return 0;
  }
  return 1;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  u128 c = 0;
  for (int i = 0; i < 4; ++i) {
    c += (u128)a[i] + b[i];
    r[i] = (uint64_t)c;
    c >>= 64;
  }
  return (uint64_t)c;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
  uint64_t br = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - b[i] - br;
    r[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  return br;
}
/* FpGeneric::add / sub (lib/algebra/fp_generic.h:161-201) */
lfo_e32 lfo_p256_add(lfo_e32 a, lfo_e32 b) {
  lfo_e32 r;
  uint64_t c = add4(r.l, a.l, b.l);
  if (c || geq_p(r.l)) sub4(r.l, r.l, P);
  return r;
}
lfo_e32 lfo_p256_sub(lfo_e32 a, lfo_e32 b) {
  lfo_e32 r;
  if (sub4(r.l, a.l, b.l)) add4(r.l, r.l, P);
  return r;
}
/* Montgomery product, R = 2^256 (fp_generic.h:484-519 with Fp256Reduce: -p^-1 mod 2^64 = 1), generic CIOS on
 * 64-bit limbs with a full m * p multiply -- deliberately NOT the shift-only reduction the device uses */
lfo_e32 lfo_p256_mul(lfo_e32 a, lfo_e32 b) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a.l[i] * b.l[j] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0]; /* * mprime (= 1) */
    c = (u128)m * P[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * P[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
    t[5] = 0;
  }
  lfo_e32 r = {{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_p(r.l)) sub4(r.l, r.l, P);
  return r;
}
static lfo_e32 rsq(void) { /* R^2 mod p by 512 doublings of 1 */
  static int init = 0;
  static lfo_e32 v;
  if (!init) {
    lfo_e32 x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; ++i) x = lfo_p256_add(x, x);
    v = x;
    init = 1;
  }
  return v;
}
lfo_e32 lfo_p256_to_mont(lfo_e32 raw) { return lfo_p256_mul(raw, rsq()); }
lfo_e32 lfo_p256_from_mont(lfo_e32 x) {
  lfo_e32 one = {{1, 0, 0, 0}};
  return lfo_p256_mul(x, one);
}
lfo_e32 lfo_p256_of_scalar(uint64_t u) {
  lfo_e32 raw = {{u, 0, 0, 0}};
  return lfo_p256_to_mont(raw);
}
lfo_e32 lfo_p256_inv(lfo_e32 x) { /* x^(p-2) */
  uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
  lfo_e32 r = lfo_p256_of_scalar(1), b = x;
  for (int i = 0; i < 256; ++i) {
    if ((e[i / 64] >> (i % 64)) & 1) r = lfo_p256_mul(r, b);
    b = lfo_p256_mul(b, b);
  }
  return r;
}
/* to_bytes_field: canonical value, little-endian (fp_generic.h:329-333) */
void lfo_p256_to_bytes(uint8_t out[32], lfo_e32 x) {
  lfo_e32 c = lfo_p256_from_mont(x);
  memcpy(out, c.l, 32);
}
/* deterministic test data: canonical values from a 64-bit LCG, stored as Montgomery images (any value < p is one) */
void lfo_p256_fill(uint64_t seed, size_t n, lfo_e32* out) {
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  for (size_t i = 0; i < n; ++i) {
    for (int k = 0; k < 4; ++k) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      out[i].l[k] = s ^ (s >> 29);
    }
    out[i].l[3] &= 0x7FFFFFFFFFFFFFFFull; /* < 2^255 < p */
  }
}

/* ---- Fp2<Fp256Base> (lib/algebra/fp2.h:77-95) */
typedef struct { lfo_e32 re, im; } c32;
static c32 cmul2(c32 a, c32 y) {
  lfo_e32 p0 = lfo_p256_mul(a.re, y.re), p1 = lfo_p256_mul(a.im, y.im);
  lfo_e32 a01 = lfo_p256_add(a.re, a.im), y01 = lfo_p256_add(y.re, y.im);
  c32 r;
  r.re = lfo_p256_sub(p0, p1);
  r.im = lfo_p256_sub(lfo_p256_sub(lfo_p256_mul(a01, y01), p0), p1);
  return r;
}
static lfo_e32 dec(const char* s) { /* decimal string -> Montgomery */
  lfo_e32 r = {{0, 0, 0, 0}};
  lfo_e32 ten = lfo_p256_of_scalar(10);
  for (; *s; ++s) r = lfo_p256_add(lfo_p256_mul(r, ten), lfo_p256_of_scalar((uint64_t)(*s - '0')));
  return r;
}
/* the root of unity of order 2^31 of Fp2<Fp256Base> (lib/circuits/mdoc/mdoc_zk.cc:82-88) */
void lfo_p256_omega(lfo_e32* re, lfo_e32* im) {
  *re = dec("112649224146410281873500457609690258373018840430489408729223714171582664680802");
  *im = dec("84087994358540907695740461427818660560182168997182378749313018254450460212908");
}

/* ---- RFFT<Fp2<Fp256Base>> (lib/algebra/rfft.h) */
static void cmul_(lfo_e32* xr, lfo_e32* xi, lfo_e32 br, lfo_e32 bi) { /* X *= B (rfft.h:380-392) */
  lfo_e32 p0 = lfo_p256_mul(*xr, br), p1 = lfo_p256_mul(*xi, bi);
  lfo_e32 a01 = lfo_p256_add(*xr, *xi), b01 = lfo_p256_add(br, bi);
  *xr = lfo_p256_sub(p0, p1);
  a01 = lfo_p256_mul(a01, b01);
  a01 = lfo_p256_sub(a01, p0);
  *xi = lfo_p256_sub(a01, p1);
}
static void cmulj_(lfo_e32* xr, lfo_e32* xi, lfo_e32 br, lfo_e32 bi) { /* X *= conj(B) (rfft.h:394-408) */
  lfo_e32 p0 = lfo_p256_mul(*xr, br), p1 = lfo_p256_mul(*xi, bi);
  lfo_e32 a01 = lfo_p256_add(*xr, *xi), b01 = lfo_p256_sub(br, bi);
  *xr = lfo_p256_add(p0, p1);
  a01 = lfo_p256_mul(a01, b01);
  a01 = lfo_p256_sub(a01, p0);
  *xi = lfo_p256_add(a01, p1);
}
#define ADD lfo_p256_add
#define SUB lfo_p256_sub
#define MUL lfo_p256_mul
static void r2hcI_2(lfo_e32* A, size_t s) { /* rfft.h:144-149 */
  lfo_e32 t = A[s];
  A[s] = SUB(A[0], t);
  A[0] = ADD(A[0], t);
}
static void r2hcI_4(lfo_e32* A, size_t s) { /* rfft.h:151-162 */
  lfo_e32 x0 = A[0], x1 = A[s], z0 = ADD(x0, x1), x2 = A[2 * s], x3 = A[3 * s], z1 = ADD(x2, x3);
  A[0] = ADD(z0, z1);
  A[2 * s] = SUB(z0, z1);
  A[s] = SUB(x0, x1);
  A[3 * s] = SUB(x3, x2);
}
static void r2hcII_4(lfo_e32* A, size_t s, c32 w8) { /* rfft.h:165-179 */
  lfo_e32 x2 = A[2 * s], x3 = A[3 * s];
  lfo_e32 z0 = MUL(ADD(x2, x3), w8.im), z1 = MUL(SUB(x2, x3), w8.re);
  lfo_e32 x0 = A[0], x1 = A[s], zero = {{0, 0, 0, 0}};
  A[0] = ADD(x0, z1);
  A[s] = SUB(x0, z1);
  A[2 * s] = SUB(x1, z0);
  A[3 * s] = SUB(zero, ADD(x1, z0));
}
static void hc2hcf_4(lfo_e32* Ar, lfo_e32* Ai, size_t s, c32 tw1, c32 tw2, c32 tw3) { /* rfft.h:181-205 */
  cmulj_(&Ar[s], &Ai[s], tw2.re, tw2.im);
  lfo_e32 y0r = ADD(Ar[0], Ar[s]), y0i = ADD(Ai[0], Ai[s]), y1r = SUB(Ar[0], Ar[s]), y1i = SUB(Ai[0], Ai[s]);
  cmulj_(&Ar[2 * s], &Ai[2 * s], tw1.re, tw1.im);
  cmulj_(&Ar[3 * s], &Ai[3 * s], tw3.re, tw3.im);
  lfo_e32 y2r = ADD(Ar[3 * s], Ar[2 * s]), y3r = SUB(Ar[3 * s], Ar[2 * s]);
  lfo_e32 y2i = ADD(Ai[2 * s], Ai[3 * s]), y3i = SUB(Ai[2 * s], Ai[3 * s]);
  Ar[0] = ADD(y0r, y2r);
  Ai[s] = SUB(y0r, y2r);
  Ar[s] = ADD(y1r, y3i);
  Ai[0] = SUB(y1r, y3i);
  Ai[3 * s] = ADD(y2i, y0i);
  Ar[2 * s] = SUB(y2i, y0i);
  Ai[2 * s] = ADD(y3r, y1i);
  Ar[3 * s] = SUB(y3r, y1i);
}
static void hc2rI_4(lfo_e32* A, size_t s) { /* rfft.h:229-238 */
  lfo_e32 y0 = ADD(A[0], A[2 * s]), y1 = SUB(A[0], A[2 * s]), y2 = ADD(A[s], A[s]), y3 = ADD(A[3 * s], A[3 * s]);
  A[0] = ADD(y0, y2);
  A[s] = SUB(y0, y2);
  A[2 * s] = SUB(y1, y3);
  A[3 * s] = ADD(y1, y3);
}
static void hc2rIII_4(lfo_e32* A, size_t s, c32 w8) { /* rfft.h:240-254 */
  lfo_e32 x0 = ADD(A[0], A[0]), x1 = ADD(A[s], A[s]), x2 = ADD(A[2 * s], A[2 * s]), x3 = ADD(A[3 * s], A[3 * s]);
  lfo_e32 zero = {{0, 0, 0, 0}};
  A[0] = ADD(x0, x1);
  A[s] = SUB(x2, x3);
  lfo_e32 z0 = MUL(SUB(x0, x1), w8.re), z1 = MUL(ADD(x3, x2), w8.im);
  A[2 * s] = SUB(z0, z1);
  A[3 * s] = SUB(zero, ADD(z0, z1));
}
static void hc2hcb_4(lfo_e32* Ar, lfo_e32* Ai, size_t s, c32 tw1, c32 tw2, c32 tw3) { /* rfft.h:256-278 */
  lfo_e32 z0 = ADD(Ar[0], Ai[s]), z1 = SUB(Ar[0], Ai[s]), z2 = ADD(Ar[s], Ai[0]), z3 = SUB(Ar[s], Ai[0]);
  lfo_e32 z4 = ADD(Ai[3 * s], Ar[2 * s]), z5 = SUB(Ai[3 * s], Ar[2 * s]), z6 = ADD(Ai[2 * s], Ar[3 * s]), z7 = SUB(Ai[2 * s], Ar[3 * s]);
  Ar[0] = ADD(z0, z2);
  Ai[0] = ADD(z5, z7);
  Ar[s] = SUB(z0, z2);
  Ai[s] = SUB(z5, z7);
  cmul_(&Ar[s], &Ai[s], tw2.re, tw2.im);
  Ar[2 * s] = SUB(z1, z6);
  Ai[2 * s] = ADD(z4, z3);
  cmul_(&Ar[2 * s], &Ai[2 * s], tw1.re, tw1.im);
  Ar[3 * s] = ADD(z1, z6);
  Ai[3 * s] = SUB(z4, z3);
  cmul_(&Ar[3 * s], &Ai[3 * s], tw3.re, tw3.im);
}
static void bitrev(lfo_e32* A, size_t n) { /* Permutations::bitrev (lib/algebra/permutations.h:27-36,93-98) */
  size_t revi = 0;
  for (size_t i = 0; i + 1 < n; ++i) {
    if (i < revi) {
      lfo_e32 t = A[i];
      A[i] = A[revi];
      A[revi] = t;
    }
    size_t bit = n;
    do {
      bit >>= 1;
      revi ^= bit;
    } while (!(revi & bit));
  }
}
/* Twiddle::reroot + the table w_[i] = omega_n^i, i < n/2 (lib/algebra/twiddle.h:36-55) */
static c32* roots_table(size_t n) {
  c32 w;
  lfo_p256_omega(&w.re, &w.im);
  for (uint64_t r = n; r < ((uint64_t)1 << 31); r += r) w = cmul2(w, w);
  c32* t = (c32*)malloc(sizeof(c32) * (n / 2 ? n / 2 : 1));
  c32 x = {lfo_p256_of_scalar(1), {{0, 0, 0, 0}}};
  for (size_t i = 0; 2 * i < n; ++i) {
    t[i] = x;
    x = cmul2(x, w);
  }
  return t;
}
/* RFFT::r2hc (rfft.h:282-329) */
void lfo_p256_r2hc(lfo_e32* A, size_t n) {
  if (n == 2) {
    r2hcI_2(A, 1);
  } else if (n >= 4) {
    c32* w = roots_table(n);
    bitrev(A, n);
    size_t m = n;
    while (m > 4) m /= 4;
    if (m == 2) {
      for (size_t k = 0; k < n; k += 2) r2hcI_2(&A[k], 1);
    } else {
      for (size_t k = 0; k < n; k += 4) r2hcI_4(&A[k], 1);
    }
    for (; m < n; m = 4 * m) {
      size_t ws = n / (4 * m);
      for (size_t k = 0; k < n; k += 4 * m) {
        size_t j;
        r2hcI_4(&A[k], m);
        for (j = 1; j + j < m; ++j) hc2hcf_4(&A[k + j], &A[k + m - j], m, w[j * ws], w[2 * j * ws], w[3 * j * ws]);
        r2hcII_4(&A[k + j], m, w[j * ws]);
      }
    }
    free(w);
  }
}
/* RFFT::hc2r (rfft.h:332-376) */
void lfo_p256_hc2r(lfo_e32* A, size_t n) {
  if (n == 2) {
    r2hcI_2(A, 1); /* hc2rI_2 is the same butterfly (rfft.h:222-227) */
  } else if (n >= 4) {
    c32* w = roots_table(n);
    size_t m = n;
    while (m > 4) {
      m /= 4;
      size_t ws = n / (4 * m);
      for (size_t k = 0; k < n; k += 4 * m) {
        size_t j;
        hc2rI_4(&A[k], m);
        for (j = 1; j + j < m; ++j) hc2hcb_4(&A[k + j], &A[k + m - j], m, w[j * ws], w[2 * j * ws], w[3 * j * ws]);
        hc2rIII_4(&A[k + j], m, w[j * ws]);
      }
    }
    if (m == 2) {
      for (size_t k = 0; k < n; k += 2) r2hcI_2(&A[k], 1);
    } else {
      for (size_t k = 0; k < n; k += 4) hc2rI_4(&A[k], 1);
    }
    bitrev(A, n);
    free(w);
  }
}

/* ReedSolomon<Fp256Base, FFTExtConvolutionFactory>::interpolate (lib/algebra/reed_solomon.h:51-110) with
 * FFTExtConvolution (lib/algebra/convolution.h:129-191): y[0..n) given -> y[n..m) */
void lfo_p256_rs_interpolate(size_t n, size_t m, lfo_e32* y) {
  if (m <= n) return;
  size_t d = n - 1, P2 = 1;
  while (P2 < m) P2 *= 2;
  lfo_e32 zero = {{0, 0, 0, 0}}, one = lfo_p256_of_scalar(1);
  lfo_e32* inv = (lfo_e32*)malloc(32 * m);
  { /* AlgebraUtil::batch_inverse_arithmetic (lib/algebra/utility.h:51-72) */
    lfo_e32 p = one, bi = zero;
    inv[0] = zero;
    for (size_t i = 1; i < m; ++i) {
      bi = ADD(bi, one);
      inv[i] = p;
      p = MUL(p, bi);
    }
    p = lfo_p256_inv(p);
    for (size_t i = m; i-- > 0;) {
      inv[i] = MUL(inv[i], p);
      p = MUL(p, bi);
      bi = SUB(bi, one);
    }
  }
  lfo_e32* lead = (lfo_e32*)malloc(32 * (m - n + 1));
  lfo_e32* binom = (lfo_e32*)malloc(32 * n);
  lead[0] = one;
  binom[0] = one;
  for (size_t i = 1; i + d < m; ++i) lead[i] = MUL(lead[i - 1], MUL(lfo_p256_of_scalar(d + i), inv[i]));
  for (size_t k = d; k < m; ++k) {
    lead[k - d] = MUL(lead[k - d], lfo_p256_of_scalar(k - d));
    if (d % 2 == 1) lead[k - d] = SUB(zero, lead[k - d]);
  }
  for (size_t i = 1; i < n; ++i) binom[i] = MUL(binom[i - 1], MUL(lfo_p256_of_scalar(n - i), inv[i]));
  for (size_t i = 1; i < n; i += 2) binom[i] = SUB(zero, binom[i]);
  /* y_fft = r2hc(pad(inverses)) / padding (convolution.h:136-152) */
  lfo_e32* yf = (lfo_e32*)calloc(P2, 32);
  memcpy(yf, inv, 32 * m);
  lfo_p256_r2hc(yf, P2);
  lfo_e32 sc = lfo_p256_inv(lfo_p256_of_scalar(P2));
  for (size_t i = 0; i < P2; ++i) yf[i] = MUL(yf[i], sc);
  /* convolution (convolution.h:157-177) */
  lfo_e32* x = (lfo_e32*)calloc(P2, 32);
  for (size_t i = 0; i < n; ++i) x[i] = MUL(binom[i], y[i]);
  lfo_p256_r2hc(x, P2);
  {
    size_t i;
    x[0] = MUL(x[0], yf[0]);
    for (i = 1; i + i < P2; ++i) cmul_(&x[i], &x[P2 - i], yf[i], yf[P2 - i]);
    x[i] = MUL(x[i], yf[i]);
  }
  lfo_p256_hc2r(x, P2);
  for (size_t i = n; i < m; ++i) y[i] = MUL(lead[i - d], x[i]);
  free(inv);
  free(lead);
  free(binom);
  free(yf);
  free(x);
}

/* MerkleCommitment::commit with LigeroCommon<Fp256Base>::column_hash (lib/merkle/merkle_commitment.h:52-61,
 * lib/ligero/ligero_param.h:432-439): 32 canonical little-endian bytes per element */
void lfo_column_leaves32(size_t nrow, size_t ld, size_t col0, size_t ncols, const lfo_e32* T, const uint8_t* nonces, uint8_t* leaves) {
  for (size_t j = 0; j < ncols; ++j) {
    lfo_sha256 s;
    lfo_sha256_init(&s);
    lfo_sha256_update(&s, nonces + 32 * j, 32);
    for (size_t i = 0; i < nrow; ++i) {
      uint8_t buf[32];
      lfo_p256_to_bytes(buf, T[i * ld + col0 + j]);
      lfo_sha256_update(&s, buf, 32);
    }
    lfo_sha256_final(&s, leaves + 32 * j);
  }
}
void lfo_column_commit32(size_t nrow, size_t ld, size_t col0, size_t ncols, const lfo_e32* T, const uint8_t* nonces, uint8_t root_out[32],
                         uint8_t* layers) {
  uint8_t* leaves = (uint8_t*)malloc(32 * ncols);
  uint8_t* lay = layers ? layers : (uint8_t*)malloc(64 * ncols);
  lfo_column_leaves32(nrow, ld, col0, ncols, T, nonces, leaves);
  lfo_merkle_build_tree(ncols, leaves, lay);
  memcpy(root_out, lay + 32, 32);
  free(leaves);
  if (!layers) free(lay);
}
