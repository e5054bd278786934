/*
 * lf_oracle.h -- CPU ORACLE for the longfellow-zk prover hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it, and only as the checker.  The product path lives in
 * longfellow-zk_amd/csrc and fails loudly when the HIP library is missing.
 *
 * Plain-C restatement of the reference algorithms (file:line cited on every
 * function in lf_oracle.c).  Parity status: PINNED -- see tests/test_oracle_*.py:
 *   - against the real reference compiled into oracle/_ref (this container),
 *   - against the reference's own KATs (beta(1), Merkle root of testvectors.md,
 *     rust/runtime/{ligero,merkle} *.bin fixtures copied as data to tests/golden),
 *   - against golden vectors generated from oracle/_ref (tests/golden/*.bin).
 *
 * Element layout (both fields): 16 bytes = two little-endian uint64 limbs,
 * exactly the reference's in-memory Elt (GF2_128: lane0 = bits 0..63,
 * lib/gf2k/sysdep.h:31-44; Fp128: Montgomery form R=2^128, lib/algebra/fp_generic.h:66-78).
 */
#ifndef LF_ORACLE_H_
#define LF_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  uint64_t l[2];
} lfo_elt;

enum { LFO_FIELD_GF2_128 = 4, LFO_FIELD_FP128 = 6 }; /* FieldID, lib/proto/circuit_io.h:24-36 */

/* ------------------------------------------------------------------ GF(2^128) */
lfo_elt lfo_gf_mul(lfo_elt a, lfo_elt b);
lfo_elt lfo_gf_mul_bitserial(lfo_elt a, lfo_elt b); /* slow cross-check */
lfo_elt lfo_gf_inv(lfo_elt a);

typedef struct {
  unsigned k;            /* subfield_log_bits: 4 -> GF(2^16), 5 -> GF(2^32) */
  unsigned sub_bits;     /* 1<<k */
  lfo_elt g;             /* subfield generator */
  lfo_elt beta[32];      /* beta[i] = g^i */
  lfo_elt w_hat[32][32]; /* LCH14 normalised subspace polynomials at beta_j */
} lfo_gf_ctx;

void lfo_gf_ctx_init(lfo_gf_ctx* c, unsigned subfield_log_bits);
lfo_elt lfo_gf_of_scalar(const lfo_gf_ctx* c, uint64_t u);
lfo_elt lfo_gf_poly_evaluation_point(const lfo_gf_ctx* c, unsigned i);

/* LCH14 additive FFT */
lfo_elt lfo_lch14_twiddle(const lfo_gf_ctx* c, unsigned i, uint64_t u);
void lfo_lch14_fft(const lfo_gf_ctx* c, unsigned l, uint64_t coset, lfo_elt* B);
void lfo_lch14_ifft(const lfo_gf_ctx* c, unsigned l, uint64_t coset, lfo_elt* B);
void lfo_lch14_bidirectional_fft(const lfo_gf_ctx* c, unsigned l, uint64_t k, lfo_elt* B);
/* y[0..n) valid -> fills y[n..m) (LCH14ReedSolomon::interpolate) */
void lfo_lch14_rs_interpolate(const lfo_gf_ctx* c, size_t n, size_t m, lfo_elt* y);

/* ------------------------------------------------------------------ Fp128 */
lfo_elt lfo_fp_add(lfo_elt a, lfo_elt b);
lfo_elt lfo_fp_sub(lfo_elt a, lfo_elt b);
lfo_elt lfo_fp_mul(lfo_elt a, lfo_elt b);     /* Montgomery product */
lfo_elt lfo_fp_to_mont(lfo_elt raw);          /* raw < p */
lfo_elt lfo_fp_from_mont(lfo_elt x);
lfo_elt lfo_fp_of_scalar(uint64_t u);
lfo_elt lfo_fp_inv(lfo_elt x);
lfo_elt lfo_fp_omega32(void);                 /* root of unity of order 2^32, Montgomery */
/* in-place backward DFT, A[j] = sum_k A[k] w_n^{jk}; omega_j of order j (power of two) */
void lfo_fp_fftb(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j);
void lfo_fp_fftf(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j);
/* y[0..n) = evaluations at 0..n-1 of deg<n poly -> fill y[n..m) (ReedSolomon::interpolate) */
void lfo_fp_rs_interpolate(size_t n, size_t m, lfo_elt* y);

/* ------------------------------------------------------------------ F64 = Fp<1> (p = 2^64 - 2^32 + 1) and F64_2 = Fp2<F64>
 * (lib/algebra/fft_test.cc:205-229).  An F64_2 element is lfo_elt{l[0] = re, l[1] = im}: the memory image of Fp2<Fp<1>>::Elt. */
uint64_t lfo_f64_add(uint64_t a, uint64_t b);
uint64_t lfo_f64_sub(uint64_t a, uint64_t b);
uint64_t lfo_f64_mul(uint64_t a, uint64_t b);  /* Montgomery product, R = 2^64 */
uint64_t lfo_f64_of_scalar(uint64_t u);        /* u < p */
uint64_t lfo_f64_from_mont(uint64_t x);
uint64_t lfo_f64_inv(uint64_t x);
uint64_t lfo_f64_omega32(void);                /* root of unity of order 2^32, Montgomery */
lfo_elt lfo_f64_2_add(lfo_elt a, lfo_elt b);
lfo_elt lfo_f64_2_sub(lfo_elt a, lfo_elt b);
lfo_elt lfo_f64_2_mul(lfo_elt a, lfo_elt b);
lfo_elt lfo_f64_2_inv(lfo_elt a);
void lfo_f64_2_fftb(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j);
void lfo_f64_2_fftf(lfo_elt* A, size_t n, lfo_elt omega_j, uint64_t j);
void lfo_f64_2_bogorng_fill(uint64_t seed, int imag, size_t n, lfo_elt* out);

/* ------------------------------------------------------------------ field-generic (field = LFO_FIELD_*) */
lfo_elt lfo_add(int field, lfo_elt a, lfo_elt b);
lfo_elt lfo_sub(int field, lfo_elt a, lfo_elt b);
lfo_elt lfo_mul(int field, lfo_elt a, lfo_elt b);
void lfo_to_bytes(int field, uint8_t out[16], lfo_elt x);

/* ------------------------------------------------------------------ SHA-256 / Merkle */
typedef struct {
  uint32_t h[8];
  uint8_t buf[64];
  uint64_t len;
} lfo_sha256;
void lfo_sha256_init(lfo_sha256* s);
void lfo_sha256_update(lfo_sha256* s, const uint8_t* p, size_t n);
void lfo_sha256_final(lfo_sha256* s, uint8_t out[32]);

/* leaves[n][32] -> layers[2n][32] (layers[1] = root), MerkleTree::build_tree */
void lfo_merkle_build_tree(size_t n, const uint8_t* leaves, uint8_t* layers);
/* leaf_j = SHA256(nonce_j || ser(T[0][col0+j]) || ... || ser(T[nrow-1][col0+j])) */
void lfo_column_leaves(int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                       const lfo_elt* tableau, const uint8_t* nonces, uint8_t* leaves);
/* leaves + tree; root_out[32]; layers may be NULL */
void lfo_column_commit(int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                       const lfo_elt* tableau, const uint8_t* nonces, uint8_t root_out[32],
                       uint8_t* layers);

/* ------------------------------------------------------------------ sumcheck pieces */
/* a0 = sum QW[2i] W[2i]; a2 = sum (QW[2i+1]-QW[2i])(W[2i+1]-W[2i]) (+ odd tail) */
void lfo_sumcheck_partials(int field, size_t n, const lfo_elt* QW, const lfo_elt* W,
                           lfo_elt* a0, lfo_elt* a2);
/* ProverLayers::evaluations: evals[3] at poly_evaluation_point(0..2) */
void lfo_sumcheck_evaluations(int field, const lfo_gf_ctx* c, size_t n, lfo_elt eq0,
                              const lfo_elt* QW, const lfo_elt* W, lfo_elt sum,
                              lfo_elt evals[3]);
/* Dense::bind, n1 = 1: out[i] = in[2i] + r (in[2i+1]-in[2i]); returns new n0 */
size_t lfo_dense_bind(int field, size_t n0, lfo_elt r, const lfo_elt* in, lfo_elt* out);
/* HQuad::bind_h in place; hc = pairs (h0,h1) of uint32; returns new n */
size_t lfo_hquad_bind_h(int field, size_t n, uint32_t* hc, lfo_elt* vc, lfo_elt r, int hand);
/* QW[h[hand]] += v * Wother[h[1-hand]] */
void lfo_qw_scatter(int field, size_t n, const uint32_t* hc, const lfo_elt* vc, int hand,
                    const lfo_elt* Wother, size_t nqw, lfo_elt* QW);

/* ------------------------------------------------------------------ quad (expanded corners) */
/* eq[i] = EQ(G0,i) + alpha*EQ(G1,i), i < n  (Eqs::raw_eq2) */
void lfo_raw_eq2(int field, size_t logn, size_t n, const lfo_elt* G0, const lfo_elt* G1, lfo_elt alpha, lfo_elt* eq);
/* ProverLayers::eval_quad with nc = 1: V[g] += kvec[vi]*W[h1]*W[h0]; assert-zero terms (kvec[vi]==0)
 * require W[h1]*W[h0] == 0.  Returns 1 if ok, 0 if an assertion failed. */
int lfo_eval_quad(int field, size_t nterms, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                  const uint32_t* vi, const lfo_elt* kvec, size_t nv, const lfo_elt* W, lfo_elt* V);
/* Quad::bind_g: hc_out[2*j..], vc_out[j]; returns the HQuad size */
size_t lfo_quad_bind_g(int field, size_t nterms, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                       const uint32_t* vi, const lfo_elt* kvec, size_t logv, const lfo_elt* G0, const lfo_elt* G1,
                       lfo_elt alpha, lfo_elt beta, uint32_t* hc_out, lfo_elt* vc_out);

/* Quad::bind_gh_all (lib/sumcheck/quad.h:188-210): sum_t prep_v(v_t, beta) eqg[g_t] EQ(H0,h0_t) EQ(H1,h1_t),
 * eqg = raw_eq2(G0, G1, alpha) */
lfo_elt lfo_quad_bind_gh_all(int field, size_t nterms, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                             const uint32_t* vi, const lfo_elt* kvec, size_t logv, size_t nv, const lfo_elt* G0,
                             const lfo_elt* G1, lfo_elt alpha, lfo_elt beta, size_t logw, size_t nw, const lfo_elt* H0,
                             const lfo_elt* H1);

/* ------------------------------------------------------------------ Ligero row combos (Blas) */
/* y[j] += a * x[j] */
void lfo_axpy(int field, size_t n, lfo_elt* y, lfo_elt a, const lfo_elt* x);
/* y[j] += a[j] * x[j] */
void lfo_vaxpy(int field, size_t n, lfo_elt* y, const lfo_elt* a, const lfo_elt* x);

/* test input generator: x <- x * 7300988, x0 = seed (lib/algebra/bogorng.h:43-51), Fp128 Montgomery */
void lfo_fp_bogorng_fill(uint64_t seed, size_t n, lfo_elt* out);
/* deterministic GF(2^128) filler: splitmix64 stream */
void lfo_gf_fill(uint64_t seed, size_t n, lfo_elt* out);

/* ---- P-256 base field Fp256Base + RFFT<Fp2<Fp256Base>> Reed-Solomon + 32-byte column hash (lf_oracle_p256.c):
 * BASELINE config 5's signature tableau (lib/algebra/{fp_p256.h,rfft.h,convolution.h:129-191,reed_solomon.h}) */
typedef struct { uint64_t l[4]; } lfo_e32; /* Fp256Base::Elt image: Montgomery form, 4 x u64 LE */
lfo_e32 lfo_p256_add(lfo_e32 a, lfo_e32 b);
lfo_e32 lfo_p256_sub(lfo_e32 a, lfo_e32 b);
lfo_e32 lfo_p256_mul(lfo_e32 a, lfo_e32 b);
lfo_e32 lfo_p256_to_mont(lfo_e32 raw);
lfo_e32 lfo_p256_from_mont(lfo_e32 x);
lfo_e32 lfo_p256_of_scalar(uint64_t u);
lfo_e32 lfo_p256_inv(lfo_e32 x);
void lfo_p256_to_bytes(uint8_t out[32], lfo_e32 x);
void lfo_p256_fill(uint64_t seed, size_t n, lfo_e32* out);
void lfo_p256_omega(lfo_e32* re, lfo_e32* im); /* root of order 2^31 of Fp2 (mdoc_zk.cc:82-88) */
void lfo_p256_r2hc(lfo_e32* A, size_t n);      /* RFFT::r2hc with that root */
void lfo_p256_hc2r(lfo_e32* A, size_t n);
void lfo_p256_rs_interpolate(size_t n, size_t m, lfo_e32* y);
void lfo_column_leaves32(size_t nrow, size_t ld, size_t col0, size_t ncols, const lfo_e32* T, const uint8_t* nonces, uint8_t* leaves);
void lfo_column_commit32(size_t nrow, size_t ld, size_t col0, size_t ncols, const lfo_e32* T, const uint8_t* nonces,
                         uint8_t root_out[32], uint8_t* layers);

#ifdef __cplusplus
}
#endif
#endif /* LF_ORACLE_H_ */
