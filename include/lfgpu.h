/*
 * lfgpu.h -- C ABI of the MI355X (gfx950) prover hot path for longfellow-zk.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Every entry point names the reference interface it replaces
 * (paths relative to the reference checkout, /root/reference in the build
 * container).  Header-only C++ adapters that present the reference's template
 * seams (ReedSolomonFactory, MerkleCommitment, ProverLayers round body) on top
 * of this ABI are in include/lfgpu_adapters.h; INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - All functions return LFGPU_OK (0) or an error code; nothing aborts
 *    (the reference's check() -> abort(), lib/util/panic.h:27-37, becomes a
 *    status).  lfgpu_last_error() returns a message for the last failure.
 *  - Field elements are the reference's in-memory Elt images, 16 bytes:
 *      field 4 (GF2_128): 2 x u64 LE, polynomial basis (lib/gf2k/gf2_128.h:64-89)
 *      field 6 (Fp128)  : 2 x u64 LE, Montgomery form R = 2^128 (lib/algebra/fp_generic.h:66-78)
 *    field 1 (P256, Fp256Base) elements are 32 bytes: 4 x u64 LE, Montgomery form R = 2^256.
 *    so adapters can pass &tableau_[0] straight through.
 *  - Pointers named d_* are DEVICE pointers (HIP); h_* are host pointers.
 *    The *_host variants stage through device memory owned by the context.
 *  - Work is enqueued on the context's stream; functions that return data to
 *    host memory synchronise that stream before returning.
 *  - ld / strides are in ELEMENTS (16 bytes), not bytes.
 */
#ifndef LFGPU_H_
#define LFGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFGPU_OK 0
#define LFGPU_ERR_ARG 1         /* bad argument (null, size, alignment) */
#define LFGPU_ERR_HIP 2         /* HIP runtime error */
#define LFGPU_ERR_UNSUPPORTED 3 /* valid request outside what this build covers */
#define LFGPU_ERR_NOMEM 4
#define LFGPU_ERR_ASSERT 5      /* a reference check() would have failed (e.g. assert-zero term) */
#define LFGPU_ERR_BUSY 6        /* the device could not place a resident kernel's workgroups together (another tenant holds
                                   CUs); the prover-level entry points handle it themselves by falling back to per-launch
                                   kernels -- only a direct caller of the resident-kernel internals can see it */

#define LFGPU_FIELD_GF2_128 4 /* FieldID, lib/proto/circuit_io.h:24-36 */
#define LFGPU_FIELD_FP128 6
#define LFGPU_FIELD_P256 1 /* Fp256Base, the P-256 base field: 32-byte elements (lib/algebra/fp_p256.h); accepted by
                              lfgpu_fp256_rs_encode_rows[_host], lfgpu_column_commit[_host] / lfgpu_column_leaves,
                              lfgpu_field_binop, lfgpu_quad_upload (kvec: 32-byte elements) and the whole prover-level ABI of
                              lfgpu_zk.h (circuits with field id 1); ld / n then count 32-byte elements.  The per-step
                              sumcheck entry points below (partials, scatter, binds, bind_g, sumcheck_layer) take the
                              16-byte fields only: for Fp256Base those steps run inside lfgpu_zk_prove / lfgpu_zk_verify */

typedef struct lfgpu_ctx lfgpu_ctx;

/* ---- context ---------------------------------------------------------- */
int lfgpu_init(int device, lfgpu_ctx** out);
int lfgpu_shutdown(lfgpu_ctx* ctx);
const char* lfgpu_last_error(const lfgpu_ctx* ctx);
int lfgpu_set_stream(lfgpu_ctx* ctx, void* hip_stream); /* NULL = default stream */
/* Gives the context a non-blocking stream of its own (destroyed by lfgpu_shutdown).  Contexts are independent -- nothing
 * below this ABI is process-global except the per-device CU budget of the resident sumcheck kernels -- so K host threads
 * may each drive one context (its circuits, provers and transcripts) on the same device concurrently; with the default
 * (NULL) stream their work would serialise.  This is the throughput mode of BM_ShaZK's loop
 * (lib/circuits/sha/flatsha256_circuit_test.cc:510-536, one proof after the other) on a GPU that one proof leaves idle. */
int lfgpu_own_stream(lfgpu_ctx* ctx);
/* RandomEngine call pattern.  The reference draws element by element (RandomEngine::elt -> Field::sample -> bytes(kBytes),
 * lib/random/random.h:38-48; one bytes(32) per Merkle nonce, lib/merkle/merkle_commitment.h:52-54).  By default the library
 * merges consecutive draws into one call of the lfgpu_rng_fn hook -- the same bytes for every engine that is a byte stream
 * (SecureRandomEngine, Transcript, the LCG test engines) and far fewer calls.  exact = 1 keeps the reference's call pattern,
 * for engines whose output depends on the call boundaries (the TestRng of rust/runtime/zk/tests/zk.rs:283-292 returns
 * {2, 0, 0, ...} for EVERY call).  Applies to lfgpu_ligero_commit and lfgpu_zk_commit on this context. */
int lfgpu_set_rng_exact_calls(lfgpu_ctx* ctx, int exact);
int lfgpu_sync(lfgpu_ctx* ctx);
/* device-memory helpers so that C / ctypes / cgo callers need no HIP runtime */
int lfgpu_malloc(lfgpu_ctx* ctx, size_t bytes, void** d_out);
int lfgpu_free(lfgpu_ctx* ctx, void* d_ptr);
int lfgpu_memcpy_h2d(lfgpu_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int lfgpu_memcpy_d2h(lfgpu_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* ---- K1: prime-field FFT ------------------------------------------------
 * Replaces FFT<Fp128>::fftb / fftf (lib/algebra/fft.h:185-201).
 * In-place, `rows` independent length-n (power of two) transforms; row r starts
 * at d_A + r*ld.  dir 0 = backward  A[j] = sum_k A[k] w_n^{jk}, dir 1 = forward
 * (w -> w^-1); unscaled.  omega is a root of unity of order omega_order (power
 * of two >= n) in Montgomery form, exactly the (omega_j, j) arguments of fftb. */
int lfgpu_fp128_fft(lfgpu_ctx* ctx, int dir, size_t rows, size_t n, const uint64_t omega[2],
                    uint64_t omega_order, void* d_A, size_t ld);
/* Same transform over F64_2 = Fp2<Fp<1>>, p = 2^64 - 2^32 + 1 -- the second field of the reference's FFT tests and
 * benchmarks (lib/algebra/fft_test.cc:205-229, BM_FFT_F64_2).  Elements are the memory image of Fp2<Fp<1>>::Elt
 * (lib/algebra/fp2.h:48-52): {re, im}, 8 bytes each, Montgomery form (R = 2^64), reduced.  omega = {re, im} likewise;
 * a root in the base field (im = 0, as in the reference's test) takes the cheaper twiddle product. */
int lfgpu_f64_2_fft(lfgpu_ctx* ctx, int dir, size_t rows, size_t n, const uint64_t omega[2],
                    uint64_t omega_order, void* d_A, size_t ld);

/* ---- K2: additive (LCH14) FFT over GF(2^128) ------------------------------
 * Replaces LCH14<GF2_128<k>>::FFT / IFFT (lib/gf2k/lch14.h:106-144).
 * subfield_log_bits = k of GF2_128<k> (4 or 5; l <= 2^k).  dir 0 = FFT, 1 = IFFT.
 * In-place on `rows` arrays of 2^l elements, row r at d_B + r*ld. */
int lfgpu_gf2128_lch14_fft(lfgpu_ctx* ctx, int subfield_log_bits, int dir, size_t rows, unsigned l,
                           uint64_t coset, void* d_B, size_t ld);

/* ---- K3 / K4: Reed-Solomon row encode -------------------------------------
 * Replaces LCH14ReedSolomon::interpolate (lib/gf2k/lch14_reed_solomon.h:49-103)
 * and ReedSolomon::interpolate (lib/algebra/reed_solomon.h:93-110) applied to
 * every row of a tableau: y[0..n) valid -> fills y[n..m), in place, row r at
 * d_T + r*ld.  This is the InterpolatorFactory seam of LigeroProver
 * (lib/ligero/ligero_prover.h:34,175,184,210,237,295). */
int lfgpu_gf2128_rs_encode_rows(lfgpu_ctx* ctx, int subfield_log_bits, size_t nrow, size_t n, size_t m,
                                void* d_T, size_t ld);
/* The three interpolate loops of LigeroProver::commit (layout_blinding_rows / layout_witness_rows /
 * layout_quadratic_rows, lib/ligero/ligero_prover.h:171-270) in one launch: rows [lo2, hi2) hold n2 valid values,
 * every other row n1; all are extended to m.  Both lengths must fit the LDS-resident kernel (<= 4096);
 * LFGPU_ERR_UNSUPPORTED otherwise (call lfgpu_gf2128_rs_encode_rows per group instead). */
int lfgpu_gf2128_rs_encode_tableau(lfgpu_ctx* ctx, int subfield_log_bits, size_t nrow, size_t n1, size_t n2, size_t lo2,
                                   size_t hi2, size_t m, void* d_T, size_t ld);
int lfgpu_fp128_rs_encode_rows(lfgpu_ctx* ctx, size_t nrow, size_t n, size_t m, const uint64_t omega[2],
                               uint64_t omega_order, void* d_T, size_t ld);

/* The P-256 base field (BASELINE config 5, the mdoc signature circuit): replaces
 * ReedSolomon<Fp256Base, FFTExtConvolutionFactory<Fp256Base, Fp2<Fp256Base>>>::interpolate
 * (lib/algebra/reed_solomon.h:93-110 over convolution.h:129-191 / rfft.h:282-376; factory built at
 * lib/circuits/mdoc/mdoc_zk.cc:485-487) on every row of a tableau of 32-byte elements; ld counts 32-byte elements.
 * No omega argument: the interpolation does not depend on which 2^k-th root of unity carries the convolution (the
 * device uses the reference's root of order 2^31, mdoc_zk.cc:82-88). */
int lfgpu_fp256_rs_encode_rows(lfgpu_ctx* ctx, size_t nrow, size_t n, size_t m, void* d_T, size_t ld);
/* Host-buffer form: what GpuReedSolomon<Fp256Base>::interpolate (include/lfgpu_adapters.h) forwards to. */
int lfgpu_fp256_rs_encode_rows_host(lfgpu_ctx* ctx, size_t nrow, size_t n, size_t m, void* h_T, size_t ld);

/* ---- K5 + K6: Merkle column commitment ------------------------------------
 * Replaces MerkleCommitment::commit (lib/merkle/merkle_commitment.h:50-64) with
 * LigeroCommon::column_hash as updhash (lib/ligero/ligero_param.h:432-439) and
 * MerkleTree::build_tree (lib/merkle/merkle_tree.h:109-114):
 *   leaf_j = SHA256(nonce_j || ser(T[0][col0+j]) || ... || ser(T[nrow-1][col0+j]))
 *   layers[i] = SHA256(layers[2i] || layers[2i+1]),  i = ncols-1 .. 1
 * d_nonces: ncols*32 bytes (drawn by the caller from its RandomEngine in leaf
 * order, merkle_commitment.h:54).  d_layers: 2*ncols*32 bytes, heap layout of
 * MerkleTree::layers_ (leaves at [ncols, 2*ncols), root at [1]); stays on the
 * device for lfgpu_merkle_open.  root_out: 32 host bytes. */
int lfgpu_column_commit(lfgpu_ctx* ctx, int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                        const void* d_T, const void* d_nonces, void* d_layers, uint8_t root_out[32]);
/* The leaf half of the above alone (enqueue only, nothing read back): d_leaves[j] = leaf digest of column col0 + j,
 * j < ncols, 32 bytes each.  Multi-GPU commit: after the column re-partition a rank hashes the columns it
 * owns, the digests are all-gathered into the leaf level of d_layers and lfgpu_merkle_build_tree finishes on every rank. */
int lfgpu_column_leaves(lfgpu_ctx* ctx, int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                        const void* d_T, const void* d_nonces, void* d_leaves);
/* MerkleTree::build_tree alone: d_layers[ncols..2*ncols) hold the leaves. */
int lfgpu_merkle_build_tree(lfgpu_ctx* ctx, size_t n, void* d_layers, uint8_t root_out[32]);
/* MerkleTree::generate_compressed_proof (lib/merkle/merkle_tree.h:122-143):
 * copies the sibling digests of the opening of pos[0..np) to h_path (capacity
 * path_cap digests) and writes their count to *npath. */
int lfgpu_merkle_open(lfgpu_ctx* ctx, size_t n, const void* d_layers, const size_t* pos, size_t np,
                      uint8_t* h_path, size_t path_cap, size_t* npath);

/* ---- K7 / K8 / K9: sumcheck round body -------------------------------------
 * Replaces the per-round body of ProverLayers::layer (lib/sumcheck/prover_layers.h:230-263). */
/* loop of ProverLayers::evaluations (:365-388): a0 = sum QW[2i] W[2i],
 * a2 = sum (QW[2i+1]-QW[2i]) (W[2i+1]-W[2i]) (+ odd tail).  Outputs to host. */
int lfgpu_sumcheck_partials(lfgpu_ctx* ctx, int field, size_t n, const void* d_QW, const void* d_W,
                            uint64_t a0[2], uint64_t a2[2]);
/* QW[h[hand]] += v * Wother[h[1-hand]] over the HQUAD terms (:235-243).
 * d_hc: n x {u32 h0, u32 h1}; d_vc: n elements; d_QW (nqw elements) is cleared first. */
int lfgpu_qw_scatter(lfgpu_ctx* ctx, int field, size_t n, const void* d_hc, const void* d_vc, int hand,
                     const void* d_Wother, size_t nqw, void* d_QW);
/* Dense::bind with n1 = 1 (lib/arrays/dense.h:70-87): out[i] = in[2i] + r (in[2i+1]-in[2i]);
 * d_out may equal d_in.  New length (n0+1)/2. */
int lfgpu_dense_bind(lfgpu_ctx* ctx, int field, size_t n0, const uint64_t r[2], const void* d_in,
                     void* d_out);
/* HQuad::bind_h (lib/sumcheck/hquad.h:90-123): order-preserving pairwise merge;
 * outputs to (d_hc_out, d_vc_out), *n_out = new count. */
int lfgpu_hquad_bind_h(lfgpu_ctx* ctx, int field, size_t n, const void* d_hc, const void* d_vc,
                       const uint64_t r[2], int hand, void* d_hc_out, void* d_vc_out, size_t* n_out);

/* ---- K10 / K11: sumcheck layer (Quad) resident on the device -----------------------
 * lfgpu_quad_upload: one layer's corners in EXPANDED form, in the order the reference's
 * Quad iterator yields them (canonical: Morton(h0,h1) then g, lib/sumcheck/equad.h:79-106;
 * delta decode of lib/sumcheck/quad.h:100-130 done by the caller), vi indexing the shared
 * constant table kvec (nk elements; a zero constant marks an assert-zero term, quad.h:213-220).
 * nv = number of output gates of the layer.  Upload once per circuit; reuse per proof. */
typedef struct lfgpu_quad lfgpu_quad;
int lfgpu_quad_upload(lfgpu_ctx* ctx, int field, size_t nterms, const uint32_t* g, const uint32_t* h0,
                      const uint32_t* h1, const uint32_t* vi, size_t nk, const void* h_kvec, size_t nv,
                      lfgpu_quad** out);
int lfgpu_quad_free(lfgpu_quad* q);
/* ProverLayers::eval_quad with nc = 1 (lib/sumcheck/prover_layers.h:278-305):
 * V[g] = sum kvec[vi] * W[h1] * W[h0]; *ok = 0 if an assert-zero term is non-zero
 * (the reference returns false / eval_circuit returns nullptr). */
int lfgpu_eval_quad(lfgpu_quad* q, size_t nw, const void* d_W, void* d_V, int* ok);
/* Quad::bind_g (lib/sumcheck/quad.h:152-185): HQUAD[(h0,h1)] = sum_g prep_v(v, beta) *
 * (EQ(G0,g) + alpha EQ(G1,g)); outputs the compact HQuad (d_hc_out: pairs of u32, d_vc_out:
 * elements; capacity nterms) and its size. */
int lfgpu_quad_bind_g(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                      const uint64_t beta[2], void* d_hc_out, void* d_vc_out, size_t* n_out);

/* Eqs::raw_eq2 (lib/arrays/eqs.h): eq[i] = EQ(G0, i) + alpha EQ(G1, i) for i < n <= 2^logn, the
 * vector bind_g folds the corners with and ZkCommon::input_constraint (lib/zk/zk_common.h:406-439) puts on the
 * inputs.  h_G0 / h_G1: logn host elements; d_eq: n device elements. */
int lfgpu_raw_eq2(lfgpu_ctx* ctx, int field, size_t logn, size_t n, const void* h_G0, const void* h_G1,
                  const uint64_t alpha[2], void* d_eq);

/* Quad::bind_gh_all (lib/sumcheck/quad.h:188-210), the verifier's combined bind_g + bind_h:
 * out = sum over the corners of prep_v(v, beta) (EQ(G0,g) + alpha EQ(G1,g)) EQ(H0,h0) EQ(H1,h1).
 * nw = number of input wires of the layer (hand indices are < nw <= 2^logw). */
int lfgpu_quad_bind_gh_all(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                           const uint64_t beta[2], size_t logw, size_t nw, const void* h_H0, const void* h_H1,
                           uint64_t out[2]);

/* ---- one whole sumcheck layer with the transcript behind a callback -------------------
 * Replaces the body of ProverLayers::layer for logc = 0 (every ZK use, lib/zk/zk_common.h:72) together
 * with the bind_g that precedes it (lib/sumcheck/prover_layers.h:140-146,185-271): HQUAD = bind_g(...),
 * then for round < logw, hand in {0,1}: QW scatter, evaluations() -> the 3 evaluations of the round
 * polynomial at poly_evaluation_point(0..2), `round` callback (the caller's round_h: subtract pad, store in
 * the proof, ts.round(poly) -> challenge), bind W[hand] and HQUAD.  The Fiat-Shamir transcript never
 * leaves the caller.  d_W: the layer's nw input wires (device; consumed).  wc_in: the two claims of the
 * previous layer; outputs: wc_out = W[R,C], W[L,C]; g_out[hand][round] = the challenges (next layer's
 * G0/G1); bound_quad = HQUAD->scalar() (ProofAux::bound_quad). */
typedef void (*lfgpu_sc_round_fn)(void* user, size_t hand, size_t round, const uint64_t evals[3][2],
                                  uint64_t challenge_out[2]);
int lfgpu_sumcheck_layer(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                         const uint64_t beta[2], size_t logw, size_t nw, void* d_W, const uint64_t wc_in[2][2],
                         lfgpu_sc_round_fn round, void* user, uint64_t wc_out[2][2], uint64_t* g_out /*[2][logw][2]*/,
                         uint64_t bound_quad[2]);

/* ---- element-wise field ops (Field::addf / subf / mulf, lib/gf2k/gf2_128.h:227-237,
 * lib/algebra/fp_generic.h:203-214): out[i] = a[i] op b[i], op 0 add, 1 sub, 2 mul.  Used by the
 * parity tests to pin the device arithmetic directly. */
int lfgpu_field_binop(lfgpu_ctx* ctx, int field, int op, size_t n, const void* d_a, const void* d_b, void* d_out);

/* ---- K12: Ligero row combinations -----------------------------------------
 * y[j] += sum_i u[i] * T[i][j], j < n  (low_degree_proof, lib/ligero/ligero_prover.h:281-291;
 * Blas::axpy lib/algebra/blas.h:62-68).  h_u: nrows host elements. */
int lfgpu_rows_axpy(lfgpu_ctx* ctx, int field, size_t nrows, size_t n, void* d_y, const uint64_t* h_u,
                    const void* d_T, size_t ld);
/* req[i][j] = T[i][col0 + idx[j]]  (compute_req :346-351; Blas::gather blas.h:104-110) */
int lfgpu_gather_columns(lfgpu_ctx* ctx, size_t nrow, size_t ld, size_t col0, const void* d_T,
                         const size_t* h_idx, size_t nreq, void* d_req);

/* ---- Ligero commit / prove on a device-resident tableau -----------------------
 * The device boundary the survey recommends: inside LigeroProver::commit the witness is
 * uploaded once, the tableau [nrow][block_enc] stays in HBM, and only the root (and later
 * y_ldt / y_dot / y_quad / req) come back.  The Fiat-Shamir transcript and the
 * RandomEngine stay with the caller: randomness is drawn through `rng` in exactly the
 * reference's order (lib/ligero/ligero_prover.h:171-270, lib/merkle/merkle_commitment.h:54). */
typedef struct {
  /* LigeroParam (lib/ligero/ligero_param.h:117-307) */
  size_t nw, nq, rateinv, nreq;
  size_t block_enc, block, dblock, block_ext, r, w, nwrow, nqtriples, nwqrow, nrow, mc_pathlen;
  size_t ildt, idot, iquad, iw, iq;
} lfgpu_ligero_param;
/* LigeroParam(nw, nq, rateinv, nreq, block_enc) ctor (block_enc = 0: the deprecated ctor that searches
 * block_enc over powers of two for the smallest proof, ligero_param.h:152-169); LFGPU_ERR_ARG where the reference
 * would check-fail ("block_enc too large").  field selects kSubFieldBytes (GF2_128<k>: 2^k/8). */
int lfgpu_ligero_param_init(lfgpu_ligero_param* p, int field, int subfield_log_bits, size_t nw, size_t nq,
                            size_t rateinv, size_t nreq, size_t block_enc);
/* RandomEngine::bytes (lib/random/random.h:32-35) */
typedef void (*lfgpu_rng_fn)(void* user, uint8_t* buf, size_t n);
typedef struct lfgpu_ligero_prover lfgpu_ligero_prover;
/* LigeroProver::commit (lib/ligero/ligero_prover.h:58-79) minus the transcript write:
 * layout (blinding / witness / quadratic rows), RS-encode every row, Merkle-commit the
 * columns [dblock, block_enc).  h_W: nw host elements; h_lqc: nq x {x,y,z}. */
int lfgpu_ligero_commit(lfgpu_ctx* ctx, int field, int subfield_log_bits, const lfgpu_ligero_param* p,
                        const void* h_W, size_t subfield_boundary, const size_t* h_lqc, lfgpu_rng_fn rng,
                        void* rng_user, uint8_t root_out[32], lfgpu_ligero_prover** out);
/* LigeroProver::commit split for a row slab [row_lo, row_hi) of the tableau -- the multi-GPU form (rows are independent
 * up to the RandomEngine's draw order, lib/ligero/ligero_prover.h:171-270).
 *  lfgpu_ligero_layout_rows: HOST ONLY (no context, no device).  Performs every RandomEngine draw of commit in the
 *    reference's order -- blinding rows, witness-row pads, quadratic-row pads, then the block_ext Merkle nonces -- and
 *    writes the un-encoded image of the slab's rows to h_rows [(row_hi-row_lo)][dblock] (rows outside the slab are drawn
 *    and dropped).  Ranks that are handed the same byte stream (one engine's output broadcast, or one seed) therefore hold
 *    consistent slabs of one tableau.  h_nonces: block_ext * 32 bytes, or NULL.
 *  lfgpu_ligero_encode_rows: uploads h_rows and RS-extends each row to block_enc into d_slab [(row_hi-row_lo)][block_enc]
 *    (rows IDOT / IQUAD carry dblock values, the others block).
 * lfgpu_ligero_commit is exactly layout_rows + encode_rows on [0, nrow) followed by lfgpu_column_commit. */
int lfgpu_ligero_layout_rows(int field, int subfield_log_bits, const lfgpu_ligero_param* p, const void* h_W,
                             size_t subfield_boundary, const size_t* h_lqc, lfgpu_rng_fn rng, void* rng_user,
                             size_t row_lo, size_t row_hi, void* h_rows, uint8_t* h_nonces);
int lfgpu_ligero_encode_rows(lfgpu_ctx* ctx, int field, int subfield_log_bits, const lfgpu_ligero_param* p,
                             size_t row_lo, size_t row_hi, const void* h_rows, void* d_slab);
/* A prover object over a slab that was laid out and encoded with the two calls above (not owning d_slab / d_layers).
 * The prove entry points below then return the slab's PARTIAL results: rows the slab does not hold contribute zero to
 * y_ldt / y_dot / y_quad (the caller folds the ranks' vectors with the field's addition and checks the W part of y_quad),
 * lfgpu_ligero_open returns the slab's rows of req.  The quadratic rows [iq, nrow) must lie in ONE slab (a triple is
 * multiplied element-wise).  d_layers (whole heap, 2*block_ext*32 bytes) and h_nonces (block_ext*32) are only needed for
 * lfgpu_ligero_open and may be NULL. */
int lfgpu_ligero_prover_from_slab(lfgpu_ctx* ctx, int field, int subfield_log_bits, const lfgpu_ligero_param* p,
                                  size_t row_lo, size_t row_hi, void* d_slab, void* d_layers, const uint8_t* h_nonces,
                                  lfgpu_ligero_prover** out);
/* ---- LigeroProver with the tableau rows sharded over the GPUs of a node, behind this ABI -------------------------
 * One process per GPU; every rank calls the same entry points with the same arguments (SPMD).  The library does the
 * orchestration -- RandomEngine stream drawn once on rank 0 and replayed everywhere, row-slab RS encode (no collective),
 * ONE all_to_all that re-partitions the encoded columns, local column hash, all_gather of the leaf digests, the tree on every
 * rank; the prove entry points below fold the ranks' partial vectors with the FIELD's addition -- and reaches the transport
 * through three hooks, the way the transcript and the RandomEngine are hooks: the caller binds them to RCCL (one
 * ncclAllGather / grouped ncclSend+ncclRecv / ncclBroadcast each; longfellow-zk_amd/parallel.py binds torch.distributed).
 * Buffers are raw bytes; on_device = 1: device memory, and the exchange must be ordered after the work already enqueued on
 * `stream` (the context's stream) and be complete, or enqueued on that stream, when the hook returns; on_device = 0: host.
 * Every hook returns 0 on success.  Results (root, y vectors, opened columns, Merkle path) are identical on every rank and
 * identical to the one-GPU lfgpu_ligero_commit fed the same RandomEngine. */
typedef struct lfgpu_comm_ops {
  void* user;
  int rank, world;
  /* every rank contributes `bytes` bytes; recv gets world * bytes, in rank order */
  int (*all_gather)(void* user, const void* send, void* recv, size_t bytes, int on_device, void* stream);
  /* rank q receives, from every rank p, the send_bytes[q] bytes at send + send_off[q] of p, into recv + recv_off[p]
   * (recv_bytes[p] = what p sends to this rank); the arrays have `world` entries */
  int (*all_to_all)(void* user, const void* send, const size_t* send_off, const size_t* send_bytes, void* recv,
                    const size_t* recv_off, const size_t* recv_bytes, int on_device, void* stream);
  int (*broadcast)(void* user, void* buf, size_t bytes, int root, int on_device, void* stream);
} lfgpu_comm_ops;
/* row slab [*row_lo, *row_hi) of rank `rank`: an even split of the rows, except that the quadratic rows [iq, nrow) all go to
 * the last rank (a triple x, y, z is multiplied element-wise) */
int lfgpu_ligero_row_shard(const lfgpu_ligero_param* p, int rank, int world, size_t* row_lo, size_t* row_hi);
/* The host half alone (no device): rank 0 performs every RandomEngine draw of LigeroProver::commit once (`rng` is only
 * called there), the byte stream is broadcast, and every rank lays out its slab from it (as lfgpu_ligero_layout_rows). */
int lfgpu_ligero_layout_rows_sharded(int field, int subfield_log_bits, const lfgpu_ligero_param* p, const void* h_W,
                                     size_t subfield_boundary, const size_t* h_lqc, lfgpu_rng_fn rng, void* rng_user,
                                     const lfgpu_comm_ops* comm, void* h_rows, uint8_t* h_nonces);
/* LigeroProver::commit (lib/ligero/ligero_prover.h:58-79) over comm->world GPUs.  The prover object it returns is used with
 * the SAME prove entry points as a one-GPU prover (low_degree_proof, dot_proof[_sparse], quadratic_proof, open): they return
 * the complete vectors on every rank. */
int lfgpu_ligero_commit_sharded(lfgpu_ctx* ctx, int field, int subfield_log_bits, const lfgpu_ligero_param* p,
                                const void* h_W, size_t subfield_boundary, const size_t* h_lqc, lfgpu_rng_fn rng,
                                void* rng_user, const lfgpu_comm_ops* comm, uint8_t root_out[32], lfgpu_ligero_prover** out);
/* host-only self-test of a caller's hooks (each with host buffers, world > 1 or 1): 0 when all three move the right bytes */
int lfgpu_comm_selftest(const lfgpu_comm_ops* comm);
/* low_degree_proof (:281-291): y[block] = T[ildt] + sum_i u_ldt[i] T[iw+i] */
int lfgpu_ligero_low_degree_proof(lfgpu_ligero_prover* pr, const void* h_u_ldt, void* h_y);
/* dot_proof (:293-309): y[dblock] = T[idot] + sum_i RS(block->dblock)([0^r | A_i]) (.) T[iw+i] */
int lfgpu_ligero_dot_proof(lfgpu_ligero_prover* pr, const void* h_A, void* h_y);
/* The same with the matrix A built on the device (inner_product_vector + layout_Aext, ligero_param.h:382-430):
 *   A[t] = scale * d_dense[t] for t < ndense  (the private-input block of the last constraint: d_dense is the
 *          device-resident EQ table of lfgpu_raw_eq2 past the public inputs, scale its alphal),
 *   A[h_idx[t]] += h_val[t] for the nsparse remaining terms (host side: strictly increasing flat indices, duplicates
 *          folded by the caller).
 * lfgpu_ligero_inner_product_rows writes the rows [0^r | A_i] at stride ld into d_rows (nrows rows; the first r + w
 * columns of every row are cleared first) -- what LigeroVerifier extends and compares (ligero_verifier.h:150-190). */
int lfgpu_ligero_inner_product_rows(lfgpu_ctx* ctx, int field, size_t w, size_t r, size_t ld, size_t nrows, const void* d_dense,
                                    size_t ndense, const uint64_t scale[2], const uint64_t* h_idx, const void* h_val,
                                    size_t nsparse, void* d_rows);
int lfgpu_ligero_dot_proof_sparse(lfgpu_ligero_prover* pr, const void* d_dense, size_t ndense, const uint64_t scale[2],
                                  const uint64_t* h_idx, const void* h_val, size_t nsparse, void* h_y);
/* quadratic_proof (:311-344); LFGPU_ERR_ASSERT if the W part of y is non-zero */
int lfgpu_ligero_quadratic_proof(lfgpu_ligero_prover* pr, const void* h_u_quad, void* h_y0, void* h_y2);
/* compute_req (:346-351) + MerkleCommitment::open (merkle_commitment.h:66-73) */
int lfgpu_ligero_open(lfgpu_ligero_prover* pr, const size_t* idx, void* h_req, uint8_t* h_nonces,
                      uint8_t* h_path, size_t path_cap, size_t* npath);
/* device pointer of the resident tableau (nrow x block_enc elements) */
int lfgpu_ligero_tableau(lfgpu_ligero_prover* pr, void** d_T);
int lfgpu_ligero_free(lfgpu_ligero_prover* pr);

/* ---- host-buffer conveniences (what the header-only adapters call) ---------- */
int lfgpu_fp128_fft_host(lfgpu_ctx* ctx, int dir, size_t n, const uint64_t omega[2], uint64_t omega_order,
                         void* h_A);
int lfgpu_f64_2_fft_host(lfgpu_ctx* ctx, int dir, size_t n, const uint64_t omega[2], uint64_t omega_order,
                         void* h_A);
int lfgpu_gf2128_lch14_fft_host(lfgpu_ctx* ctx, int subfield_log_bits, int dir, unsigned l, uint64_t coset,
                                void* h_B);
int lfgpu_gf2128_rs_encode_rows_host(lfgpu_ctx* ctx, int subfield_log_bits, size_t nrow, size_t n, size_t m,
                                     void* h_T, size_t ld);
int lfgpu_fp128_rs_encode_rows_host(lfgpu_ctx* ctx, size_t nrow, size_t n, size_t m, const uint64_t omega[2],
                                    uint64_t omega_order, void* h_T, size_t ld);
int lfgpu_column_commit_host(lfgpu_ctx* ctx, int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                             const void* h_T, const uint8_t* h_nonces, uint8_t* h_layers,
                             uint8_t root_out[32]);

#ifdef __cplusplus
}
#endif
#endif /* LFGPU_H_ */
