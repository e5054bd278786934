/* lfgpu_zk.h -- C ABI of the ZK prover host driver (the callers of the hot path).
 *
 * lfgpu.h is the kernel-level boundary; this header is one level up: the host control flow of
 *   ZkProver::commit / ZkProver::prove        lib/zk/zk_prover.h:72-149
 *   ZkCommon::verifier_constraints            lib/zk/zk_common.h:49-136,406-439
 *   LigeroProver::prove                       lib/ligero/ligero_prover.h:84-146
 *   ZkProof::write                            lib/zk/zk_proof.h:90-185
 *   CircuitRep::from_bytes (LFC1)             lib/proto/circuit.h, lib/proto/circuit_reader.h:55-233
 * written in C++ inside the library, with every data-parallel step on the device through the lfgpu.h kernels.
 * Fields: GF2_128<4> (the field of BM_ShaZK_fp2_128, lib/circuits/sha/flatsha256_circuit_test.cc:510-536; LCH14
 * Reed-Solomon) and Fp128 (field id 6; ReedSolomonFactory over FFTConvolutionFactory with the 2^32-order root, as
 * lib/zk/zk_test.cc:252-330 sets it up); other field ids in the circuit header return LFGPU_ERR_UNSUPPORTED.
 * Elements cross the ABI as the reference's in-memory Elt images (Fp128: Montgomery form) and the wire as
 * to_bytes_field images.
 *
 * The Fiat-Shamir transcript and the RandomEngine are the CALLER's: they are reached through the hooks below, so
 * an integration passes thin wrappers over proofs::Transcript / proofs::RandomEngine.  A built-in
 * transcript with the reference's exact construction (SHA-256 state + AES-256-ECB counter PRF,
 * lib/random/transcript.h:33-190) is provided for stand-alone use and for the tests.
 */
#ifndef LFGPU_ZK_H_
#define LFGPU_ZK_H_

#include "lfgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Fiat-Shamir transcript hooks (proofs::Transcript, lib/random/transcript.h:70-190) ---- */
typedef struct lfgpu_transcript_ops {
  void* user;
  /* Transcript::write(const uint8_t*, size_t): tag 0 || u64 length || bytes */
  void (*write_bytes)(void* user, const uint8_t* data, size_t n);
  /* Transcript::write(const Elt&, F): tag 1 || to_bytes_field image (elt16 = that 16-byte image) */
  void (*write_elt)(void* user, const uint8_t* elt16);
  /* Transcript::write(const Elt[], ince, n, F): tag 2 || u64 count || images */
  void (*write_elt_array)(void* user, const uint8_t* elts16, size_t n);
  /* RandomEngine::bytes: n PRF bytes */
  void (*gen_bytes)(void* user, uint8_t* out, size_t n);
  /* Transcript::clone(): a new `user` with the same state (ZkProver::prove runs the sumcheck prover on a
   * copy, zk_prover.h:117-124); release with free_clone */
  void* (*clone)(void* user);
  void (*free_clone)(void* user);
  /* the element writes for a field whose to_bytes_field image is not 16 bytes (Fp256Base, field id 1: nbytes = 32):
   * tag 1 || image[nbytes];  tag 2 || u64 count || count images.  The 16-byte fields never call them; a transcript that
   * only serves those may leave them NULL (the P-256 prover then fails with LFGPU_ERR_ARG). */
  void (*write_elt_sized)(void* user, const uint8_t* elt, size_t nbytes);
  void (*write_elt_array_sized)(void* user, const uint8_t* elts, size_t n, size_t nbytes);
} lfgpu_transcript_ops;

/* built-in transcript, byte-identical to the reference's (pinned by the ZK fixtures and FIPS-197/180-4 vectors) */
typedef struct lfgpu_transcript lfgpu_transcript;
lfgpu_transcript* lfgpu_transcript_new(const uint8_t* init, size_t n); /* Transcript(init, n) */
void lfgpu_transcript_free(lfgpu_transcript* t);
void lfgpu_transcript_get_ops(lfgpu_transcript* t, lfgpu_transcript_ops* ops);
/* direct access for tests */
void lfgpu_transcript_write_bytes(lfgpu_transcript* t, const uint8_t* data, size_t n);
void lfgpu_transcript_write_elt(lfgpu_transcript* t, const uint8_t* elt16);
void lfgpu_transcript_write_elt_array(lfgpu_transcript* t, const uint8_t* elts16, size_t n);
void lfgpu_transcript_bytes(lfgpu_transcript* t, uint8_t* out, size_t n);
/* ... for elements whose to_bytes_field image is nbytes long (Fp256Base: 32) */
void lfgpu_transcript_write_elt_sized(lfgpu_transcript* t, const uint8_t* elt, size_t nbytes);
void lfgpu_transcript_write_elt_array_sized(lfgpu_transcript* t, const uint8_t* elts, size_t n, size_t nbytes);
/* primitives the transcript is built from (known-answer tested) */
void lfgpu_sha256(const uint8_t* data, size_t n, uint8_t out[32]);
void lfgpu_aes256_ecb_block(const uint8_t key[32], const uint8_t in[16], uint8_t out[16]);
/* host-side GF(2^128) product used by the driver's bookkeeping (GF2_128::mulf, lib/gf2k/gf2_128.h:233-246):
 * PCLMULQDQ when available, portable otherwise */
void lfgpu_host_gf2128_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);
/* SHA-NI / AES-NI / PCLMULQDQ dispatch: force_portable = 1 / 0 switches the portable C++ paths on / off, < 0 only queries;
 * returns 1 when the hardware paths are active. */
int lfgpu_crypto_hw(int force_portable);

/* ---- circuit (LFC1 wire format -> device-resident layers) ---- */
typedef struct lfgpu_circuit lfgpu_circuit;
typedef struct {
  int field; /* LFGPU_FIELD_* */
  size_t nv, nc, npub_in, subfield_boundary, ninputs, nl, logv, nterms;
  uint8_t id[32];
} lfgpu_circuit_info;
/* CircuitRep::from_bytes (lib/proto/circuit.h): parses the LFC1 bytes, delta-decodes every layer's corners
 * (circuit_reader.h:55-233) and uploads them with lfgpu_quad_upload.  LFGPU_ERR_ARG on malformed input. */
int lfgpu_circuit_from_lfc1(lfgpu_ctx* ctx, const uint8_t* bytes, size_t len, lfgpu_circuit** out);
/* A second handle on the same uploaded circuit for ANOTHER context of the same device (throughput mode: K host threads, each
 * with its own context -- lfgpu_own_stream -- prover and transcript, one copy of the circuit in HBM).  The device arrays are
 * reference-counted: the handles may be freed in any order.  `src` must not be in use by another thread during the call. */
int lfgpu_circuit_share(lfgpu_ctx* ctx, const lfgpu_circuit* src, lfgpu_circuit** out);
int lfgpu_circuit_get_info(const lfgpu_circuit* c, lfgpu_circuit_info* info);
int lfgpu_circuit_layer_info(const lfgpu_circuit* c, size_t layer, size_t* logw, size_t* nw, size_t* nterms);
int lfgpu_circuit_free(lfgpu_circuit* c);

/* ---- ZkProver ---- */
typedef struct lfgpu_zk_prover lfgpu_zk_prover;
/* ZkProver(c, F, rsf) + ZkProof(c, rateinv, nreq[, block_enc]) (zk_proof.h:63-76): block_enc = 0 searches. */
int lfgpu_zk_prover_new(lfgpu_ctx* ctx, const lfgpu_circuit* c, size_t rateinv, size_t nreq, size_t block_enc,
                        lfgpu_zk_prover** out);
int lfgpu_zk_prover_param(const lfgpu_zk_prover* zk, lfgpu_ligero_param* p);
/* More than one GPU behind the same prover (BASELINE config 5 names it).  Every rank constructs the prover on its
 * own device and calls commit / prove / proof_write with the same arguments and its own copy of the transcript (SPMD); `rng`
 * is only drawn from on rank 0 (its stream is broadcast).  A Ligero tableau of at least min_tableau_bytes is committed with
 * its rows sharded over the communicator's GPUs (lfgpu_ligero_commit_sharded in lfgpu.h: one all_to_all + one all_gather per
 * commit, field-sum folds of the y vectors in prove); below the threshold the proof runs replicated, which is what the
 * real mdoc / flatsha256 circuits want (their tableaux are 2.5 - 20 MB: a second GPU only adds latency -- use independent
 * proofs per GPU instead, lfgpu_own_stream / bench.py zk_throughput).  The sumcheck always runs replicated.  Every rank ends
 * with the same proof bytes as a one-GPU prover fed the same RandomEngine.  comm = NULL: back to one GPU.  All three
 * fields (the mdoc signature circuit over Fp256Base has a 2.5 MB tableau: leave it replicated). */
int lfgpu_zk_prover_set_comm(lfgpu_zk_prover* zk, const lfgpu_comm_ops* comm, size_t min_tableau_bytes);
/* ZkProver::commit (zk_prover.h:72-96): fill_pad from `rng`, Ligero-commit witness||pad, root -> transcript.
 * h_W: ninputs host elements (public inputs first). */
int lfgpu_zk_commit(lfgpu_zk_prover* zk, const void* h_W, lfgpu_rng_fn rng, void* rng_user,
                    const lfgpu_transcript_ops* ts, uint8_t root_out[32]);
/* ZkProver::prove (zk_prover.h:98-149): *ok = 0 when the witness does not satisfy the circuit (the reference
 * returns false); otherwise the proof is held by the prover object. */
int lfgpu_zk_prove(lfgpu_zk_prover* zk, const void* h_W, const lfgpu_transcript_ops* ts, int* ok);
/* ZkProof::write (zk_proof.h:90-185): the wire bytes of commitment || sumcheck proof || Ligero proof.
 * Call with buf = NULL to get the size. */
int lfgpu_zk_proof_write(const lfgpu_zk_prover* zk, uint8_t* buf, size_t cap, size_t* nbytes);
/* host milliseconds spent in the last commit / prove, split by phase:
 * [0] commit total, [1] prove total, [2] eval_circuit, [3] sumcheck, [4] constraints, [5] ligero prove */
int lfgpu_zk_timings(const lfgpu_zk_prover* zk, double ms[6]);
int lfgpu_zk_prover_free(lfgpu_zk_prover* zk);

/* ---- ZkVerifier ----
 * ZkVerifier::recv_commitment + verify (lib/zk/zk_verifier.h:68-94) on the wire bytes of ZkProof::write
 * (parsed as ZkProof::read does, lib/zk/zk_proof.h:107-112,218-345).  h_pub: the npub_in public inputs.
 * *ok = 1 iff the proof is accepted; *why (optional) names the failing check with the reference's strings
 * (lib/ligero/ligero_verifier.h:90-130) or "proof does not parse".  The transcript must be in the same
 * state the prover's was before its commit (e.g. a fresh Transcript over the same seed). */
int lfgpu_zk_verify(lfgpu_ctx* ctx, const lfgpu_circuit* c, size_t rateinv, size_t nreq, size_t block_enc,
                    const uint8_t* proof, size_t proof_len, const void* h_pub, const lfgpu_transcript_ops* ts, int* ok,
                    const char** why);

/* The same with ZkVerifier::recv_commitment already done by the caller (the commitment root -- the first 32 proof bytes --
 * written to the transcript with write_bytes): the reference keeps the two calls apart, and run_mdoc_verifier receives both
 * commitments and draws its MAC key before it verifies either proof (lib/circuits/mdoc/mdoc_zk.cc:676-681,704-705). */
int lfgpu_zk_verify_committed(lfgpu_ctx* ctx, const lfgpu_circuit* c, size_t rateinv, size_t nreq, size_t block_enc,
                              const uint8_t* proof, size_t proof_len, const void* h_pub, const lfgpu_transcript_ops* ts,
                              int* ok, const char** why);

#ifdef __cplusplus
}
#endif
#endif /* LFGPU_ZK_H_ */
