// zkint.h -- what zk.hip (the 16-byte fields) and zk256.hip (Fp256Base, 32-byte elements) share below the C ABI.
#pragma once
#include <memory>
#include <vector>

#include "../../include/lfgpu_zk.h"
#include "ctx.h"
#include "quad.h"

struct lfgpu_circuit {
  lfgpu_ctx* c = nullptr;
  lfgpu_circuit_info info{};
  struct Layer {
    size_t logw, nw, nterms;
    lfgpu_quad* q;
  };
  std::vector<Layer> layers;
  // nterms zero bytes: what initialize_sumcheck_fiat_shamir hashes per proof (zk_common.h:177-179); shared by the handles of lfgpu_circuit_share
  std::shared_ptr<const std::vector<uint8_t>> zeros;
  ~lfgpu_circuit() {
    for (auto& l : layers)
      if (l.q) lfgpu_quad_free(l.q);
  }
};

// ---- zk256.hip: ZkProver<Fp256Base, .> (BASELINE config 5, the mdoc signature circuit).  Same entry points as the
// 16-byte fields (lfgpu_zk_prover_new / commit / prove / proof_write dispatch on the circuit's field id).
struct Zk256;
int zk256_new(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, Zk256** out);
int zk256_param(const Zk256* z, lfgpu_ligero_param* p);
int zk256_commit(Zk256* z, const void* h_W, lfgpu_rng_fn rng, void* rng_user, const lfgpu_transcript_ops* ts, uint8_t root_out[32], bool draws_only = false);
int zk256_prove(Zk256* z, const void* h_W, const lfgpu_transcript_ops* ts, int* ok);
int zk256_proof_write(const Zk256* z, uint8_t* buf, size_t cap, size_t* nbytes);
int zk256_timings(const Zk256* z, double ms[6]);
void zk256_free(Zk256* z);
void zk256_set_comm(Zk256* z, const lfgpu_comm_ops* comm, size_t min_tableau_bytes);  // comm == nullptr: one GPU
int zk256_verify(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, const uint8_t* proof, size_t proof_len, const void* h_pub,
                 const lfgpu_transcript_ops* ts, bool committed, int* ok, const char** why);
// MerkleTreeVerifier::verify_compressed_proof (lib/merkle/merkle_tree.h:160-209), host (zk.hip)
bool lf_merkle_verify(size_t n, const uint8_t root[32], const uint8_t* path, size_t npath, const uint8_t* leaves, const size_t* pos, size_t np);
// K11 over 32-byte elements (quad.hip forwards field 1 here)
int lf256_eval_quad_async(lfgpu_quad* q, const void* d_W, void* d_V, int* d_fail);
