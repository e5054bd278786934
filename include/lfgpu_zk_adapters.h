// lfgpu_zk_adapters.h -- header-only C++17 adapter that presents the reference's ZkProver on top of the prover-level C ABI
// (include/lfgpu_zk.h).  Where lfgpu_adapters.h swaps a template argument INSIDE the reference's prover (the Reed-Solomon
// interpolator), this one swaps the prover itself: the caller keeps its Circuit, Dense, ZkProof, Transcript and
// RandomEngine objects and its two calls,
//
//     ZkProver<Field, RSFactory> p(circuit, F, rsf);      ->   lfgpu::GpuZkProver<Field, proofs::ReadBuffer> p(ctx, lfc1, len, F);
//     p.commit(zkp, W, tp, rng);                                p.commit(zkp, W, tp, rng);
//     if (!p.prove(zkp, W, tp)) ...                             if (!p.prove(zkp, W, tp)) ...
//
// (reference lib/zk/zk_prover.h:60-149; call sites lib/circuits/mdoc/mdoc_zk.cc:494-522, the body of run_mdoc_prover.)
// Evaluation of the circuit, the padded sumcheck, the constraint replay and the Ligero commit / prove then run in
// liblfgpu.so (csrc/zk.hip for GF2_128 / Fp128, csrc/zk256.hip for Fp256Base); the Fiat-Shamir transcript and the random
// engine stay the CALLER's objects, reached through hooks, so the bytes written to / drawn from them are the reference's.
// The proof comes back through ZkProof::read on the wire bytes, i.e. in exactly the form a verifier would receive it.
//
// lfgpu::GpuZkVerifier<Field> does the same for ZkVerifier (recv_commitment / verify).
//
// `lfc1` = the circuit in the reference's own serialisation (CircuitWriter::to_bytes, or the span of the decompressed
// circuit file that CircuitReader::from_bytes consumed for this circuit).  This header includes no reference header.
#ifndef LFGPU_ZK_ADAPTERS_H_
#define LFGPU_ZK_ADAPTERS_H_

#include <optional>

#include "lfgpu_adapters.h"
#include "lfgpu_zk.h"

namespace lfgpu {

namespace detail {
// the caller's Transcript (lib/random/transcript.h:70-190) behind lfgpu_transcript_ops
template <class Field, class TranscriptT>
struct TranscriptHook {
  using Elt = typename Field::Elt;
  TranscriptT* t;
  const Field* F;
  std::unique_ptr<TranscriptT> owned;  // clones own their transcript
  static TranscriptHook* self(void* u) { return static_cast<TranscriptHook*>(u); }
  static Elt elt(const Field& F, const uint8_t* image) {
    auto e = F.of_bytes_field(image);  // the library only ever hands over images of field elements
    if (!e.has_value()) check(nullptr, LFGPU_ERR_ASSERT, "transcript hook: not a field element");
    return e.value();
  }
  static void write_bytes(void* u, const uint8_t* d, size_t n) { self(u)->t->write(d, n); }
  static void write_elt(void* u, const uint8_t* e) { self(u)->t->write(elt(*self(u)->F, e), *self(u)->F); }
  static void write_elt_array(void* u, const uint8_t* e, size_t n) {
    std::vector<Elt> v(n ? n : 1);
    for (size_t i = 0; i < n; ++i) v[i] = elt(*self(u)->F, e + Field::kBytes * i);
    self(u)->t->write(v.data(), 1, n, *self(u)->F);
  }
  static void write_elt_sized(void* u, const uint8_t* e, size_t) { write_elt(u, e); }
  static void write_elt_array_sized(void* u, const uint8_t* e, size_t n, size_t) { write_elt_array(u, e, n); }
  static void gen_bytes(void* u, uint8_t* o, size_t n) { self(u)->t->bytes(o, n); }
  static void* clone(void* u) {  // Transcript::clone (:95-99); guaranteed elision: Transcript has no copy / move constructor
    TranscriptHook* h = new TranscriptHook{nullptr, self(u)->F, nullptr};
    h->owned.reset(new TranscriptT(self(u)->t->clone()));
    h->t = h->owned.get();
    return h;
  }
  static void free_clone(void* u) { delete self(u); }
  lfgpu_transcript_ops ops() {
    lfgpu_transcript_ops o;
    o.user = this;
    o.write_bytes = write_bytes;
    o.write_elt = write_elt;
    o.write_elt_array = write_elt_array;
    o.gen_bytes = gen_bytes;
    o.clone = clone;
    o.free_clone = free_clone;
    o.write_elt_sized = write_elt_sized;
    o.write_elt_array_sized = write_elt_array_sized;
    return o;
  }
};
template <class RandomEngineT>
void rng_hook(void* user, uint8_t* buf, size_t n) {  // RandomEngine::bytes (lib/random/random.h:32-35)
  static_cast<RandomEngineT*>(user)->bytes(buf, n);
}
}  // namespace detail

// Drop-in for ZkProver<Field, ReedSolomonFactory> (lib/zk/zk_prover.h:45-149).  ReadBufferT = proofs::ReadBuffer.
template <class Field, class ReadBufferT>
class GpuZkProver {
 public:
  // comm != nullptr: one process per GPU, every rank constructs the prover on its own Context and calls commit / prove with the
  // same arguments (SPMD; only rank 0's RandomEngine is drawn from).  Ligero tableaux of at least min_tableau_bytes are
  // committed with their rows sharded over the communicator's GPUs (lfgpu_zk_prover_set_comm, include/lfgpu_zk.h); the hooks
  // are the caller's binding of all_gather / all_to_all / broadcast to RCCL.  All three fields.
  GpuZkProver(const Context& ctx, const uint8_t* lfc1, size_t len, const Field& F, const lfgpu_comm_ops* comm = nullptr, size_t min_tableau_bytes = 0)
      : c_(ctx), f_(F), have_comm_(comm != nullptr), min_tableau_bytes_(min_tableau_bytes) {
    if (comm) comm_ = *comm;
    check(c_.get(), lfgpu_circuit_from_lfc1(c_.get(), lfc1, len, &circuit_), "lfgpu_circuit_from_lfc1");
  }
  ~GpuZkProver() {
    if (zk_) lfgpu_zk_prover_free(zk_);
    if (circuit_) lfgpu_circuit_free(circuit_);
  }
  GpuZkProver(const GpuZkProver&) = delete;
  GpuZkProver& operator=(const GpuZkProver&) = delete;

  // ZkProver::commit (zk_prover.h:72-96).  zkp.param carries rate, nreq and block_enc as the ZkProof constructor set them.
  template <class ZkProofT, class DenseT, class TranscriptT, class RandomEngineT>
  void commit(ZkProofT& zkp, const DenseT& W, TranscriptT& tp, RandomEngineT& rng) {
    if (!zk_) {
      check(c_.get(), lfgpu_zk_prover_new(c_.get(), circuit_, zkp.param.rateinv, zkp.param.nreq, zkp.param.block_enc, &zk_), "lfgpu_zk_prover_new");
      if (have_comm_) check(c_.get(), lfgpu_zk_prover_set_comm(zk_, &comm_, min_tableau_bytes_), "lfgpu_zk_prover_set_comm");
      lfgpu_ligero_param p;
      check(c_.get(), lfgpu_zk_prover_param(zk_, &p), "lfgpu_zk_prover_param");
      if (p.nrow != zkp.param.nrow || p.block != zkp.param.block || p.nw != zkp.param.nw || p.block_ext != zkp.param.block_ext)
        check(c_.get(), LFGPU_ERR_ASSERT, "GpuZkProver: Ligero parameters differ from the caller's ZkProof");
    }
    detail::TranscriptHook<Field, TranscriptT> hook{&tp, &f_, nullptr};
    const lfgpu_transcript_ops ops = hook.ops();
    uint8_t root[32];
    check(c_.get(), lfgpu_zk_commit(zk_, W.v_.data(), detail::rng_hook<RandomEngineT>, &rng, &ops, root), "lfgpu_zk_commit");
    std::memcpy(zkp.com.root.data, root, 32);
  }

  // ZkProver::prove (zk_prover.h:98-149): false when the witness does not satisfy the circuit
  template <class ZkProofT, class DenseT, class TranscriptT>
  bool prove(ZkProofT& zkp, const DenseT& W, TranscriptT& tp) {
    if (!zk_) check(c_.get(), LFGPU_ERR_ARG, "must run commit before prove");
    detail::TranscriptHook<Field, TranscriptT> hook{&tp, &f_, nullptr};
    const lfgpu_transcript_ops ops = hook.ops();
    int ok = 0;
    check(c_.get(), lfgpu_zk_prove(zk_, W.v_.data(), &ops, &ok), "lfgpu_zk_prove");
    if (!ok) return false;
    size_t n = 0;
    check(c_.get(), lfgpu_zk_proof_write(zk_, nullptr, 0, &n), "lfgpu_zk_proof_write");
    std::vector<uint8_t> wire(n);
    check(c_.get(), lfgpu_zk_proof_write(zk_, wire.data(), wire.size(), &n), "lfgpu_zk_proof_write");
    ReadBufferT rb(wire.data(), n);
    if (!zkp.read(rb, f_)) check(c_.get(), LFGPU_ERR_ASSERT, "GpuZkProver: ZkProof::read rejected the library's wire bytes");
    return true;
  }
  // host milliseconds of the last commit / prove by phase (lfgpu_zk_timings)
  void timings(double ms[6]) const { check(c_.get(), lfgpu_zk_timings(zk_, ms), "lfgpu_zk_timings"); }

 private:
  const Context& c_;
  const Field& f_;
  lfgpu_circuit* circuit_ = nullptr;
  lfgpu_zk_prover* zk_ = nullptr;
  bool have_comm_ = false;
  lfgpu_comm_ops comm_{};
  size_t min_tableau_bytes_ = 0;
};

// Drop-in for ZkVerifier<Field, ReedSolomonFactory> (lib/zk/zk_verifier.h:42-94; call sites lib/circuits/mdoc/mdoc_zk.cc:669-705):
// recv_commitment writes the commitment to the caller's transcript exactly as LigeroTranscript::write_commitment does,
// verify serialises the caller's ZkProof with its own ZkProof::write and hands the wire bytes to lfgpu_zk_verify_committed.
template <class Field>
class GpuZkVerifier {
 public:
  GpuZkVerifier(const Context& ctx, const uint8_t* lfc1, size_t len, size_t rate, size_t nreq, size_t block_enc, const Field& F)
      : c_(ctx), f_(F), rate_(rate), nreq_(nreq), block_enc_(block_enc) {
    check(c_.get(), lfgpu_circuit_from_lfc1(c_.get(), lfc1, len, &circuit_), "lfgpu_circuit_from_lfc1");
  }
  ~GpuZkVerifier() {
    if (circuit_) lfgpu_circuit_free(circuit_);
  }
  GpuZkVerifier(const GpuZkVerifier&) = delete;
  GpuZkVerifier& operator=(const GpuZkVerifier&) = delete;

  template <class ZkProofT, class TranscriptT>
  void recv_commitment(const ZkProofT& zk, TranscriptT& t) const {
    t.write(zk.com.root.data, 32);
  }
  template <class ZkProofT, class DenseT, class TranscriptT>
  bool verify(const ZkProofT& zk, const DenseT& pub, TranscriptT& tv) const {
    std::vector<uint8_t> wire;
    zk.write(wire, f_);
    detail::TranscriptHook<Field, TranscriptT> hook{&tv, &f_, nullptr};
    const lfgpu_transcript_ops ops = hook.ops();
    int ok = 0;
    const char* why = nullptr;
    check(c_.get(),
          lfgpu_zk_verify_committed(c_.get(), circuit_, rate_, nreq_, block_enc_, wire.data(), wire.size(), pub.v_.data(), &ops, &ok, &why),
          "lfgpu_zk_verify_committed");
    if (!ok && why) std::fprintf(stderr, "lfgpu: verify failed: %s\n", why);
    return ok != 0;
  }

 private:
  const Context& c_;
  const Field& f_;
  size_t rate_, nreq_, block_enc_;
  lfgpu_circuit* circuit_ = nullptr;
};

}  // namespace lfgpu
#endif  // LFGPU_ZK_ADAPTERS_H_
