// ref_flatsha.cc -- build-container-only fixture generator (TEST INFRASTRUCTURE, see lf_oracle.h).
//
// Drives the REAL reference (headers + a few .cc files compiled where they lie under
// /root/reference/lib; no reference source is copied) to produce the golden data for BASELINE
// configs[3] "Full sumcheck rounds for BM_ShaZK_fp2_128 circuit":
//   * the flatsha256 GF2_128 circuit for `nb` SHA blocks, serialized with the reference's own
//     CircuitWriter (LFC1 wire format, lib/proto/circuit_writer.h:38-132),
//   * the witness vector of the benchmark message 'a' x len (same inputs as
//     BM_ShaSumcheckProver_fp2_128, lib/circuits/sha/flatsha256_circuit_test.cc:470-488),
//   * the sumcheck proof of run_prover() (lib/sumcheck/testing.h:37-56, transcript "testing", no pad):
//     per layer, per round, per hand the transmitted evaluations p(0), p(2), then wc[0], wc[1].
// Usage: gen_flatsha <nb> <out_prefix>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "arrays/dense.h"
#include "circuits/compiler/compiler.h"
#include "circuits/logic/bit_plucker.h"
#include "circuits/logic/bit_plucker_encoder.h"
#include "circuits/logic/compiler_backend.h"
#include "circuits/logic/logic.h"
#include "circuits/sha/flatsha256_circuit.h"
#include "circuits/sha/flatsha256_witness.h"
#include "circuits/sha/sha256_test_values.h"
#include "gf2k/gf2_128.h"
#include "algebra/convolution.h"
#include "algebra/fp_p128.h"
#include "algebra/reed_solomon.h"
#include "proto/circuit_io.h"
#include "proto/circuit_writer.h"
#include "random/transcript.h"
#include "sumcheck/circuit.h"
#include "sumcheck/prover.h"
#include "gf2k/lch14_reed_solomon.h"
#include "random/random.h"
#include "util/log.h"
#include "zk/zk_proof.h"
#include "zk/zk_prover.h"
#include "zk/zk_verifier.h"

using namespace proofs;

// the LCG of rust/runtime/ligero/tests/ligero.rs:28-43 (the engine the reference's own C++ fixtures use)
class LcgRng : public RandomEngine {
 public:
  explicit LcgRng(uint64_t seed) : s_(seed) {}
  void bytes(uint8_t* buf, size_t n) override {
    for (size_t i = 0; i < n; ++i) {
      s_ = s_ * 6364136223846793005ull + 1442695040888963407ull;
      buf[i] = static_cast<uint8_t>(s_ >> 32);
    }
  }

 private:
  uint64_t s_;
};
#ifdef REF_FP128  // same generator over the prime field Fp128 (lib/algebra/fp_p128.h), as lib/zk/zk_test.cc:252-330 sets it up
using F128 = Fp128<>;
static const FieldID kFieldId = FP128_ID;
using FftConv = FFTConvolutionFactory<F128>;
using RSFactory = ReedSolomonFactory<F128, FftConv>;
#else
using F128 = GF2_128<>;
static const FieldID kFieldId = GF2_128_ID;
using RSFactory = LCH14ReedSolomonFactory<F128>;
#endif
constexpr size_t kPlucker = 2;

static void dump(const std::string& path, const void* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, 1, n, f) != n) {
    fprintf(stderr, "cannot write %s\n", path.c_str());
    exit(1);
  }
  fclose(f);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const size_t nb = strtoul(argv[1], nullptr, 10);
  const std::string prefix = argv[2];
  set_log_level(ERROR);
  const F128 Fs;

  // ---- circuit (one copy), as the benchmark builds it
  using Backend = CompilerBackend<F128>;
  using L = Logic<F128, Backend>;
  using Sha = FlatSHA256Circuit<L, BitPlucker<L, kPlucker>>;
  QuadCircuit<F128> Q(Fs);
  const Backend cbk(&Q);
  const L lc(&cbk, Fs);
  Sha sha(lc);
  auto nbv = lc.template vinput<8>();
  std::vector<typename L::v8> in(64 * nb);
  for (auto& x : in) x = lc.template vinput<8>();
  auto target = lc.template vinput<256>();
  std::vector<typename Sha::BlockWitness> bw(nb);
  for (auto& b : bw) b.input(lc);
  sha.assert_message_hash(nb, nbv, in.data(), target, bw.data());
  std::unique_ptr<Circuit<F128>> C = Q.mkcircuit(1);
  // optional: declare the first argv[3] inputs public and inputs below argv[4] known to lie in the subfield (the SHA
  // circuit's first 1 + 8 + 512*nb + 256 inputs are the constant one and bits), as the mdoc hash circuit does; the
  // gates are unchanged.  Exercises the public-input binding, subfield sampling and the run-length wire encoding.
  if (argc > 3) C->npub_in = strtoul(argv[3], nullptr, 10);
  if (argc > 4) C->subfield_boundary = strtoul(argv[4], nullptr, 10);
  check(C->npub_in <= C->ninputs && C->subfield_boundary <= 1 + 8 + 512 * nb + 256, "override out of range");

  // ---- witness for the message 'a' x len
  const size_t nbench = sizeof(kSha_benchmark_) / sizeof(kSha_benchmark_[0]);
  size_t bi = nb - 1;
  if (bi > nbench) bi = nbench - 1;
  std::vector<uint8_t> msg(kSha_benchmark_[bi].len, 'a');
  uint8_t numb;
  std::vector<uint8_t> inb(64 * nb);
  std::vector<FlatSHA256Witness::BlockWitness> bwb(nb);
  FlatSHA256Witness::transform_and_witness_message(msg.size(), msg.data(), nb, numb, inb.data(), bwb.data());
  const uint8_t* hash = kSha_benchmark_[bi].hash;
  Dense<F128> W(1, C->ninputs);
  size_t wi = 0;
  auto bit = [&](bool b) { W.v_[wi++] = b ? Fs.one() : Fs.zero(); };
  W.v_[wi++] = Fs.one();
  for (size_t i = 0; i < 8; ++i) bit((numb >> i) & 1);
  for (size_t j = 0; j < nb * 64; ++j)
    for (size_t i = 0; i < 8; ++i) bit((inb[j] >> i) & 1);
  for (size_t j = 0; j < 256; ++j) bit((hash[(255 - j) / 8] >> (j % 8)) & 1);
  BitPluckerEncoder<F128, kPlucker> enc(Fs);
  auto pushv = [&](uint32_t v) {
    auto a = enc.mkpacked_v32(v);
    for (size_t i = 0; i < a.size(); ++i) W.v_[wi++] = a[i];
  };
  for (size_t j = 0; j < nb; ++j) {
    for (size_t k = 0; k < 48; ++k) pushv(bwb[j].outw[k]);
    for (size_t k = 0; k < 64; ++k) {
      pushv(bwb[j].oute[k]);
      pushv(bwb[j].outa[k]);
    }
    for (size_t k = 0; k < 8; ++k) pushv(bwb[j].h1[k]);
  }
  check(wi == C->ninputs, "witness size");

  // ---- serialize
  std::vector<uint8_t> bytes;
  CircuitWriter<F128> cw(Fs, kFieldId);
  cw.to_bytes(*C, bytes);
  dump(prefix + ".lfc1", bytes.data(), bytes.size());
  dump(prefix + ".w", W.v_.data(), 16 * C->ninputs);

  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  // ---- reference sumcheck prover (run_prover, transcript "testing")
  Proof<F128> proof(C->nl);
  typename Prover<F128>::inputs pin;
  Prover<F128> prover(Fs);
  auto t0 = std::chrono::steady_clock::now();
  auto V = prover.eval_circuit(&pin, C.get(), W.clone(), Fs);
  auto t1 = std::chrono::steady_clock::now();
  check(V != nullptr, "eval_circuit failed");
  for (size_t i = 0; i < V->n1_; ++i) check(V->v_[i] == Fs.zero(), "non-zero output");
  Transcript ts((const uint8_t*)"testing", 7);
  prover.prove(&proof, nullptr, C.get(), pin, ts);
  auto t2 = std::chrono::steady_clock::now();
  std::vector<uint8_t> pr;
  auto put = [&](const F128::Elt& e) {
    uint8_t b[16];
    Fs.to_bytes_field(b, e);
    pr.insert(pr.end(), b, b + 16);
  };
  for (size_t ly = 0; ly < C->nl; ++ly) {
    for (size_t r = 0; r < C->l[ly].logw; ++r)
      for (size_t h = 0; h < 2; ++h) {
        put(proof.l[ly].hp[h][r].t_[0]);
        put(proof.l[ly].hp[h][r].t_[2]);
      }
    put(proof.l[ly].wc[0]);
    put(proof.l[ly].wc[1]);
  }
  dump(prefix + ".scproof", pr.data(), pr.size());
  size_t nterms = 0, rounds = 0;
  for (auto& ly : C->l) {
    nterms += ly.nterms();
    rounds += ly.logw;
  }
  // ---- full ZK proof (BM_ShaZK_fp2_128 body, flatsha256_circuit_test.cc:510-536) under a deterministic RNG:
  // rate 7, 132 queries, transcript "test", LCG seed 100.  Dumps every component of the proof.
  double zk_commit_ms = 0, zk_prove_ms = 0, zk_verify_ms = 0;
  size_t z_block_enc = 0, z_nrow = 0, z_block = 0, z_dblock = 0, z_nw = 0;
  {
#ifdef REF_FP128
    const FftConv fft(Fs, Fs.of_string("164956748514267535023998284330560247862"), 1ull << 32);
    const RSFactory rsf(fft, Fs);
#else
    const RSFactory rsf(Fs);
#endif
    Transcript tp((const uint8_t*)"test", 4);
    LcgRng rng(100);
    ZkProof<F128> zk(*C, 7, 132);
    ZkProver<F128, RSFactory> zp(*C, Fs, rsf);
    auto z0 = std::chrono::steady_clock::now();
    zp.commit(zk, W, tp, rng);
    auto z1 = std::chrono::steady_clock::now();
    bool ok = zp.prove(zk, W, tp);
    auto z2 = std::chrono::steady_clock::now();
    check(ok, "zk prove failed");
    zk_commit_ms = ms(z0, z1);
    zk_prove_ms = ms(z1, z2);
    const auto& P = zk.param;
    z_block_enc = P.block_enc; z_nrow = P.nrow; z_block = P.block; z_dblock = P.dblock; z_nw = P.nw;
    std::vector<uint8_t> zb;
    auto pute = [&](const F128::Elt& e) {
      uint8_t b[16];
      Fs.to_bytes_field(b, e);
      zb.insert(zb.end(), b, b + 16);
    };
    zb.insert(zb.end(), zk.com.root.data, zk.com.root.data + 32);
    for (size_t ly = 0; ly < C->nl; ++ly) {
      for (size_t r = 0; r < C->l[ly].logw; ++r)
        for (size_t h = 0; h < 2; ++h) {
          pute(zk.proof.l[ly].hp[h][r].t_[0]);
          pute(zk.proof.l[ly].hp[h][r].t_[2]);
        }
      pute(zk.proof.l[ly].wc[0]);
      pute(zk.proof.l[ly].wc[1]);
    }
    for (auto& e : zk.com_proof.y_ldt) pute(e);
    for (auto& e : zk.com_proof.y_dot) pute(e);
    for (auto& e : zk.com_proof.y_quad_0) pute(e);
    for (auto& e : zk.com_proof.y_quad_2) pute(e);
    for (auto& e : zk.com_proof.req) pute(e);
    for (auto& nn : zk.com_proof.merkle.nonce) zb.insert(zb.end(), nn.bytes, nn.bytes + 32);
    uint64_t np = zk.com_proof.merkle.path.size();
    for (int i = 0; i < 8; ++i) zb.push_back(static_cast<uint8_t>(np >> (8 * i)));
    for (auto& d : zk.com_proof.merkle.path) zb.insert(zb.end(), d.data, d.data + 32);
    dump(prefix + ".zkproof", zb.data(), zb.size());
    std::vector<uint8_t> wire;  // the reference's own serialization (ZkProof::write, zk_proof.h:90-185)
    zk.write(wire, Fs);
    dump(prefix + ".zkwire", wire.data(), wire.size());
    // the reference verifier on its own proof (timing baseline for lfgpu_zk_verify)
    {
      ZkVerifier<F128, RSFactory> zv(*C, rsf, 7, 132, Fs);
      Transcript tv((const uint8_t*)"test", 4);
      auto v0 = std::chrono::steady_clock::now();
      zv.recv_commitment(zk, tv);
      bool vok = zv.verify(zk, W, tv);
      auto v1 = std::chrono::steady_clock::now();
      check(vok, "reference verifier rejected the reference proof");
      zk_verify_ms = ms(v0, v1);
    }
  }
  printf("{\"nb\": %zu, \"nl\": %zu, \"ninputs\": %zu, \"npub_in\": %zu, \"nterms\": %zu, \"round_hands\": %zu, \"lfc1_bytes\": %zu, "
         "\"ref_eval_circuit_ms\": %.2f, \"ref_sumcheck_ms\": %.2f, \"zk_nw\": %zu, \"zk_block_enc\": %zu, \"zk_block\": %zu, "
         "\"zk_dblock\": %zu, \"zk_nrow\": %zu, \"ref_zk_commit_ms\": %.2f, \"ref_zk_prove_ms\": %.2f, \"ref_zk_verify_ms\": %.2f}\n",
         nb, (size_t)C->nl, (size_t)C->ninputs, (size_t)C->npub_in, nterms, 2 * rounds, bytes.size(), ms(t0, t1), ms(t1, t2),
         z_nw, z_block_enc, z_block, z_dblock, z_nrow, zk_commit_ms, zk_prove_ms, zk_verify_ms);
  return 0;
}
