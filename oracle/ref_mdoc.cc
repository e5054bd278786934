// ref_mdoc.cc -- BASELINE config 5 fixture generator (build container only).  Repeats the body of run_mdoc_prover
// (lib/circuits/mdoc/mdoc_zk.cc:398-546: circuit parse, fill_witness, both commits, MAC key, update_macs, both proves)
// with a DETERMINISTIC RandomEngine on the reference's own example (kZkSpecs[0], mdoc_tests[0], age_over_18 --
// lib/circuits/mdoc/mdoc_zk_test.cc:652-685), so that the witnesses are the ones a real proof uses, and then proves the
// HASH circuit (GF2_128; public inputs, subfield boundary, block_enc 4151) stand-alone with ZkProver under the fixtures'
// transcript ("test") and LCG engine.  Dumps: the hash circuit in LFC1, its final witness, length + SHA-256 of the
// stand-alone proof's wire bytes; the same for the signature circuit (Fp256Base, 32-byte elements), proved stand-alone with the
// reference's ReedSolomonFactory<Fp256Base, FFTExtConvolutionFactory> (mdoc_zk.cc:75-76).
// The reference's mdoc_zk.cc is compiled where it lies by including it (its helpers have no header).
#include <chrono>
#include <cstdio>
#include <string>

#include "circuits/mdoc/mdoc_zk.cc"

#include "circuits/mdoc/mdoc_examples.h"
#include "circuits/mdoc/mdoc_test_attributes.h"
#include "proto/circuit_writer.h"

namespace proofs {
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

static void dump(const std::string& path, const void* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, 1, n, f) != n) {
    fprintf(stderr, "cannot write %s\n", path.c_str());
    exit(1);
  }
  fclose(f);
}

int mdoc_fixture(const std::string& prefix) {
  set_log_level(ERROR);
  const ZkSpecStruct* zk_spec = &kZkSpecs[0];
  uint8_t* bcp;
  size_t bcsz;
  if (generate_circuit(zk_spec, &bcp, &bcsz) != CIRCUIT_GENERATION_SUCCESS) return 3;
  const MdocTests* test = &mdoc_tests[0];
  const RequestedAttribute attrs[] = {test::age_over_18};
  const size_t attrs_len = 1;
  Elt pkX, pkY;
  if (!parsePk(test->pkx.as_pointer, test->pky.as_pointer, pkX, pkY)) return 4;

  const f2_p256 p256_2(p256_base);
  const f_128 Fs;
  std::unique_ptr<Circuit<Fp256Base>> c_sig;
  std::unique_ptr<Circuit<f_128>> c_hash;
  {
    std::vector<uint8_t> bytes(kCircuitSizeMax);
    size_t full_size = decompress(bytes, bcp, bcsz);
    if (full_size == 0) return 5;
    ReadBuffer rb(bytes.data(), full_size);
    CircuitReader<Fp256Base> cr_s(p256_base, P256_ID);
    c_sig = cr_s.from_bytes(rb, false);
    CircuitReader<f_128> cr_h(Fs, GF2_128_ID);
    c_hash = cr_h.from_bytes(rb, false);
    if (!c_sig || !c_hash) return 6;
  }
  auto W_sig = Dense<Fp256Base>(1, c_sig->ninputs);
  auto W_hash = Dense<f_128>(1, c_hash->ninputs);
  DenseFiller<Fp256Base> sig_filler(W_sig);
  DenseFiller<f_128> hash_filler(W_hash);
  // fill_witness takes the concrete SecureRandomEngine (it draws the MAC key shares): the witness of this fixture is
  // therefore one valid witness among many, and is stored; everything after it is deterministic
  SecureRandomEngine srng;
  LcgRng rng(42);
  ProverState state;
  if (fill_witness(sig_filler, hash_filler, test->mdoc, test->mdoc_size, pkX, pkY, test->transcript, test->transcript_size, attrs, attrs_len,
                   (const uint8_t*)test->now, state, srng, Fs, zk_spec->version) != MDOC_PROVER_SUCCESS)
    return 7;

  Transcript tp(test->transcript, test->transcript_size, zk_spec->version);
  const Elt2 omega = p256_2.of_string(kRootX, kRootY);
  const FftExtConvolutionFactory fft_b(p256_base, p256_2, omega, 1ull << 31);
  const RSFactory_b rsf_b(fft_b, p256_base);
  const RSFactory rsf(Fs);
  const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
  const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
  ZkProof<f_128> h_zk(*c_hash, r, req, zk_spec->block_enc_hash);
  ZkProof<Fp256Base> sig_zk(*c_sig, r, req, zk_spec->block_enc_sig);
  ZkProver<f_128, RSFactory> hash_p(*c_hash, Fs, rsf);
  ZkProver<Fp256Base, RSFactory_b> sig_p(*c_sig, p256_base, rsf_b);
  hash_p.commit(h_zk, W_hash, tp, rng);
  sig_p.commit(sig_zk, W_sig, tp, rng);
  gf2k av = generate_mac_key(tp), macs[6];
  uint8_t macs_b[6 * f_128::kBytes];
  compute_macs(3, state.common, macs, macs_b, state.ap, av);
  update_macs(W_sig, W_hash, kSigMacIndex, getHashMacIndex(attrs_len, zk_spec->version), macs, av, Fs);
  if (!hash_p.prove(h_zk, W_hash, tp)) return 8;
  if (!sig_p.prove(sig_zk, W_sig, tp)) return 9;

  // ---- the hash circuit alone, with the final witness, under the fixtures' transcript and engine
  std::vector<uint8_t> cb;
  CircuitWriter<f_128> cw(Fs, GF2_128_ID);
  cw.to_bytes(*c_hash, cb);
  dump(prefix + ".hash.lfc1", cb.data(), cb.size());
  dump(prefix + ".hash.w", W_hash.v_.data(), 16 * c_hash->ninputs);
  Transcript ts((const uint8_t*)"test", 4);
  LcgRng rng2(100);
  ZkProof<f_128> hz(*c_hash, r, req, zk_spec->block_enc_hash);
  ZkProver<f_128, RSFactory> hp(*c_hash, Fs, rsf);
  auto t0 = std::chrono::steady_clock::now();
  hp.commit(hz, W_hash, ts, rng2);
  auto t1 = std::chrono::steady_clock::now();
  if (!hp.prove(hz, W_hash, ts)) return 10;
  auto t2 = std::chrono::steady_clock::now();
  std::vector<uint8_t> wire;
  hz.write(wire, Fs);
  dump(prefix + ".hash.zkwire", wire.data(), wire.size());
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  size_t nterms = 0;
  for (auto& ly : c_hash->l) nterms += ly.nterms();

  // ---- the signature circuit (Fp256Base) alone, same convention
  std::vector<uint8_t> sb;
  CircuitWriter<Fp256Base> sw(p256_base, P256_ID);
  sw.to_bytes(*c_sig, sb);
  dump(prefix + ".sig.lfc1", sb.data(), sb.size());
  dump(prefix + ".sig.w", W_sig.v_.data(), 32 * c_sig->ninputs);
  Transcript ts2((const uint8_t*)"test", 4);
  LcgRng rng3(100);
  ZkProof<Fp256Base> sz(*c_sig, r, req, zk_spec->block_enc_sig);
  ZkProver<Fp256Base, RSFactory_b> sp(*c_sig, p256_base, rsf_b);
  auto t3 = std::chrono::steady_clock::now();
  sp.commit(sz, W_sig, ts2, rng3);
  auto t4 = std::chrono::steady_clock::now();
  if (!sp.prove(sz, W_sig, ts2)) return 11;
  auto t5 = std::chrono::steady_clock::now();
  std::vector<uint8_t> swire;
  sz.write(swire, p256_base);
  dump(prefix + ".sig.zkwire", swire.data(), swire.size());
  size_t snterms = 0;
  for (auto& ly : c_sig->l) snterms += ly.nterms();
  printf(
      "{\"spec\": 0, \"hash\": {\"nl\": %zu, \"ninputs\": %zu, \"npub_in\": %zu, \"subfield_boundary\": %zu, \"nterms\": %zu, \"lfc1_bytes\": %zu, "
      "\"block_enc\": %zu, \"nrow\": %zu, \"block\": %zu, \"nw\": %zu, \"rate\": %zu, \"nreq\": %zu, \"ref_commit_ms\": %.2f, \"ref_prove_ms\": %.2f}, "
      "\"sig\": {\"nl\": %zu, \"ninputs\": %zu, \"npub_in\": %zu, \"block_enc\": %zu, \"nrow\": %zu, \"block\": %zu, \"dblock\": %zu, \"nw\": %zu, "
      "\"subfield_boundary\": %zu, \"nterms\": %zu, \"lfc1_bytes\": %zu, \"ref_commit_ms\": %.2f, \"ref_prove_ms\": %.2f}}\n",
      c_hash->nl, c_hash->ninputs, c_hash->npub_in, c_hash->subfield_boundary, nterms, cb.size(), hz.param.block_enc, hz.param.nrow, hz.param.block,
      hz.param.nw, r, req, ms(t0, t1), ms(t1, t2), c_sig->nl, c_sig->ninputs, c_sig->npub_in, sig_zk.param.block_enc, sig_zk.param.nrow,
      sig_zk.param.block, sig_zk.param.dblock, sig_zk.param.nw, c_sig->subfield_boundary, snterms, sb.size(), ms(t3, t4), ms(t4, t5));
  return 0;
}
}  // namespace proofs

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  return proofs::mdoc_fixture(argv[1]);
}
