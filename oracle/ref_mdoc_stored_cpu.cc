// ref_mdoc_stored_cpu.cc -- CPU only, the reference against its own stored data: do the reference's C++ provers
// (ZkProver<f_128, RSFactory> / ZkProver<Fp256Base, RSFactory_b>) inside run_mdoc_prover's flow (lib/circuits/mdoc/mdoc_zk.cc:494-538)
// reproduce the mdoc proof strings the reference keeps under rust/applications/mdoc_zk/artifacts/proofs/ ?  They do, for all twelve
// specs, which is what makes those files known-answer vectors for the library (oracle/ref_mdoc_gpu.cc `stored`,
// tests/test_reference_integration.py).  Inputs recovered from the reference's prior_zk.rs test and prover.rs:
//   document mdoc_tests[3]; attributes = the first num_attributes of {family_name, birth_date, issue_date, height};
//   RandomEngine = DeterministicRng(42) (rust/runtime/random/src/deterministic.rs:18-41), its first 96 bytes spent by
//   generate_mac_ap (mac.rs:18-22) before the hash commit; witnesses = the stored input vectors (public inputs included);
//   MACs = the first 96 bytes of the stored proof (public); circuits = generate_circuit(zk_spec), or the compressed pair under
//   artifacts/circuits/<hash> for the versions the C++ generator no longer builds.
//   usage: mdoc_stored_cpu <index in kZkSpecs> [reference root]
// Test infrastructure (built by oracle/Makefile from the reference sources where they lie; never part of the product).
#include <cstdio>
#include <string>
#include <vector>

#include "circuits/mdoc/mdoc_zk.cc"

#include "circuits/mdoc/mdoc_examples.h"
#include "circuits/mdoc/mdoc_test_attributes.h"

using namespace proofs;

class RustDeterministicRng : public RandomEngine {
 public:
  explicit RustDeterministicRng(uint64_t s) : s_(s) {}
  void bytes(uint8_t* buf, size_t n) override {
    for (size_t i = 0; i < n; ++i) {
      s_ = s_ * 6364136223846793005ull + 1ull;
      buf[i] = static_cast<uint8_t>(s_ >> 56);
    }
  }

 private:
  uint64_t s_;
};
static std::vector<uint8_t> slurp(const std::string& p) {
  std::vector<uint8_t> v;
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return v;
  v.resize(size_t(1) << 23);
  v.resize(fread(v.data(), 1, v.size(), f));
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  set_log_level(ERROR);
  if (argc < 2) return 2;
  const int spec = atoi(argv[1]);
  if (spec < 0 || spec >= (int)kNumZkSpecs) return 2;
  const std::string ref = argc > 2 ? argv[2] : "/root/reference";
  const ZkSpecStruct* zk_spec = &kZkSpecs[spec];
  const std::string art = ref + "/rust/applications/mdoc_zk/artifacts/", base = art + "proofs/" + zk_spec->circuit_hash;
  const std::vector<uint8_t> stored = slurp(base + ".bin"), wh = slurp(base + "_hash_witness.bin"), wsg = slurp(base + "_sig_witness.bin");
  if (stored.size() < 96 || wh.empty() || wsg.empty()) return 2;
  uint8_t* bcp = nullptr;
  size_t bcsz = 0;
  std::vector<uint8_t> cfile;
  bool generated = true;
  if (generate_circuit(zk_spec, &bcp, &bcsz) != CIRCUIT_GENERATION_SUCCESS) {
    generated = false;
    cfile = slurp(art + "circuits/" + zk_spec->circuit_hash);
    if (cfile.empty()) return 3;
    bcp = cfile.data();
    bcsz = cfile.size();
  }
  const MdocTests* test = &mdoc_tests[3];
  const RequestedAttribute attrs[4] = {test::familyname_mustermann, test::birthdate_1971_09_01, test::issue_date_2024_03_15, test::height_175};
  const size_t attrs_len = zk_spec->num_attributes;
  Elt pkX, pkY;
  if (!parsePk(test->pkx.as_pointer, test->pky.as_pointer, pkX, pkY)) return 4;
  const f2_p256 p256_2(p256_base);
  const f_128 Fs;
  std::vector<uint8_t> bytes(kCircuitSizeMax);
  const size_t full_size = decompress(bytes, bcp, bcsz);
  if (full_size == 0) return 5;
  ReadBuffer rb(bytes.data(), full_size);
  CircuitReader<Fp256Base> cr_s(p256_base, P256_ID);
  auto c_sig = cr_s.from_bytes(rb, false);
  CircuitReader<f_128> cr_h(Fs, GF2_128_ID);
  auto c_hash = cr_h.from_bytes(rb, false);
  if (!c_sig || !c_hash) return 6;
  if (wh.size() != f_128::kBytes * c_hash->ninputs || wsg.size() != Fp256Base::kBytes * c_sig->ninputs) return 7;
  auto W_sig = Dense<Fp256Base>(1, c_sig->ninputs);
  auto W_hash = Dense<f_128>(1, c_hash->ninputs);
  for (size_t i = 0; i < c_hash->ninputs; ++i) {
    auto e = Fs.of_bytes_field(&wh[f_128::kBytes * i]);
    if (!e.has_value()) return 8;
    W_hash.v_[i] = e.value();
  }
  for (size_t i = 0; i < c_sig->ninputs; ++i) {
    auto e = p256_base.of_bytes_field(&wsg[Fp256Base::kBytes * i]);
    if (!e.has_value()) return 8;
    W_sig.v_[i] = e.value();
  }
  gf2k macs[6];
  for (size_t i = 0; i < 6; ++i) macs[i] = Fs.of_bytes_field(&stored[f_128::kBytes * i]).value();
  const Elt2 omega = p256_2.of_string(kRootX, kRootY);
  const FftExtConvolutionFactory fft_b(p256_base, p256_2, omega, 1ull << 31);
  const RSFactory_b rsf_b(fft_b, p256_base);
  const RSFactory rsf(Fs);
  ZkProver<f_128, RSFactory> hash_p(*c_hash, Fs, rsf);
  ZkProver<Fp256Base, RSFactory_b> sig_p(*c_sig, p256_base, rsf_b);
  Transcript tp(test->transcript, test->transcript_size, zk_spec->version);
  RustDeterministicRng rng(42);
  uint8_t mac_ap[96];
  rng.bytes(mac_ap, sizeof(mac_ap));
  const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
  const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
  ZkProof<f_128> h_zk(*c_hash, r, req, zk_spec->block_enc_hash);
  ZkProof<Fp256Base> sig_zk(*c_sig, r, req, zk_spec->block_enc_sig);
  hash_p.commit(h_zk, W_hash, tp, rng);
  sig_p.commit(sig_zk, W_sig, tp, rng);
  gf2k av = generate_mac_key(tp);
  size_t pub_diff = 0;
  {  // the public inputs as the verifier derives them: the stored vectors must already hold exactly these
    auto pub_hash = Dense<f_128>(1, c_hash->npub_in);
    auto pub_sig = Dense<Fp256Base>(1, c_sig->npub_in);
    DenseFiller<f_128> hf(pub_hash);
    DenseFiller<Fp256Base> sf(pub_sig);
    if (!fill_public_inputs(sf, hf, pkX, pkY, test->transcript, test->transcript_size, attrs, attrs_len, (const uint8_t*)test->now, (const uint8_t*)test->doc_type,
                            strlen(test->doc_type), macs, av, Fs, zk_spec->version))
      return 10;
    if (hf.size() != c_hash->npub_in || sf.size() != c_sig->npub_in) return 11;
    for (size_t i = 0; i < c_hash->npub_in; ++i) pub_diff += !(W_hash.v_[i] == pub_hash.v_[i]);
    for (size_t i = 0; i < c_sig->npub_in; ++i) pub_diff += !(W_sig.v_[i] == pub_sig.v_[i]);
    for (size_t i = 0; i < c_hash->npub_in; ++i) W_hash.v_[i] = pub_hash.v_[i];
    for (size_t i = 0; i < c_sig->npub_in; ++i) W_sig.v_[i] = pub_sig.v_[i];
  }
  if (!hash_p.prove(h_zk, W_hash, tp)) return 12;
  if (!sig_p.prove(sig_zk, W_sig, tp)) return 13;
  std::vector<uint8_t> buf(stored.begin(), stored.begin() + 96);
  h_zk.write(buf, Fs);
  sig_zk.write(buf, p256_base);
  printf("{\"spec\": %d, \"circuit_hash\": \"%s\", \"version\": %zu, \"attributes\": %zu, \"circuit\": \"%s\", \"stored_bytes\": %zu, \"produced_bytes\": %zu, "
         "\"public_inputs_differing_from_stored\": %zu, \"identical\": %s}\n",
         spec, zk_spec->circuit_hash, (size_t)zk_spec->version, attrs_len, generated ? "generate_circuit" : "artifact file", stored.size(), buf.size(), pub_diff,
         buf == stored ? "true" : "false");
  return buf == stored ? 0 : 1;
}
