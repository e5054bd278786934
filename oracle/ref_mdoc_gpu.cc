// ref_mdoc_gpu.cc -- BASELINE config 5 end to end, EXECUTED with the library in the reference's place: the body of
// run_mdoc_prover (lib/circuits/mdoc/mdoc_zk.cc:398-546) on the reference's own example (kZkSpecs[0], mdoc_tests[0],
// age_over_18 -- mdoc_zk_test.cc:652-685) twice over the same witness and the same deterministic RandomEngine:
//   (a) with the reference's ZkProver<f_128, RSFactory> / ZkProver<Fp256Base, RSFactory_b>                (one CPU thread)
//   (b) with lfgpu::GpuZkProver<f_128, ReadBuffer> / lfgpu::GpuZkProver<Fp256Base, ReadBuffer>            (liblfgpu.so)
// and then the body of run_mdoc_verifier (:548-712) with lfgpu::GpuZkVerifier in the place of both ZkVerifiers.
// Everything else -- circuit generation and parsing, CBOR/mdoc witness filling, the shared transcript, the MAC key drawn
// from it between the commits and the proofs, update_macs, ZkProof::write of both proofs into the mdoc proof string -- is
// the reference's code in both runs.  Prints the SHA-256 of the two proof strings (they must be equal), the reference
// verifier's verdict on (b), and wall times.  Built by oracle/Makefile (_ref/mdoc_gpu) in the build container from the
// reference sources where they lie; runs on the GPU box.
#include <chrono>
#include <cstdio>
#include <string>

#include "circuits/mdoc/mdoc_zk.cc"

#include "circuits/mdoc/mdoc_examples.h"
#include "circuits/mdoc/mdoc_test_attributes.h"

#include "lfgpu_zk_adapters.h"

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

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::string sha_hex(const std::vector<uint8_t>& b) {
  uint8_t dg[32];
  SHA256 sha;
  sha.Update(b.data(), b.size());
  sha.DigestData(dg);
  char s[65];
  for (int i = 0; i < 32; ++i) snprintf(s + 2 * i, 3, "%02x", dg[i]);
  return s;
}

// mdoc_zk.cc:494-538 with the two provers as parameters
template <class HashProver, class SigProver>
static bool prove_both(HashProver& hash_p, SigProver& sig_p, const Circuit<f_128>& c_hash, const Circuit<Fp256Base>& c_sig, const Dense<f_128>& W_hash0,
                       const Dense<Fp256Base>& W_sig0, const MdocTests* test, const ZkSpecStruct* zk_spec, const ProverState& state, size_t attrs_len, const f_128& Fs,
                       std::vector<uint8_t>& buf, double ms[2]) {
  auto W_hash_p = W_hash0.clone();  // update_macs writes into the witnesses: every run starts from the same ones
  auto W_sig_p = W_sig0.clone();
  Dense<f_128>& W_hash = *W_hash_p;
  Dense<Fp256Base>& W_sig = *W_sig_p;
  Transcript tp(test->transcript, test->transcript_size, zk_spec->version);
  LcgRng rng(42);
  const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
  const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
  ZkProof<f_128> h_zk(c_hash, r, req, zk_spec->block_enc_hash);
  ZkProof<Fp256Base> sig_zk(c_sig, r, req, zk_spec->block_enc_sig);
  const double t0 = now_ms();
  hash_p.commit(h_zk, W_hash, tp, rng);
  sig_p.commit(sig_zk, W_sig, tp, rng);
  const double t1 = now_ms();
  gf2k av = generate_mac_key(tp), macs[6];
  uint8_t macs_b[6 * f_128::kBytes];
  compute_macs(3, state.common, macs, macs_b, state.ap, av);
  update_macs(W_sig, W_hash, kSigMacIndex, getHashMacIndex(attrs_len, zk_spec->version), macs, av, Fs);
  const double t2 = now_ms();
  if (!hash_p.prove(h_zk, W_hash, tp)) return false;
  if (!sig_p.prove(sig_zk, W_sig, tp)) return false;
  const double t3 = now_ms();
  ms[0] = t1 - t0;
  ms[1] = t3 - t2;
  buf.clear();
  buf.insert(buf.begin(), macs_b, macs_b + 6 * f_128::kBytes);  // [6 mac values] [hash proof] [sig proof]
  h_zk.write(buf, Fs);
  sig_zk.write(buf, p256_base);
  return true;
}

// mdoc_zk.cc:632-712 (proof parsing, both recv_commitment, MAC key, public inputs, both verify) with the two verifiers as
// parameters; returns 1 accept, 0 reject, < 0 parse / input failure
template <class HashVerifier, class SigVerifier>
static int verify_both(HashVerifier& hash_v, SigVerifier& sig_v, const Circuit<f_128>& c_hash, const Circuit<Fp256Base>& c_sig, const std::vector<uint8_t>& zbuf,
                       const MdocTests* test, const RequestedAttribute* attrs, size_t attrs_len, const Elt& pkX, const Elt& pkY, const ZkSpecStruct* zk_spec,
                       const f_128& Fs, double* ms) {
  const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
  const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
  ZkProof<f_128> pr_hash(c_hash, r, req, zk_spec->block_enc_hash);
  ZkProof<Fp256Base> pr_sig(c_sig, r, req, zk_spec->block_enc_sig);
  ReadBuffer rb(zbuf);
  gf2k macs[6];
  for (size_t i = 0; i < 6; ++i) macs[i] = Fs.of_bytes_field(rb.next(f_128::kBytes)).value();
  if (!pr_hash.read(rb, Fs) || !pr_sig.read(rb, p256_base) || rb.remaining() != 0) return -1;
  const double t0 = now_ms();
  class Transcript tv(test->transcript, test->transcript_size, zk_spec->version);
  hash_v.recv_commitment(pr_hash, tv);
  sig_v.recv_commitment(pr_sig, tv);
  gf2k av = generate_mac_key(tv);
  auto pub_hash = Dense<f_128>(1, c_hash.npub_in);
  auto pub_sig = Dense<Fp256Base>(1, c_sig.npub_in);
  DenseFiller<f_128> hash_filler(pub_hash);
  DenseFiller<Fp256Base> sig_filler(pub_sig);
  if (!fill_public_inputs(sig_filler, hash_filler, pkX, pkY, test->transcript, test->transcript_size, attrs, attrs_len, (const uint8_t*)test->now,
                          (const uint8_t*)test->doc_type, strlen(test->doc_type), macs, av, Fs, zk_spec->version))
    return -2;
  if (hash_filler.size() != c_hash.npub_in || sig_filler.size() != c_sig.npub_in) return -3;
  const bool ok = hash_v.verify(pr_hash, pub_hash, tv);
  const bool ok2 = sig_v.verify(pr_sig, pub_sig, tv);
  *ms = now_ms() - t0;
  return ok && ok2 ? 1 : 0;
}

// ---- the reference's own STORED mdoc proofs (rust/applications/mdoc_zk/artifacts/proofs/<circuit hash>{.bin, _hash_witness.bin,
// _sig_witness.bin}; loaded by rust/applications/mdoc_zk/runtime/tests/all/prior_zk.rs:131-145).  The artifact of kZkSpecs[i] is the
// proof that run_mdoc_prover_inner (rust/applications/mdoc_zk/runtime/src/prover.rs:53-215) makes for mdoc_tests[3] and the first
// num_attributes of {family_name, birth_date, issue_date, height} with DeterministicRng(42)
// (rust/runtime/random/src/deterministic.rs:18-41): that engine's first 96 bytes are generate_mac_ap's six u128, the rest feeds
// hash commit, then signature commit.  With the stored witnesses (whole input vectors, public part included) the reference's C++
// provers reproduce the stored 360 596 bytes exactly (checked in the build container); here lfgpu::GpuZkProver must, and
// lfgpu::GpuZkVerifier must accept the stored bytes inside run_mdoc_verifier's body.
class RustDeterministicRng : public RandomEngine {
 public:
  explicit RustDeterministicRng(uint64_t seed) : s_(seed) {}
  void bytes(uint8_t* buf, size_t n) override {
    for (size_t i = 0; i < n; ++i) {
      s_ = s_ * 6364136223846793005ull + 1ull;
      buf[i] = static_cast<uint8_t>(s_ >> 56);
    }
  }

 private:
  uint64_t s_;
};
static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) return v;
  v.resize(size_t(1) << 23);
  v.resize(fread(v.data(), 1, v.size(), f));
  fclose(f);
  return v;
}
template <class HashProver, class SigProver>
static bool prove_stored(HashProver& hash_p, SigProver& sig_p, const Circuit<f_128>& c_hash, const Circuit<Fp256Base>& c_sig, Dense<f_128>& W_hash,
                         Dense<Fp256Base>& W_sig, const MdocTests* test, const RequestedAttribute* attrs, size_t attrs_len, const Elt& pkX, const Elt& pkY,
                         const ZkSpecStruct* zk_spec, const std::vector<uint8_t>& stored, const f_128& Fs, std::vector<uint8_t>& buf, double ms[2]) {
  gf2k macs[6];  // public: the head of the proof string
  for (size_t i = 0; i < 6; ++i) macs[i] = Fs.of_bytes_field(&stored[f_128::kBytes * i]).value();
  Transcript tp(test->transcript, test->transcript_size, zk_spec->version);
  RustDeterministicRng rng(42);
  uint8_t mac_ap[96];
  rng.bytes(mac_ap, sizeof(mac_ap));  // generate_mac_ap (mac.rs:18-22)
  const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
  const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
  ZkProof<f_128> h_zk(c_hash, r, req, zk_spec->block_enc_hash);
  ZkProof<Fp256Base> sig_zk(c_sig, r, req, zk_spec->block_enc_sig);
  const double t0 = now_ms();
  hash_p.commit(h_zk, W_hash, tp, rng);
  sig_p.commit(sig_zk, W_sig, tp, rng);
  const double t1 = now_ms();
  gf2k av = generate_mac_key(tp);
  {  // the public inputs as the verifier builds them (the stored witnesses already hold exactly these)
    auto pub_hash = Dense<f_128>(1, c_hash.npub_in);
    auto pub_sig = Dense<Fp256Base>(1, c_sig.npub_in);
    DenseFiller<f_128> hf(pub_hash);
    DenseFiller<Fp256Base> sf(pub_sig);
    if (!fill_public_inputs(sf, hf, pkX, pkY, test->transcript, test->transcript_size, attrs, attrs_len, (const uint8_t*)test->now, (const uint8_t*)test->doc_type,
                            strlen(test->doc_type), macs, av, Fs, zk_spec->version))
      return false;
    if (hf.size() != c_hash.npub_in || sf.size() != c_sig.npub_in) return false;
    for (size_t i = 0; i < c_hash.npub_in; ++i) W_hash.v_[i] = pub_hash.v_[i];
    for (size_t i = 0; i < c_sig.npub_in; ++i) W_sig.v_[i] = pub_sig.v_[i];
  }
  const double t2 = now_ms();
  if (!hash_p.prove(h_zk, W_hash, tp)) return false;
  if (!sig_p.prove(sig_zk, W_sig, tp)) return false;
  ms[0] = t1 - t0;
  ms[1] = now_ms() - t2;
  buf.assign(stored.begin(), stored.begin() + 6 * f_128::kBytes);
  h_zk.write(buf, Fs);
  sig_zk.write(buf, p256_base);
  return true;
}
// circuit_path: the compressed circuit pair as the reference ships it (rust/applications/mdoc_zk/artifacts/circuits/<hash>) for the
// older specs whose circuits generate_circuit no longer builds; nullptr = generate_circuit(zk_spec)
int mdoc_stored(int spec, const char* proof_path, const char* hash_w_path, const char* sig_w_path, const char* circuit_path, bool with_ref) {
  set_log_level(ERROR);
  if (spec < 0 || spec >= (int)kNumZkSpecs) return 2;
  const ZkSpecStruct* zk_spec = &kZkSpecs[spec];
  const std::vector<uint8_t> stored = slurp(proof_path), wh = slurp(hash_w_path), wsg = slurp(sig_w_path);
  if (stored.size() < 96 || wh.empty() || wsg.empty()) return 2;
  uint8_t* bcp = nullptr;
  size_t bcsz = 0;
  std::vector<uint8_t> cfile;
  if (circuit_path) {
    cfile = slurp(circuit_path);
    if (cfile.empty()) return 3;
    bcp = cfile.data();
    bcsz = cfile.size();
  } else if (generate_circuit(zk_spec, &bcp, &bcsz) != CIRCUIT_GENERATION_SUCCESS) {
    return 3;
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
  const size_t sig_len = full_size - rb.remaining();
  CircuitReader<f_128> cr_h(Fs, GF2_128_ID);
  auto c_hash = cr_h.from_bytes(rb, false);
  const size_t hash_len = full_size - rb.remaining() - sig_len;
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
  const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
  const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
  std::vector<uint8_t> proof_ref, proof_gpu;
  double ms_ref[2] = {0, 0}, ms_gpu[2] = {0, 0}, ms_gpu_verify = 0, dummy;
  if (with_ref) {  // the reference's provers (one CPU thread, ~20 s): the artifact pins the reference itself
    const Elt2 omega = p256_2.of_string(kRootX, kRootY);
    const FftExtConvolutionFactory fft_b(p256_base, p256_2, omega, 1ull << 31);
    const RSFactory_b rsf_b(fft_b, p256_base);
    const RSFactory rsf(Fs);
    ZkProver<f_128, RSFactory> hash_p(*c_hash, Fs, rsf);
    ZkProver<Fp256Base, RSFactory_b> sig_p(*c_sig, p256_base, rsf_b);
    if (!prove_stored(hash_p, sig_p, *c_hash, *c_sig, W_hash, W_sig, test, attrs, attrs_len, pkX, pkY, zk_spec, stored, Fs, proof_ref, ms_ref)) return 9;
  }
  int verdict = -9, verdict_bad = -9;
  {
    lfgpu::Context ctx(0);
    lfgpu::GpuZkProver<Fp256Base, ReadBuffer> sig_p(ctx, bytes.data(), sig_len, p256_base);
    lfgpu::GpuZkProver<f_128, ReadBuffer> hash_p(ctx, bytes.data() + sig_len, hash_len, Fs);
    for (int rep = 0; rep < 2; ++rep)
      if (!prove_stored(hash_p, sig_p, *c_hash, *c_sig, W_hash, W_sig, test, attrs, attrs_len, pkX, pkY, zk_spec, stored, Fs, proof_gpu, ms_gpu)) return 10;
    lfgpu::GpuZkVerifier<Fp256Base> sig_v(ctx, bytes.data(), sig_len, r, req, zk_spec->block_enc_sig, p256_base);
    lfgpu::GpuZkVerifier<f_128> hash_v(ctx, bytes.data() + sig_len, hash_len, r, req, zk_spec->block_enc_hash, Fs);
    verdict = verify_both(hash_v, sig_v, *c_hash, *c_sig, stored, test, attrs, attrs_len, pkX, pkY, zk_spec, Fs, &ms_gpu_verify);
    // one flipped bit in a MAC (a public input of both circuits), in the hash-circuit proof and in the signature-circuit proof:
    // every one must be rejected (a parse failure counts as a rejection)
    verdict_bad = 0;
    for (size_t pos : {size_t(40), stored.size() / 2, stored.size() - 60000}) {
      std::vector<uint8_t> bad = stored;
      bad[pos] ^= 1;
      if (verify_both(hash_v, sig_v, *c_hash, *c_sig, bad, test, attrs, attrs_len, pkX, pkY, zk_spec, Fs, &dummy) == 1) verdict_bad = 1;
    }
  }
  printf(
      "{\"stored_artifact\": \"%s\", \"spec\": %d, \"version\": %zu, \"attributes\": %zu, \"stored_bytes\": %zu, \"stored_sha256\": \"%s\", \"gpu_bytes\": %zu, "
      "\"gpu_proof_identical_to_stored\": %s, \"reference_prover_ran\": %s, \"reference_proof_identical_to_stored\": %s, "
      "\"gpu_verifiers_accept_stored\": %s, \"gpu_verifiers_reject_flipped_bit\": %s, "
      "\"gpu_ms\": {\"commit\": %.2f, \"prove\": %.2f, \"verify\": %.2f}, \"ref_ms\": {\"commit\": %.2f, \"prove\": %.2f}}\n",
      zk_spec->circuit_hash, spec, (size_t)zk_spec->version, attrs_len, stored.size(), sha_hex(stored).c_str(), proof_gpu.size(), proof_gpu == stored ? "true" : "false",
      with_ref ? "true" : "false", with_ref && proof_ref == stored ? "true" : "false", verdict == 1 ? "true" : "false", verdict_bad == 0 ? "true" : "false", ms_gpu[0], ms_gpu[1],
      ms_gpu_verify, ms_ref[0], ms_ref[1]);
  if (!circuit_path) free(bcp);
  if (proof_gpu != stored) return 11;
  if (with_ref && proof_ref != stored) return 12;
  return verdict == 1 && verdict_bad == 0 ? 0 : 13;
}

// which of the reference's own examples (mdoc_zk_test.cc:118-240): 0 = kZkSpecs[0], mdoc_tests[0], age_over_18 (the BASELINE
// config); 1 = kZkSpecs[0], mdoc_tests[3], familyname_mustermann (another document, a text attribute); 2 = kZkSpecs[1] -- the
// TWO-attribute circuits, a different pair of circuits -- mdoc_tests[3], age_over_18 + familyname_mustermann
int mdoc_gpu(int reps, bool with_ref, int which) {
  set_log_level(ERROR);
  const ZkSpecStruct* zk_spec = &kZkSpecs[which == 2 ? 1 : 0];
  uint8_t* bcp;
  size_t bcsz;
  const double tg0 = now_ms();
  if (generate_circuit(zk_spec, &bcp, &bcsz) != CIRCUIT_GENERATION_SUCCESS) return 3;
  const double tg1 = now_ms();
  const MdocTests* test = &mdoc_tests[which == 0 ? 0 : 3];
  const RequestedAttribute attrs_all[3][2] = {{test::age_over_18, test::age_over_18},
                                              {test::familyname_mustermann, test::familyname_mustermann},
                                              {test::age_over_18, test::familyname_mustermann}};
  const RequestedAttribute* attrs = attrs_all[which];
  const size_t attrs_len = which == 2 ? 2 : 1;
  Elt pkX, pkY;
  if (!parsePk(test->pkx.as_pointer, test->pky.as_pointer, pkX, pkY)) return 4;
  const f2_p256 p256_2(p256_base);
  const f_128 Fs;
  std::unique_ptr<Circuit<Fp256Base>> c_sig;
  std::unique_ptr<Circuit<f_128>> c_hash;
  std::vector<uint8_t> bytes(kCircuitSizeMax);
  const size_t full_size = decompress(bytes, bcp, bcsz);
  if (full_size == 0) return 5;
  size_t sig_len, hash_len;
  double t_parse_ref;
  {
    const double t0 = now_ms();
    ReadBuffer rb(bytes.data(), full_size);
    CircuitReader<Fp256Base> cr_s(p256_base, P256_ID);
    c_sig = cr_s.from_bytes(rb, false);
    sig_len = full_size - rb.remaining();
    CircuitReader<f_128> cr_h(Fs, GF2_128_ID);
    c_hash = cr_h.from_bytes(rb, false);
    hash_len = full_size - rb.remaining() - sig_len;
    if (!c_sig || !c_hash) return 6;
    t_parse_ref = now_ms() - t0;
  }
  auto W_sig = Dense<Fp256Base>(1, c_sig->ninputs);
  auto W_hash = Dense<f_128>(1, c_hash->ninputs);
  DenseFiller<Fp256Base> sig_filler(W_sig);
  DenseFiller<f_128> hash_filler(W_hash);
  SecureRandomEngine srng;  // fill_witness takes this concrete type (MAC key shares): both runs below share the witness it makes
  ProverState state;
  const double tw0 = now_ms();
  if (fill_witness(sig_filler, hash_filler, test->mdoc, test->mdoc_size, pkX, pkY, test->transcript, test->transcript_size, attrs, attrs_len,
                   (const uint8_t*)test->now, state, srng, Fs, zk_spec->version) != MDOC_PROVER_SUCCESS)
    return 7;
  const double t_witness = now_ms() - tw0;

  // (a) the reference's provers
  std::vector<uint8_t> proof_ref;
  double ms_ref[2] = {0, 0};
  if (with_ref) {
    const Elt2 omega = p256_2.of_string(kRootX, kRootY);
    const FftExtConvolutionFactory fft_b(p256_base, p256_2, omega, 1ull << 31);
    const RSFactory_b rsf_b(fft_b, p256_base);
    const RSFactory rsf(Fs);
    ZkProver<f_128, RSFactory> hash_p(*c_hash, Fs, rsf);
    ZkProver<Fp256Base, RSFactory_b> sig_p(*c_sig, p256_base, rsf_b);
    if (!prove_both(hash_p, sig_p, *c_hash, *c_sig, W_hash, W_sig, test, zk_spec, state, attrs_len, Fs, proof_ref, ms_ref)) return 8;
  }

  // (b) the library's provers behind the same two calls; then the library's verifiers behind ZkVerifier's calls
  std::vector<uint8_t> proof_gpu;
  double ms_gpu[2] = {0, 0}, t_upload, ms_gpu_verify = 0, ms_ref_verify_body = 0;
  int gpu_verdict = -9, gpu_verdict_bad = -9, ref_verdict = -9;
  {
    lfgpu::Context ctx(0);
    const double t0 = now_ms();
    lfgpu::GpuZkProver<Fp256Base, ReadBuffer> sig_p(ctx, bytes.data(), sig_len, p256_base);
    lfgpu::GpuZkProver<f_128, ReadBuffer> hash_p(ctx, bytes.data() + sig_len, hash_len, Fs);
    t_upload = now_ms() - t0;
    for (int rep = 0; rep < reps; ++rep)  // the last repetition is the one reported (tables, twiddles and bind structure cached)
      if (!prove_both(hash_p, sig_p, *c_hash, *c_sig, W_hash, W_sig, test, zk_spec, state, attrs_len, Fs, proof_gpu, ms_gpu)) return 9;
    const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
    const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
    lfgpu::GpuZkVerifier<Fp256Base> sig_v(ctx, bytes.data(), sig_len, r, req, zk_spec->block_enc_sig, p256_base);
    lfgpu::GpuZkVerifier<f_128> hash_v(ctx, bytes.data() + sig_len, hash_len, r, req, zk_spec->block_enc_hash, Fs);
    for (int rep = 0; rep < reps; ++rep)
      gpu_verdict = verify_both(hash_v, sig_v, *c_hash, *c_sig, proof_gpu, test, attrs, attrs_len, pkX, pkY, zk_spec, Fs, &ms_gpu_verify);
    std::vector<uint8_t> bad = proof_gpu;  // one flipped bit in the signature proof's sumcheck section
    bad[bad.size() - 150000] ^= 1;
    double dummy;
    gpu_verdict_bad = verify_both(hash_v, sig_v, *c_hash, *c_sig, bad, test, attrs, attrs_len, pkX, pkY, zk_spec, Fs, &dummy);
  }
  {  // the same verifier body with the reference's ZkVerifiers (timing beside it)
    const Elt2 omega = p256_2.of_string(kRootX, kRootY);
    const FftExtConvolutionFactory fft_b(p256_base, p256_2, omega, 1ull << 31);
    const RSFactory_b rsf_b(fft_b, p256_base);
    const RSFactory rsf(Fs);
    const size_t r = zk_spec->version < 7 ? kLigeroRate : kLigeroRatev7;
    const size_t req = zk_spec->version < 7 ? kLigeroNreq : kLigeroNreqv7;
    ZkVerifier<f_128, RSFactory> hash_v(*c_hash, rsf, r, req, zk_spec->block_enc_hash, Fs);
    ZkVerifier<Fp256Base, RSFactory_b> sig_v(*c_sig, rsf_b, r, req, zk_spec->block_enc_sig, p256_base);
    ref_verdict = verify_both(hash_v, sig_v, *c_hash, *c_sig, proof_gpu, test, attrs, attrs_len, pkX, pkY, zk_spec, Fs, &ms_ref_verify_body);
  }

  // the reference's verifier on the library's proof string
  const double tv0 = now_ms();
  const MdocVerifierErrorCode vr =
      run_mdoc_verifier(bcp, bcsz, test->pkx.as_pointer, test->pky.as_pointer, test->transcript, test->transcript_size, attrs, attrs_len,
                        (const char*)test->now, proof_gpu.data(), proof_gpu.size(), test->doc_type, zk_spec);
  const double t_verify = now_ms() - tv0;
  printf(
      "{\"case\": %d, \"attributes\": %zu, \"proof_bytes\": %zu, \"gpu_sha256\": \"%s\", \"ref_sha256\": \"%s\", \"identical\": %s, \"reference_verifier_accepts_gpu_proof\": %s, "
      "\"gpu_ms\": {\"commit\": %.2f, \"prove\": %.2f, \"total\": %.2f}, \"ref_ms\": {\"commit\": %.2f, \"prove\": %.2f, \"total\": %.2f}, "
      "\"host_ms\": {\"generate_circuit\": %.1f, \"reference_parse\": %.1f, \"fill_witness\": %.1f, \"gpu_parse_upload\": %.1f, \"reference_verify\": %.1f}, "
      "\"verify\": {\"gpu_verifiers_accept\": %s, \"gpu_verifiers_reject_flipped_bit\": %s, \"reference_verifiers_accept\": %s, \"gpu_ms\": %.2f, \"ref_ms\": %.2f}, "
      "\"circuit_bytes\": {\"sig\": %zu, \"hash\": %zu}}\n",
      which, attrs_len, proof_gpu.size(), sha_hex(proof_gpu).c_str(), with_ref ? sha_hex(proof_ref).c_str() : "", with_ref && proof_ref == proof_gpu ? "true" : "false",
      vr == MDOC_VERIFIER_SUCCESS ? "true" : "false", ms_gpu[0], ms_gpu[1], ms_gpu[0] + ms_gpu[1], ms_ref[0], ms_ref[1], ms_ref[0] + ms_ref[1], tg1 - tg0,
      t_parse_ref, t_witness, t_upload, t_verify, gpu_verdict == 1 ? "true" : "false", gpu_verdict_bad == 0 ? "true" : "false",
      ref_verdict == 1 ? "true" : "false", ms_gpu_verify, ms_ref_verify_body, sig_len, hash_len);
  free(bcp);
  if (with_ref && proof_ref != proof_gpu) return 10;
  if (gpu_verdict != 1 || gpu_verdict_bad != 0 || ref_verdict != 1) return 12;
  return vr == MDOC_VERIFIER_SUCCESS ? 0 : 11;
}
}  // namespace proofs

int main(int argc, char** argv) {
  if (argc >= 6 && std::string(argv[1]) == "stored") {  // stored <spec> <proof> <hash witness> <sig witness> [--circuit <file>] [--with-ref]
    const char* circuit = nullptr;
    bool with_ref = false;
    for (int i = 6; i < argc; ++i) {
      if (std::string(argv[i]) == "--with-ref") with_ref = true;
      if (std::string(argv[i]) == "--circuit" && i + 1 < argc) circuit = argv[++i];
    }
    return proofs::mdoc_stored(atoi(argv[2]), argv[3], argv[4], argv[5], circuit, with_ref);
  }
  const int reps = argc > 1 ? atoi(argv[1]) : 2;
  const bool with_ref = !(argc > 2 && std::string(argv[2]) == "--no-ref");
  const int which = argc > 3 ? atoi(argv[3]) : 0;
  if (which < 0 || which > 2) return 2;
  return proofs::mdoc_gpu(reps < 1 ? 1 : reps, with_ref, which);
}
