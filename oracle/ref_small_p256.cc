// ref_small_p256.cc -- a SMALL ZK fixture over Fp256Base from the real reference (build container only): the circuit of the
// reference's own zk_test "rfc_sgonal" example (lib/zk/zk_test.cc:250-271: 2 n = (s - 2) m^2 - (s - 4) m, one public input
// n, private m and s), compiled over the P-256 base field instead of Fp128, proved with ZkProver<Fp256Base, .> under the
// fixtures' transcript ("test") and LCG RandomEngine (seed 100), rate 4, 6 queries (the parameters of that test).  Pins the
// library's Fp256Base prover on tiny layers (1-2 variables, HQUADs of a handful of entries), where the mdoc signature
// circuit (2^9 .. 2^16 wires per layer) never goes.  Prints one JSON line: circuit and witness in hex, proof length + SHA-256.
#include <cstdio>
#include <vector>

#include "algebra/convolution.h"
#include "algebra/fp2.h"
#include "algebra/reed_solomon.h"
#include "arrays/dense.h"
#include "circuits/compiler/circuit_dump.h"
#include "circuits/compiler/compiler.h"
#include "circuits/logic/compiler_backend.h"
#include "circuits/logic/logic.h"
#include "ec/p256.h"
#include "proto/circuit_io.h"
#include "proto/circuit_writer.h"
#include "random/random.h"
#include "random/transcript.h"
#include "sumcheck/circuit.h"
#include "util/crypto.h"
#include "util/log.h"
#include "zk/zk_proof.h"
#include "zk/zk_prover.h"
#include "zk/zk_verifier.h"

using namespace proofs;

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

static void hex(const char* key, const uint8_t* p, size_t n, bool last = false) {
  printf("\"%s\": \"", key);
  for (size_t i = 0; i < n; ++i) printf("%02x", p[i]);
  printf("\"%s", last ? "" : ", ");
}

int main() {
  set_log_level(ERROR);
  using CompilerBackend = CompilerBackend<Fp256Base>;
  using LogicCircuit = Logic<Fp256Base, CompilerBackend>;
  using EltW = LogicCircuit::EltW;
  const Fp256Base& F = p256_base;
  std::unique_ptr<Circuit<Fp256Base>> circuit;
  {
    QuadCircuit<Fp256Base> Q(F);
    CompilerBackend cbk(&Q);
    const LogicCircuit LC(&cbk, F);
    EltW n = LC.eltw_input();
    Q.private_input();
    EltW m = LC.eltw_input();
    EltW s = LC.eltw_input();
    LC.assert_eq(LC.sub(LC.mul(LC.sub(s, LC.konst(2)), LC.mul(m, m)), LC.mul(LC.sub(s, LC.konst(4)), m)), LC.mul(n, LC.konst(2)));
    circuit = Q.mkcircuit(1);
  }
  std::vector<uint8_t> cb;
  CircuitWriter<Fp256Base> cw(F, P256_ID);
  cw.to_bytes(*circuit, cb);
  auto W = Dense<Fp256Base>(1, circuit->ninputs);
  DenseFiller<Fp256Base> filler(W);
  filler.push_back(F.one());
  filler.push_back(F.of_scalar(45));
  filler.push_back(F.of_scalar(5));
  filler.push_back(F.of_scalar(6));
  if (filler.size() != circuit->ninputs) return 3;

  using f2_p256 = Fp2<Fp256Base>;
  using FftExtConvolutionFactory = FFTExtConvolutionFactory<Fp256Base, f2_p256>;
  using RSFactory_b = ReedSolomonFactory<Fp256Base, FftExtConvolutionFactory>;
  const f2_p256 p256_2(F);
  // the root of unity of order 2^31 in the quadratic extension (lib/circuits/mdoc/mdoc_zk.cc:82-88)
  const auto omega = p256_2.of_string("112649224146410281873500457609690258373018840430489408729223714171582664680802",
                                      "84087994358540907695740461427818660560182168997182378749313018254450460212908");
  const FftExtConvolutionFactory fft_b(F, p256_2, omega, 1ull << 31);
  const RSFactory_b rsf(fft_b, F);
  ZkProof<Fp256Base> zk(*circuit, 4, 6);
  ZkProver<Fp256Base, RSFactory_b> zp(*circuit, F, rsf);
  Transcript tp((const uint8_t*)"test", 4);
  LcgRng rng(100);
  zp.commit(zk, W, tp, rng);
  if (!zp.prove(zk, W, tp)) return 4;
  std::vector<uint8_t> wire;
  zk.write(wire, F);
  // the reference's verifier on it
  ZkVerifier<Fp256Base, RSFactory_b> zv(*circuit, rsf, 4, 6, F);
  Transcript tv((const uint8_t*)"test", 4);
  zv.recv_commitment(zk, tv);
  auto pub = Dense<Fp256Base>(1, circuit->npub_in);
  for (size_t i = 0; i < circuit->npub_in; ++i) pub.v_[i] = W.v_[i];
  const bool vok = zv.verify(zk, pub, tv);
  uint8_t dg[32];
  proofs::SHA256 sha;
  sha.Update(wire.data(), wire.size());
  sha.DigestData(dg);
  printf("{");
  hex("lfc1", cb.data(), cb.size());
  hex("witness", (const uint8_t*)W.v_.data(), 32 * circuit->ninputs);
  hex("zk_wire", wire.data(), wire.size());
  hex("zk_wire_sha256", dg, 32);
  printf("\"nl\": %zu, \"ninputs\": %zu, \"npub_in\": %zu, \"nv\": %zu, \"block_enc\": %zu, \"nrow\": %zu, \"block\": %zu, \"rate\": 4, \"nreq\": 6, "
         "\"reference_verifier_accepts\": %s}\n",
         circuit->nl, circuit->ninputs, circuit->npub_in, circuit->nv, zk.param.block_enc, zk.param.nrow, zk.param.block, vok ? "true" : "false");
  return vok ? 0 : 5;
}
