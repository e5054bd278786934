// ref_zk_adapters.cc -- the swap INTEGRATION.md describes, EXECUTED: the reference's own ZkProver / LigeroProver /
// sumcheck prover / transcript / serializer (all compiled from /root/reference/lib where they lie), with ONE template
// argument replaced -- InterpolatorFactory = lfgpu::GpuReedSolomonFactory<Field> from include/lfgpu_adapters.h, so every
// Reed-Solomon row extension inside ZkProver::commit and LigeroProver::prove runs in the HIP kernels of liblfgpu.so.
// Reads a circuit in the reference's LFC1 format and a witness (the fixtures under tests/golden), proves with the
// fixtures' RandomEngine (LCG seed 100) and transcript ("test"), serializes with ZkProof::write and prints the SHA-256 of
// the wire bytes -- which tests/test_reference_integration.py compares with the SHA-256 the unmodified reference produced.
// -DREF_FP128: Fp128; -DREF_P256: Fp256Base (the mdoc signature circuit, 32-byte elements).
// Built by oracle/Makefile (targets _ref/zk_adapters, _fp, _p256) in the build container; runs on the GPU box.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "algebra/fp_p128.h"
#include "algebra/fp_p256.h"
#include "arrays/dense.h"
#include "gf2k/gf2_128.h"
#include "proto/circuit_io.h"
#include "proto/circuit_reader.h"
#include "random/random.h"
#include "random/transcript.h"
#include "sumcheck/circuit.h"
#include "util/crypto.h"
#include "util/log.h"
#include "util/readbuffer.h"
#include "zk/zk_proof.h"
#include "zk/zk_prover.h"

#include "lfgpu_adapters.h"

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

#if defined(REF_P256)  // the mdoc signature circuit's field: 32-byte elements (BASELINE config 5)
using F128 = Fp256<true>;  // = Fp256Base (lib/ec/p256.h:42)
static const FieldID kFieldId = P256_ID;
#elif defined(REF_FP128)
using F128 = Fp128<>;
static const FieldID kFieldId = FP128_ID;
#else
using F128 = GF2_128<>;
static const FieldID kFieldId = GF2_128_ID;
#endif
using GpuFactory = lfgpu::GpuReedSolomonFactory<F128>;

static std::vector<uint8_t> slurp(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    fprintf(stderr, "cannot open %s\n", path);
    exit(2);
  }
  std::vector<uint8_t> b;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + n);
  fclose(f);
  return b;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s circuit.lfc1 witness.bin [block_enc]\n", argv[0]);
    return 2;
  }
  const size_t block_enc = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;  // 0: LigeroParam picks it (zk_proof.h:63-68)
  set_log_level(ERROR);
  const F128 Fs;
  std::vector<uint8_t> cb = slurp(argv[1]), wb = slurp(argv[2]);
  CircuitReader<F128> reader(Fs, kFieldId);
  ReadBuffer rb(cb.data(), cb.size());
  std::unique_ptr<Circuit<F128>> C = reader.from_bytes(rb, /*enforce_circuit_id=*/false);
  if (!C) {
    fprintf(stderr, "circuit does not parse\n");
    return 3;
  }
  if (wb.size() != sizeof(F128::Elt) * C->ninputs) {
    fprintf(stderr, "witness has %zu bytes, circuit wants %zu inputs\n", wb.size(), C->ninputs);
    return 3;
  }
  Dense<F128> W(1, C->ninputs);
  memcpy(W.v_.data(), wb.data(), wb.size());  // the fixture holds in-memory Elt images

  lfgpu::Context ctx(0);
#if defined(REF_FP128) && !defined(REF_P256)
  const F128::Elt omega = Fs.of_string("164956748514267535023998284330560247862");
  const GpuFactory rsf(ctx, &omega, 1ull << 32);
#else
  const GpuFactory rsf(ctx);
#endif
  double commit_ms = 0, prove_ms = 0;
  std::unique_ptr<ZkProof<F128>> zkp;
  for (int rep = 0; rep < 2; ++rep) {  // the second repetition is the timed one (device tables and twiddles cached)
    Transcript tp((const uint8_t*)"test", 4);
    LcgRng rng(100);
    zkp = block_enc ? std::make_unique<ZkProof<F128>>(*C, 7, 132, block_enc) : std::make_unique<ZkProof<F128>>(*C, 7, 132);
    ZkProver<F128, GpuFactory> zp(*C, Fs, rsf);
    auto t0 = std::chrono::steady_clock::now();
    zp.commit(*zkp, W, tp, rng);
    auto t1 = std::chrono::steady_clock::now();
    if (!zp.prove(*zkp, W, tp)) {
      fprintf(stderr, "prove failed\n");
      return 4;
    }
    auto t2 = std::chrono::steady_clock::now();
    commit_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    prove_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
  }
  ZkProof<F128>& zk = *zkp;
  std::vector<uint8_t> wire;
  zk.write(wire, Fs);
  uint8_t dg[32];
  proofs::SHA256 sha;
  sha.Update(wire.data(), wire.size());
  sha.DigestData(dg);
  printf("{\"wire_bytes\": %zu, \"wire_sha256\": \"", wire.size());
  for (int i = 0; i < 32; ++i) printf("%02x", dg[i]);
  printf("\", \"block_enc\": %zu, \"nrow\": %zu, \"commit_ms\": %.2f, \"prove_ms\": %.2f}\n", zk.param.block_enc, zk.param.nrow, commit_ms, prove_ms);
  return 0;
}
