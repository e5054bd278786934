// ref_zk_adapters.cc -- the swap INTEGRATION.md describes, EXECUTED: the reference's own ZkProver / LigeroProver /
// sumcheck prover / transcript / serializer (all compiled from /root/reference/lib where they lie), with ONE template
// argument replaced -- InterpolatorFactory = lfgpu::GpuReedSolomonFactory<Field> from include/lfgpu_adapters.h, so every
// Reed-Solomon row extension inside ZkProver::commit and LigeroProver::prove runs in the HIP kernels of liblfgpu.so.
// Reads a circuit in the reference's LFC1 format and a witness (the fixtures under tests/golden), proves with the
// fixtures' RandomEngine (LCG seed 100) and transcript ("test"), serializes with ZkProof::write and prints the SHA-256 of
// the wire bytes -- which tests/test_reference_integration.py compares with the SHA-256 the unmodified reference produced.
// Built by oracle/Makefile (target _ref/zk_adapters[_fp]) in the build container; runs on the GPU box.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "algebra/fp_p128.h"
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

#ifdef REF_FP128
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
    fprintf(stderr, "usage: %s circuit.lfc1 witness.bin\n", argv[0]);
    return 2;
  }
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
  if (wb.size() != 16 * C->ninputs) {
    fprintf(stderr, "witness has %zu bytes, circuit wants %zu inputs\n", wb.size(), C->ninputs);
    return 3;
  }
  Dense<F128> W(1, C->ninputs);
  memcpy(W.v_.data(), wb.data(), wb.size());  // the fixture holds in-memory Elt images

  lfgpu::Context ctx(0);
#ifdef REF_FP128
  const F128::Elt omega = Fs.of_string("164956748514267535023998284330560247862");
  const GpuFactory rsf(ctx, &omega, 1ull << 32);
#else
  const GpuFactory rsf(ctx);
#endif
  Transcript tp((const uint8_t*)"test", 4);
  LcgRng rng(100);
  ZkProof<F128> zk(*C, 7, 132);
  ZkProver<F128, GpuFactory> zp(*C, Fs, rsf);
  zp.commit(zk, W, tp, rng);
  if (!zp.prove(zk, W, tp)) {
    fprintf(stderr, "prove failed\n");
    return 4;
  }
  std::vector<uint8_t> wire;
  zk.write(wire, Fs);
  uint8_t dg[32];
  proofs::SHA256 sha;
  sha.Update(wire.data(), wire.size());
  sha.DigestData(dg);
  printf("{\"wire_bytes\": %zu, \"wire_sha256\": \"", wire.size());
  for (int i = 0; i < 32; ++i) printf("%02x", dg[i]);
  printf("\", \"block_enc\": %zu, \"nrow\": %zu}\n", zk.param.block_enc, zk.param.nrow);
  return 0;
}
