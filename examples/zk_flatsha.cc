// zk_flatsha.cc -- the prover-level C ABI (include/lfgpu_zk.h) driven from C++, the reference's host language:
// load an LFC1 circuit (CircuitRep::to_bytes output) and a witness, commit + prove + serialise, then verify the wire
// bytes, and print the timings.  No Python, no torch: this is what a C++ caller of the reference's ZkProver /
// ZkVerifier (lib/zk/zk_prover.h, zk_verifier.h) links against.
//
//   g++ -std=c++17 -O2 -Iinclude examples/zk_flatsha.cc -Llongfellow-zk_amd -llfgpu -Wl,-rpath,$PWD/longfellow-zk_amd -o zk_flatsha
//   ./zk_flatsha circuit.lfc1 witness.bin [reps]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <vector>

#include "lfgpu_zk.h"

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    fprintf(stderr, "cannot read %s\n", path);
    exit(2);
  }
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
#define CK(ctx, call)                                                          \
  do {                                                                         \
    int rc_ = (call);                                                          \
    if (rc_ != LFGPU_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lfgpu_last_error(ctx));    \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// RandomEngine for the prover: the built-in AES-CTR PRF keyed by a seed transcript (a real deployment passes its CSPRNG)
static void rng_bytes(void* user, uint8_t* buf, size_t n) { lfgpu_transcript_bytes((lfgpu_transcript*)user, buf, n); }

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s circuit.lfc1 witness.bin [reps]\n", argv[0]);
    return 2;
  }
  const std::vector<uint8_t> lfc1 = slurp(argv[1]), wit = slurp(argv[2]);
  const int reps = argc > 3 ? atoi(argv[3]) : 3;
  lfgpu_ctx* ctx = nullptr;
  if (lfgpu_init(0, &ctx) != LFGPU_OK) {
    fprintf(stderr, "no MI355X / HIP device: there is no CPU fallback\n");
    return 1;
  }
  lfgpu_circuit* circ = nullptr;
  CK(ctx, lfgpu_circuit_from_lfc1(ctx, lfc1.data(), lfc1.size(), &circ));
  lfgpu_circuit_info info;
  CK(ctx, lfgpu_circuit_get_info(circ, &info));
  if (wit.size() != info.ninputs * 16) {
    fprintf(stderr, "witness has %zu bytes, the circuit wants %zu inputs x 16\n", wit.size(), info.ninputs);
    return 2;
  }
  lfgpu_zk_prover* zk = nullptr;
  CK(ctx, lfgpu_zk_prover_new(ctx, circ, /*rateinv=*/7, /*nreq=*/132, /*block_enc=*/0, &zk));
  lfgpu_transcript* rng = lfgpu_transcript_new((const uint8_t*)"rng seed", 8);
  std::vector<uint8_t> proof;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  double best_prove = 1e30, best_verify = 1e30;
  for (int r = 0; r < reps; ++r) {
    lfgpu_transcript* tp = lfgpu_transcript_new((const uint8_t*)"test", 4);
    lfgpu_transcript_ops ops;
    lfgpu_transcript_get_ops(tp, &ops);
    uint8_t root[32];
    int ok = 0;
    const auto t0 = now();
    CK(ctx, lfgpu_zk_commit(zk, wit.data(), rng_bytes, rng, &ops, root));
    CK(ctx, lfgpu_zk_prove(zk, wit.data(), &ops, &ok));
    const auto t1 = now();
    lfgpu_transcript_free(tp);
    if (!ok) {
      fprintf(stderr, "the witness does not satisfy the circuit\n");
      return 1;
    }
    size_t n = 0;
    CK(ctx, lfgpu_zk_proof_write(zk, nullptr, 0, &n));
    proof.resize(n);
    CK(ctx, lfgpu_zk_proof_write(zk, proof.data(), proof.size(), &n));
    lfgpu_transcript* tv = lfgpu_transcript_new((const uint8_t*)"test", 4);
    lfgpu_transcript_get_ops(tv, &ops);
    const char* why = "";
    const auto t2 = now();
    CK(ctx, lfgpu_zk_verify(ctx, circ, 7, 132, 0, proof.data(), proof.size(), wit.data() /*public inputs come first*/, &ops, &ok, &why));
    const auto t3 = now();
    lfgpu_transcript_free(tv);
    if (!ok) {
      fprintf(stderr, "verifier rejected the proof: %s\n", why);
      return 1;
    }
    if (ms(t0, t1) < best_prove) best_prove = ms(t0, t1);
    if (ms(t2, t3) < best_verify) best_verify = ms(t2, t3);
  }
  double ph[6];
  CK(ctx, lfgpu_zk_timings(zk, ph));
  printf("{\"field\": %d, \"layers\": %zu, \"terms\": %zu, \"inputs\": %zu, \"proof_bytes\": %zu, \"commit_prove_ms\": %.3f, \"verify_ms\": %.3f, "
         "\"last_phases_ms\": {\"commit\": %.3f, \"eval_circuit\": %.3f, \"sumcheck\": %.3f, \"constraints\": %.3f, \"ligero_prove\": %.3f}}\n",
         info.field, info.nl, info.nterms, info.ninputs, proof.size(), best_prove, best_verify, ph[0], ph[2], ph[3], ph[4], ph[5]);
  lfgpu_transcript_free(rng);
  lfgpu_zk_prover_free(zk);
  lfgpu_circuit_free(circ);
  lfgpu_shutdown(ctx);
  return 0;
}
