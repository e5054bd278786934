// zk_throughput.cc -- throughput mode from C++: K host threads, each with its own lfgpu context
// (lfgpu_own_stream), prover and transcripts, ONE copy of the circuit in HBM (lfgpu_circuit_share); every proof is checked by
// the verifier.  The reference's loop is one proof after the other on one core (BM_ShaZK_fp2_128,
// lib/circuits/sha/flatsha256_circuit_test.cc:510-536); on the GPU a single proof is latency-bound, K of them fill the device.
//
//   g++ -std=c++17 -O2 -pthread -Iinclude examples/zk_throughput.cc -Llongfellow-zk_amd -llfgpu -Wl,-rpath,$PWD/longfellow-zk_amd -o zk_throughput
//   GPU_MAX_HW_QUEUES=16 ./zk_throughput circuit.lfc1 witness.bin K seconds
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <thread>
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
static void rng_bytes(void* user, uint8_t* buf, size_t n) { lfgpu_transcript_bytes((lfgpu_transcript*)user, buf, n); }

struct Worker {
  lfgpu_ctx* ctx = nullptr;
  lfgpu_circuit* circ = nullptr;
  lfgpu_zk_prover* zk = nullptr;
  lfgpu_transcript* rng = nullptr;
  long proofs = 0;
  int failed = 0;
  char err[600] = {0};
};

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s circuit.lfc1 witness.bin K seconds\n", argv[0]);
    return 2;
  }
  setenv("GPU_MAX_HW_QUEUES", "16", 0);  // before the first HIP call: HIP's default of 4 hardware queues serialises more streams
  const std::vector<uint8_t> lfc1 = slurp(argv[1]), wit = slurp(argv[2]);
  const int K = atoi(argv[3]);
  const double seconds = atof(argv[4]);
  lfgpu_ctx* base = nullptr;
  if (lfgpu_init(0, &base) != LFGPU_OK) {
    fprintf(stderr, "no MI355X / HIP device: there is no CPU fallback\n");
    return 1;
  }
  lfgpu_circuit* circ = nullptr;
  if (lfgpu_circuit_from_lfc1(base, lfc1.data(), lfc1.size(), &circ) != LFGPU_OK) {
    fprintf(stderr, "circuit: %s\n", lfgpu_last_error(base));
    return 1;
  }
  std::vector<Worker> ws(K);
  for (int i = 0; i < K; ++i) {  // set-up is sequential (lfgpu_circuit_share wants the source idle)
    Worker& w = ws[i];
    if (lfgpu_init(0, &w.ctx) != LFGPU_OK || lfgpu_own_stream(w.ctx) != LFGPU_OK || lfgpu_circuit_share(w.ctx, circ, &w.circ) != LFGPU_OK ||
        lfgpu_zk_prover_new(w.ctx, w.circ, 7, 132, 0, &w.zk) != LFGPU_OK) {
      fprintf(stderr, "worker %d: %s\n", i, w.ctx ? lfgpu_last_error(w.ctx) : "lfgpu_init");
      return 1;
    }
    char seed[32];
    const int n = snprintf(seed, sizeof(seed), "rng seed %d", i);
    w.rng = lfgpu_transcript_new((const uint8_t*)seed, (size_t)n);
  }
  std::atomic<bool> go{false}, stop{false};
  auto job = [&](Worker& w) {
    std::vector<uint8_t> proof;
    while (!go.load()) std::this_thread::yield();
    while (!stop.load()) {
      lfgpu_transcript* tp = lfgpu_transcript_new((const uint8_t*)"test", 4);
      lfgpu_transcript_ops ops;
      lfgpu_transcript_get_ops(tp, &ops);
      uint8_t root[32];
      int ok = 0;
      size_t n = 0;
      int rc = lfgpu_zk_commit(w.zk, wit.data(), rng_bytes, w.rng, &ops, root);
      if (rc == LFGPU_OK) rc = lfgpu_zk_prove(w.zk, wit.data(), &ops, &ok);
      if (rc == LFGPU_OK && ok) rc = lfgpu_zk_proof_write(w.zk, nullptr, 0, &n);
      lfgpu_transcript_free(tp);
      if (rc == LFGPU_OK && ok) {
        proof.resize(n);
        rc = lfgpu_zk_proof_write(w.zk, proof.data(), proof.size(), &n);
      }
      if (rc == LFGPU_OK && ok && (w.proofs & 7) == 0) {  // every eighth proof of a worker goes through the verifier
        lfgpu_transcript* tv = lfgpu_transcript_new((const uint8_t*)"test", 4);
        lfgpu_transcript_get_ops(tv, &ops);
        const char* why = "";
        int acc = 0;
        rc = lfgpu_zk_verify(w.ctx, w.circ, 7, 132, 0, proof.data(), n, wit.data(), &ops, &acc, &why);
        lfgpu_transcript_free(tv);
        if (rc == LFGPU_OK && !acc) ok = 0;
      }
      if (rc != LFGPU_OK || !ok) {
        w.failed = 1;
        snprintf(w.err, sizeof(w.err), "rc %d ok %d: %s", rc, ok, lfgpu_last_error(w.ctx));
        return;
      }
      ++w.proofs;
    }
  };
  for (Worker& w : ws) {  // warm-up: the first proof of a handle records its per-circuit caches
    lfgpu_transcript* tp = lfgpu_transcript_new((const uint8_t*)"test", 4);
    lfgpu_transcript_ops ops;
    lfgpu_transcript_get_ops(tp, &ops);
    uint8_t root[32];
    int ok = 0;
    const bool good = lfgpu_zk_commit(w.zk, wit.data(), rng_bytes, w.rng, &ops, root) == LFGPU_OK && lfgpu_zk_prove(w.zk, wit.data(), &ops, &ok) == LFGPU_OK && ok;
    lfgpu_transcript_free(tp);
    if (!good) {
      fprintf(stderr, "warm-up proof failed: %s\n", lfgpu_last_error(w.ctx));
      return 1;
    }
  }
  go.store(false);
  std::vector<std::thread> th;
  for (Worker& w : ws) th.emplace_back(job, std::ref(w));
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true);
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& t : th) t.join();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long total = 0;
  int failed = 0;
  for (Worker& w : ws) {
    total += w.proofs;
    if (w.failed) {
      failed = 1;
      fprintf(stderr, "worker failed: %s\n", w.err);
    }
  }
  printf("{\"K\": %d, \"proofs\": %ld, \"wall_s\": %.3f, \"proofs_per_s\": %.2f, \"all_verified_samples_accepted\": %s}\n", K, total, wall, total / wall, failed ? "false" : "true");
  for (Worker& w : ws) {
    lfgpu_transcript_free(w.rng);
    lfgpu_zk_prover_free(w.zk);
    lfgpu_circuit_free(w.circ);
    lfgpu_shutdown(w.ctx);
  }
  lfgpu_circuit_free(circ);
  lfgpu_shutdown(base);
  return failed;
}
