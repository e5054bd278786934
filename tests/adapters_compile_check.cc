// adapters_compile_check.cc -- build-container-only check that include/lfgpu_adapters.h
// satisfies the reference's template seams: the reference's own LigeroProver is instantiated
// with lfgpu::GpuReedSolomonFactory as its InterpolatorFactory (the swap shown in
// INTEGRATION.md), and include/lfgpu_zk_adapters.h (GpuZkProver / GpuZkVerifier) is instantiated with the reference's
// ZkProof, Dense, Transcript, RandomEngine and ReadBuffer for GF2_128 and Fp256Base.  Compiled with `g++ -c` against
// /root/reference/lib; never run (the executed versions are oracle/ref_zk_adapters.cc and oracle/ref_mdoc_gpu.cc).
#include "algebra/fp_p128.h"
#include "gf2k/gf2_128.h"
#include "ligero/ligero_param.h"
#include "ligero/ligero_prover.h"
#include "random/random.h"
#include "random/transcript.h"

#include "algebra/fp.h"
#include "algebra/fp2.h"
#include "algebra/fp_p256.h"
#include "arrays/dense.h"
#include "sumcheck/circuit.h"
#include "util/readbuffer.h"
#include "zk/zk_proof.h"

#include "lfgpu_adapters.h"
#include "lfgpu_zk_adapters.h"

namespace {
using GF = proofs::GF2_128<>;
using FP = proofs::Fp128<>;

template <class Field>
void commit_with_gpu_factory(const Field& F, const lfgpu::Context& ctx, const typename Field::Elt* W,
                             proofs::RandomEngine& rng) {
  using Factory = lfgpu::GpuReedSolomonFactory<Field>;
  proofs::LigeroParam<Field> p(1000, 0, 4, 36, 4096);
  proofs::LigeroProver<Field, Factory> prover(p);
  proofs::LigeroCommitment<Field> com;
  proofs::Transcript ts((const uint8_t*)"test", 4);
  Factory rsf(ctx);
  prover.commit(com, ts, W, 0, nullptr, rsf, rng, F);
  proofs::LigeroProof<Field> proof(&p);
  proofs::LigeroHash h{};
  prover.prove(proof, ts, 0, 0, nullptr, h, nullptr, rsf, F);
}
}  // namespace

// the prover-level drop-ins (include/lfgpu_zk_adapters.h) with the reference's ZkProof / Dense / Transcript / RandomEngine
template <class Field>
bool zk_with_gpu_prover_and_verifier(const Field& F, const lfgpu::Context& ctx, const proofs::Circuit<Field>& c, const uint8_t* lfc1, size_t len,
                                     proofs::RandomEngine& rng) {
  proofs::ZkProof<Field> zkp(c, 7, 132);
  proofs::Dense<Field> W(1, c.ninputs);
  proofs::Transcript tp((const uint8_t*)"test", 4), tv((const uint8_t*)"test", 4);
  lfgpu::GpuZkProver<Field, proofs::ReadBuffer> prover(ctx, lfc1, len, F);
  prover.commit(zkp, W, tp, rng);
  if (!prover.prove(zkp, W, tp)) return false;
  lfgpu::GpuZkVerifier<Field> verifier(ctx, lfc1, len, 7, 132, 0, F);
  verifier.recv_commitment(zkp, tv);
  return verifier.verify(zkp, W, tv);
}

void lfgpu_adapters_compile_check(const lfgpu::Context& ctx, proofs::RandomEngine& rng) {
  {
    static const proofs::Fp256<true> p256;  // Fp256Base
    const proofs::Circuit<GF>* cg = nullptr;
    const proofs::Circuit<proofs::Fp256<true>>* cp = nullptr;
    if (cg) (void)zk_with_gpu_prover_and_verifier<GF>(GF(), ctx, *cg, nullptr, 0, rng);
    if (cp) (void)zk_with_gpu_prover_and_verifier<proofs::Fp256<true>>(p256, ctx, *cp, nullptr, 0, rng);
    lfgpu::GpuReedSolomonFactory<proofs::Fp256<true>> rs256(ctx);
    (void)rs256.make(455, 4096);
  }
  static const GF gf;
  static const FP fp;
  commit_with_gpu_factory<GF>(gf, ctx, nullptr, rng);
  commit_with_gpu_factory<FP>(fp, ctx, nullptr, rng);
  lfgpu::GpuLCH14<GF> lch(ctx);
  lch.FFT(3, 0, nullptr);
  lfgpu::GpuFFT<FP>::fftb(ctx, nullptr, 8, fp.one(), 8);
  {  // the F64_2 of lib/algebra/fft_test.cc:205-229
    using F64 = proofs::Fp<1>;
    using F64_2 = proofs::Fp2<F64>;
    static_assert(lfgpu::detail::is_f64_2<F64_2>::value && !lfgpu::detail::is_f64_2<FP>::value, "field detection");
    static const F64 f64("18446744069414584321");
    static const F64_2 f64_2(f64);
    lfgpu::GpuFFT<F64_2>::fftf(ctx, nullptr, 8, f64_2.one(), 8);
  }
  lfgpu::GpuSumcheckRound<GF> sc(ctx);
  GF::Elt a0, a2;
  sc.partials(0, nullptr, nullptr, a0, a2);
  lfgpu::GpuMerkleCommitment mc(5, ctx);
  (void)mc;
}
