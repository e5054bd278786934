// adapters_compile_check.cc -- build-container-only check that include/lfgpu_adapters.h
// satisfies the reference's template seams: the reference's own LigeroProver is instantiated
// with lfgpu::GpuReedSolomonFactory as its InterpolatorFactory (the swap shown in
// INTEGRATION.md).  Compiled with `g++ -c` against /root/reference/lib; never run.
#include "algebra/fp_p128.h"
#include "gf2k/gf2_128.h"
#include "ligero/ligero_param.h"
#include "ligero/ligero_prover.h"
#include "random/random.h"
#include "random/transcript.h"

#include "lfgpu_adapters.h"

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

void lfgpu_adapters_compile_check(const lfgpu::Context& ctx, proofs::RandomEngine& rng) {
  static const GF gf;
  static const FP fp;
  commit_with_gpu_factory<GF>(gf, ctx, nullptr, rng);
  commit_with_gpu_factory<FP>(fp, ctx, nullptr, rng);
  lfgpu::GpuLCH14<GF> lch(ctx);
  lch.FFT(3, 0, nullptr);
  lfgpu::GpuFFT<FP>::fftb(ctx, nullptr, 8, fp.one(), 8);
  lfgpu::GpuSumcheckRound<GF> sc(ctx);
  GF::Elt a0, a2;
  sc.partials(0, nullptr, nullptr, a0, a2);
  lfgpu::GpuMerkleCommitment mc(5, ctx);
  (void)mc;
}
