// ref_adapters_exec.cc -- the remaining adapters of include/lfgpu_adapters.h, EXECUTED next to the reference classes they
// stand in for (both compiled into one program: the reference from /root/reference/lib where it lies, the adapters over
// liblfgpu.so), on the reference's own test generator (Bogorng, lib/algebra/bogorng.h:43-51):
//   lfgpu::GpuFFT<Fp128<>>          vs FFT<Fp128<>>::fftb / fftf            lib/algebra/fft.h:185-201
//   lfgpu::GpuLCH14<GF2_128<4>>     vs LCH14<GF2_128<4>>::FFT / IFFT        lib/gf2k/lch14.h:106-144
//   lfgpu::GpuMerkleCommitment      vs MerkleCommitment::commit / open      lib/merkle/merkle_commitment.h:50-73
//                                      with LigeroCommon::column_hash as the leaf callback (lib/ligero/ligero_param.h:432-439)
//   lfgpu::GpuSumcheckRound<Field>  vs ProverLayers::evaluations, Dense::bind, HQuad::bind_h (the round body,
//                                      lib/sumcheck/prover_layers.h:230-263,357-402; lib/arrays/dense.h:70-87; hquad.h:90-123)
// Prints one JSON line {"fft": n_ok, ..., "all_ok": true}; exit code 0 only when every comparison is bit-exact.
// Built by oracle/Makefile (_ref/adapters_exec) in the build container; runs on the GPU box (tests/test_reference_integration.py).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "algebra/bogorng.h"
#include "algebra/fft.h"
#include "algebra/fp_p128.h"
#include "arrays/dense.h"
#include "gf2k/gf2_128.h"
#include "gf2k/lch14.h"
#include "ligero/ligero_param.h"
#include "merkle/merkle_commitment.h"
#include "random/random.h"
#include "sumcheck/hquad.h"
#include "util/log.h"
#define private public  // ProverLayers::evaluations is private
#include "sumcheck/prover_layers.h"
#undef private

#include "lfgpu_adapters.h"

using namespace proofs;
using GF4 = GF2_128<4>;
using FP = Fp128<>;

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

template <class Field>
static std::vector<typename Field::Elt> bogo(const Field& F, size_t n, uint64_t seed) {
  Bogorng<Field> rng(&F);
  std::vector<typename Field::Elt> v(n);
  for (uint64_t i = 0; i < seed % 17; ++i) (void)rng.next();  // different streams per call site
  for (auto& e : v) e = rng.next();
  return v;
}
template <class Elt>
static bool same(const Elt* a, const Elt* b, size_t n) {
  return memcmp(a, b, n * sizeof(Elt)) == 0;
}

// device staging through the C ABI (the sumcheck adapter takes device pointers, as the integration keeps W / QW resident)
struct DevBuf {
  lfgpu_ctx* c;
  void* d = nullptr;
  DevBuf(lfgpu_ctx* ctx, const void* h, size_t bytes) : c(ctx) {
    lfgpu::check(c, lfgpu_malloc(c, bytes ? bytes : 16, &d), "lfgpu_malloc");
    if (h && bytes) lfgpu::check(c, lfgpu_memcpy_h2d(c, d, h, bytes), "h2d");
  }
  ~DevBuf() { lfgpu_free(c, d); }
  void get(void* h, size_t bytes) const { lfgpu::check(c, lfgpu_memcpy_d2h(c, h, d, bytes), "d2h"); }
};

template <class Field>
static int sumcheck_round_cases(const lfgpu::Context& ctx, const Field& F, uint64_t seed) {
  using Elt = typename Field::Elt;
  int ok = 0;
  lfgpu::GpuSumcheckRound<Field> gr(ctx);
  ProverLayers<Field> pl(F);
  for (size_t n : {size_t(1), size_t(2), size_t(7), size_t(64), size_t(1001), size_t(4096), size_t(70001)}) {
    std::vector<Elt> QW = bogo(F, n, seed + n), W = bogo(F, n, seed + 3 * n + 1);
    const Elt sum = bogo(F, 1, seed + 5)[0], r = bogo(F, 1, seed + 11)[0];
    // evaluations: a0, a2 from the device; the polynomial's three values exactly as ProverLayers::evaluations forms them
    DevBuf dQW(ctx.get(), QW.data(), 16 * n), dW(ctx.get(), W.data(), 16 * n);
    Elt a0, a2;
    gr.partials(n, dQW.d, dW.d, a0, a2);
    typename ProverLayers<Field>::WPoly coef, got, want = pl.evaluations(n, F.one(), QW.data(), W.data(), sum, F);
    coef[0] = a0;
    coef[2] = a2;
    coef[1] = sum;
    F.sub(coef[1], coef[0]);
    F.sub(coef[1], coef[0]);
    F.sub(coef[1], coef[2]);
    for (int k = 0; k < 3; ++k) got[k] = coef.eval_monomial(F.poly_evaluation_point(k), F);
    bool good = got[0] == want[0] && got[1] == want[1] && got[2] == want[2];
    // Dense::bind
    Dense<Field> d(n, 1);
    memcpy(&d.v_[0], W.data(), 16 * n);
    d.bind(r, F);
    DevBuf dOut(ctx.get(), nullptr, 16 * ((n + 1) / 2));
    gr.bind_dense(n, r, dW.d, dOut.d);
    std::vector<Elt> bound((n + 1) / 2);
    dOut.get(bound.data(), 16 * bound.size());
    good = good && d.n0_ == bound.size() && same(&d.v_[0], bound.data(), bound.size());
    ok += good ? 1 : 0;
    if (!good) fprintf(stderr, "sumcheck round mismatch at n = %zu\n", n);
  }
  // HQuad::bind_h on a canonical (sorted) corner list with merges, lone even and lone odd entries, both hands
  for (int hand = 0; hand < 2; ++hand) {
    const size_t n = 5000;
    using HQ = HQuad<Field>;
    HQ h(n);
    std::vector<uint32_t> hc(2 * n);
    std::vector<Elt> vc = bogo(F, n, seed + 77 + hand);
    uint32_t a = 0, b = 0;
    for (size_t i = 0; i < n; ++i) {  // strictly increasing in the bound hand's corner, a new other-hand corner every 37 entries
      a += 1 + (uint32_t)((i * 2654435761u >> 7) % 3);
      if (i % 37 == 36) {
        ++b;
        a = (uint32_t)(i % 5);
      }
      hc[2 * i + hand] = a;
      hc[2 * i + 1 - hand] = b;
      h.hc_[i].h[0] = typename HQ::quad_corner_t(hc[2 * i]);
      h.hc_[i].h[1] = typename HQ::quad_corner_t(hc[2 * i + 1]);
      h.vc_[i].v = vc[i];
    }
    const Elt r = bogo(F, 1, seed + 13)[0];
    h.bind_h(r, hand, F);
    DevBuf dhc(ctx.get(), hc.data(), 8 * n), dvc(ctx.get(), vc.data(), 16 * n), ohc(ctx.get(), nullptr, 8 * n), ovc(ctx.get(), nullptr, 16 * n);
    const size_t nout = gr.bind_hquad(n, dhc.d, dvc.d, r, hand, ohc.d, ovc.d);
    std::vector<uint32_t> ghc(2 * n);
    std::vector<Elt> gvc(n);
    ohc.get(ghc.data(), 8 * nout);
    ovc.get(gvc.data(), 16 * nout);
    bool good = nout == h.n_;
    for (size_t i = 0; good && i < nout; ++i)
      good = ghc[2 * i] == (uint32_t)size_t(h.hc_[i].h[0]) && ghc[2 * i + 1] == (uint32_t)size_t(h.hc_[i].h[1]) && gvc[i] == h.vc_[i].v;
    ok += good ? 1 : 0;
    if (!good) fprintf(stderr, "HQuad::bind_h mismatch, hand %d\n", hand);
  }
  return ok;
}

int main() {
  set_log_level(ERROR);
  const GF4 Fg;
  const FP Fp;
  lfgpu::Context ctx(0);
  int fft_ok = 0, fft_n = 0, lch_ok = 0, lch_n = 0, mc_ok = 0, mc_n = 0;

  {  // FFT<Fp128>
    const FP::Elt omega = Fp.of_string("164956748514267535023998284330560247862");
    const uint64_t order = 1ull << 32;
    for (size_t logn : {1, 4, 8, 13, 16, 18}) {
      const size_t n = size_t(1) << logn;
      for (int dir = 0; dir < 2; ++dir, ++fft_n) {
        std::vector<FP::Elt> A = bogo(Fp, n, logn + dir), B = A;
        if (dir == 0) {
          FFT<FP>::fftb(A.data(), n, omega, order, Fp);
          lfgpu::GpuFFT<FP>::fftb(ctx, B.data(), n, omega, order);
        } else {
          FFT<FP>::fftf(A.data(), n, omega, order, Fp);
          lfgpu::GpuFFT<FP>::fftf(ctx, B.data(), n, omega, order);
        }
        fft_ok += same(A.data(), B.data(), n) ? 1 : 0;
      }
    }
  }
  {  // LCH14<GF2_128<4>>
    LCH14<GF4> ref(Fg);
    lfgpu::GpuLCH14<GF4> gpu(ctx);
    for (size_t l : {1, 5, 10, 13, 16}) {
      for (size_t coset : {size_t(0), size_t(3)}) {
        if (((coset + 1) << l) > (size_t(1) << 16)) continue;  // the evaluation domain is the 2^16-element subfield
        for (int dir = 0; dir < 2; ++dir, ++lch_n) {
          const size_t n = size_t(1) << l;
          std::vector<GF4::Elt> A = bogo(Fg, n, l + 2 * coset + dir), B = A;
          if (dir == 0) {
            ref.FFT(l, coset, A.data());
            gpu.FFT(l, coset, B.data());
          } else {
            ref.IFFT(l, coset, A.data());
            gpu.IFFT(l, coset, B.data());
          }
          lch_ok += same(A.data(), B.data(), n) ? 1 : 0;
        }
      }
    }
  }
  {  // MerkleCommitment over the columns of a tableau
    struct Shape {
      size_t nrow, ld, col0, n;
    };
    for (const Shape s : {Shape{20, 512, 91, 300}, Shape{150, 8192, 1819, 6373}, Shape{3, 64, 0, 1}}) {
      ++mc_n;
      std::vector<GF4::Elt> T = bogo(Fg, s.nrow * s.ld, s.n);
      LcgRng r1(7), r2(7);
      MerkleCommitment mc(s.n);
      const Digest root = mc.commit([&](size_t j, proofs::SHA256& sha) { LigeroCommon<GF4>::column_hash(s.nrow, &T[j + s.col0], s.ld, sha, Fg); }, r1);
      lfgpu::GpuMerkleCommitment gmc(s.n, ctx);
      const lfgpu::Digest32 groot = gmc.commit(LFGPU_FIELD_GF2_128, s.nrow, s.ld, s.col0, T.data(), r2);
      bool good = memcmp(root.data, groot.data, 32) == 0;
      std::vector<size_t> pos;
      for (size_t i = 0; i < s.n && pos.size() < 9; i += 1 + s.n / 7) pos.push_back((i * 5 + 3) % s.n);
      std::sort(pos.begin(), pos.end());
      pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
      MerkleProof mp(pos.size());
      mc.open(mp, pos.data(), pos.size());
      std::vector<lfgpu::Digest32> gn, gp;
      gmc.open(gn, gp, pos.data(), pos.size());
      good = good && gn.size() == pos.size() && gp.size() == mp.path.size();
      for (size_t i = 0; good && i < gn.size(); ++i) good = memcmp(gn[i].data, mp.nonce[i].bytes, 32) == 0;
      for (size_t i = 0; good && i < gp.size(); ++i) good = memcmp(gp[i].data, mp.path[i].data, 32) == 0;
      mc_ok += good ? 1 : 0;
    }
  }
  const int sc_g = sumcheck_round_cases(ctx, Fg, 1), sc_f = sumcheck_round_cases(ctx, Fp, 2), sc_n = 9;
  const bool all = fft_ok == fft_n && lch_ok == lch_n && mc_ok == mc_n && sc_g == sc_n && sc_f == sc_n;
  printf("{\"fft\": [%d, %d], \"lch14\": [%d, %d], \"merkle_commitment\": [%d, %d], \"sumcheck_round_gf2128\": [%d, %d], \"sumcheck_round_fp128\": [%d, %d], \"all_ok\": %s}\n",
         fft_ok, fft_n, lch_ok, lch_n, mc_ok, mc_n, sc_g, sc_n, sc_f, sc_n, all ? "true" : "false");
  return all ? 0 : 1;
}
