// lfgpu_adapters.h -- header-only C++17 adapters that present the reference's template
// seams on top of the C ABI (include/lfgpu.h).  A reference maintainer includes this
// header, links liblfgpu.so and swaps ONE template argument / member; the callers
// (ZkProver, LigeroProver, ProverLayers) keep their signatures.
//
//   reference seam (file:line)                                   adapter here
//   ------------------------------------------------------------ --------------------------
//   InterpolatorFactory concept: make(n,m)->interpolate(Elt*)    lfgpu::GpuReedSolomonFactory<Field>
//     LCH14ReedSolomonFactory  lib/gf2k/lch14_reed_solomon.h:112-123
//     ReedSolomonFactory       lib/algebra/reed_solomon.h:133-147
//     used by LigeroProver     lib/ligero/ligero_prover.h:34,175,184,210,237,295
//   MerkleCommitment::commit/open  lib/merkle/merkle_commitment.h:50-73   lfgpu::GpuMerkleCommitment
//   FFT<Field>::fftb / fftf   lib/algebra/fft.h:185-201              lfgpu::GpuFFT<Field>
//   LCH14<Field>::FFT / IFFT  lib/gf2k/lch14.h:106-144               lfgpu::GpuLCH14<Field>
//   per-round body of ProverLayers::layer lib/sumcheck/prover_layers.h:230-263
//                                                                    lfgpu::GpuSumcheckRound<Field>
//
// The adapters only need from `Field`: Elt (16 bytes, trivially copyable), kBytes,
// kCharacteristicTwo, and for GF2_128 kSubFieldLogBits -- i.e. the reference's
// GF2_128<k> (lib/gf2k/gf2_128.h:35-64) and Fp128<> (lib/algebra/fp_p128.h:88) fit
// unchanged.  Failures of the C ABI are turned back into the reference's error
// behaviour: `lfgpu::check()` prints and aborts like proofs::check (lib/util/panic.h:27-37).
//
// This file compiles against the reference headers (tests/adapters_compile_check.cc does
// that in the build container) but does not include any of them.
#ifndef LFGPU_ADAPTERS_H_
#define LFGPU_ADAPTERS_H_

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <type_traits>
#include <vector>

#include "lfgpu.h"

namespace lfgpu {

inline void check(lfgpu_ctx* ctx, int rc, const char* what) {
  if (rc != LFGPU_OK) {
    std::fprintf(stderr, "lfgpu: %s failed (%d): %s\n", what, rc, ctx ? lfgpu_last_error(ctx) : "");
    std::abort();  // same contract as proofs::check()
  }
}

// One context per process/GPU; borrowed by every adapter (like `const Field&` in the reference).
class Context {
 public:
  explicit Context(int device = 0) { check(nullptr, lfgpu_init(device, &ctx_), "lfgpu_init"); }
  ~Context() { lfgpu_shutdown(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  lfgpu_ctx* get() const { return ctx_; }

 private:
  lfgpu_ctx* ctx_ = nullptr;
};

template <class Field>
constexpr int field_id() {  // GF2_128<k>, Fp128, Fp256Base (the only 32-byte field on the path: mdoc_zk.cc:70-76)
  return Field::kCharacteristicTwo ? LFGPU_FIELD_GF2_128 : Field::kBytes == 32 ? LFGPU_FIELD_P256 : LFGPU_FIELD_FP128;
}
template <class Field>
constexpr int subfield_log_bits() {
  if constexpr (Field::kCharacteristicTwo) {
    return static_cast<int>(Field::kSubFieldLogBits);
  } else {
    return 0;
  }
}

// ---------------------------------------------------------------- InterpolatorFactory
// Drop-in for LCH14ReedSolomonFactory<Field> / ReedSolomonFactory<Field, ...>.
// interpolate(y): y[0..n) valid, fills y[n..m) in place (host buffer, as in the reference).
// interpolate_rows(): the batched form LigeroProver's row loops collapse to.
template <class Field>
class GpuReedSolomon {
  using Elt = typename Field::Elt;
  static_assert(sizeof(Elt) == (field_id<Field>() == LFGPU_FIELD_P256 ? 32 : 16), "in-memory Elt image = kBytes");

 public:
  GpuReedSolomon(size_t n, size_t m, const Context& c, const uint64_t omega[2], uint64_t omega_order)
      : n_(n), m_(m), c_(c), omega_order_(omega_order) {
    omega_[0] = omega ? omega[0] : 0;
    omega_[1] = omega ? omega[1] : 0;
  }
  void interpolate(Elt y[/*m*/]) const { interpolate_rows(y, 1, m_); }
  void interpolate_rows(Elt* T, size_t nrow, size_t ld) const {
    if constexpr (Field::kCharacteristicTwo) {
      check(c_.get(), lfgpu_gf2128_rs_encode_rows_host(c_.get(), subfield_log_bits<Field>(), nrow, n_, m_, T, ld),
            "lfgpu_gf2128_rs_encode_rows_host");
    } else if constexpr (field_id<Field>() == LFGPU_FIELD_P256) {
      check(c_.get(), lfgpu_fp256_rs_encode_rows_host(c_.get(), nrow, n_, m_, T, ld), "lfgpu_fp256_rs_encode_rows_host");
    } else {
      check(c_.get(), lfgpu_fp128_rs_encode_rows_host(c_.get(), nrow, n_, m_, omega_, omega_order_, T, ld),
            "lfgpu_fp128_rs_encode_rows_host");
    }
  }

 private:
  size_t n_, m_;
  const Context& c_;
  uint64_t omega_[2];
  uint64_t omega_order_;
};

template <class Field>
class GpuReedSolomonFactory {
 public:
  // GF2_128 and Fp256Base: GpuReedSolomonFactory(ctx).  Fp128: pass the root of unity (Montgomery image)
  // and its order, as FFTConvolutionFactory does (lib/algebra/convolution.h:114-115).  Fp256Base replaces
  // ReedSolomonFactory<Fp256Base, FFTExtConvolutionFactory<Fp256Base, Fp2<Fp256Base>>> (mdoc_zk.cc:75-76,485-487); the
  // interpolation does not depend on the extension-field root, so none is passed.
  explicit GpuReedSolomonFactory(const Context& c, const typename Field::Elt* omega = nullptr,
                                 uint64_t omega_order = 0)
      : c_(c), omega_order_(omega_order) {
    omega_[0] = omega_[1] = 0;
    if (omega && field_id<Field>() == LFGPU_FIELD_FP128) std::memcpy(omega_, omega, 16);
  }
  std::unique_ptr<GpuReedSolomon<Field>> make(size_t n, size_t m) const {
    return std::make_unique<GpuReedSolomon<Field>>(n, m, c_, omega_, omega_order_);
  }

 private:
  const Context& c_;
  uint64_t omega_[2];
  uint64_t omega_order_;
};

// ---------------------------------------------------------------- FFTs
namespace detail {
// Fp2<Fp<1>> (lib/algebra/fp2.h:36-41 names its BaseField; the prime fields do not): the F64_2 of lib/algebra/fft_test.cc:205-229
template <class Field, class = void>
struct is_f64_2 : std::false_type {};
template <class Field>
struct is_f64_2<Field, std::void_t<typename Field::BaseField>> : std::bool_constant<Field::kBytes == 16 && Field::BaseField::kBytes == 8> {};
}  // namespace detail

template <class Field>
struct GpuFFT {  // FFT<Field>::fftb / fftf (lib/algebra/fft.h:185-201), host buffers; Field = Fp128<> or Fp2<Fp<1>>
  using Elt = typename Field::Elt;
  static_assert(sizeof(Elt) == 16, "16-byte elements");
  static void run(const Context& c, int dir, Elt A[/*n*/], size_t n, const Elt& omega_j, uint64_t j) {
    uint64_t w[2];
    std::memcpy(w, &omega_j, 16);
    if constexpr (detail::is_f64_2<Field>::value)
      check(c.get(), lfgpu_f64_2_fft_host(c.get(), dir, n, w, j, A), "lfgpu_f64_2_fft_host");
    else
      check(c.get(), lfgpu_fp128_fft_host(c.get(), dir, n, w, j, A), "lfgpu_fp128_fft_host");
  }
  static void fftb(const Context& c, Elt A[/*n*/], size_t n, const Elt& omega_j, uint64_t j) { run(c, 0, A, n, omega_j, j); }
  static void fftf(const Context& c, Elt A[/*n*/], size_t n, const Elt& omega_j, uint64_t j) { run(c, 1, A, n, omega_j, j); }
};

template <class Field>
class GpuLCH14 {  // LCH14<Field>::FFT / IFFT (lib/gf2k/lch14.h:106-144), host buffers
  using Elt = typename Field::Elt;

 public:
  explicit GpuLCH14(const Context& c) : c_(c) {}
  void FFT(size_t l, size_t coset, Elt B[]) const {
    check(c_.get(), lfgpu_gf2128_lch14_fft_host(c_.get(), subfield_log_bits<Field>(), 0, (unsigned)l, coset, B), "lch14 FFT");
  }
  void IFFT(size_t l, size_t coset, Elt B[]) const {
    check(c_.get(), lfgpu_gf2128_lch14_fft_host(c_.get(), subfield_log_bits<Field>(), 1, (unsigned)l, coset, B), "lch14 IFFT");
  }

 private:
  const Context& c_;
};

// ---------------------------------------------------------------- MerkleCommitment
// Same surface as proofs::MerkleCommitment, specialised to what LigeroProver::commit passes as
// `updhash` (LigeroCommon::column_hash over the tableau, ligero_prover.h:71-75): the caller hands
// the tableau instead of a per-leaf callback, so that the leaf loop can run on the GPU.
// RandomEngineT needs `void bytes(uint8_t*, size_t)` (lib/random/random.h:32-35).
struct Digest32 {
  uint8_t data[32];
};
class GpuMerkleCommitment {
 public:
  GpuMerkleCommitment(size_t n, const Context& c) : n_(n), c_(c), nonce_(32 * n), layers_(64 * n) {}

  template <class Elt, class RandomEngineT>
  Digest32 commit(int field, size_t nrow, size_t ld, size_t col0, const Elt* tableau, RandomEngineT& rng) {
    for (size_t i = 0; i < n_; ++i) rng.bytes(&nonce_[32 * i], 32);  // draw order of merkle_commitment.h:52-54
    Digest32 root;
    check(c_.get(),
          lfgpu_column_commit_host(c_.get(), field, nrow, ld, col0, n_, tableau, nonce_.data(), layers_.data(), root.data),
          "lfgpu_column_commit_host");
    return root;
  }
  // open(): nonces of the opened leaves + compressed path (merkle_commitment.h:66-73)
  void open(std::vector<Digest32>& nonces, std::vector<Digest32>& path, const size_t pos[], size_t np) const {
    nonces.resize(np);
    for (size_t i = 0; i < np; ++i) std::memcpy(nonces[i].data, &nonce_[32 * pos[i]], 32);
    std::vector<bool> tree(2 * n_, false);
    for (size_t i = 0; i < np; ++i) tree[pos[i] + n_] = true;
    for (size_t i = n_; i-- > 1;) tree[i] = tree[2 * i] || tree[2 * i + 1];
    for (size_t i = n_; i-- > 1;) {
      if (tree[i]) {
        size_t child = 2 * i;
        if (tree[child]) child = 2 * i + 1;
        if (!tree[child]) {
          Digest32 d;
          std::memcpy(d.data, &layers_[32 * child], 32);
          path.push_back(d);
        }
      }
    }
  }

 private:
  size_t n_;
  const Context& c_;
  std::vector<uint8_t> nonce_, layers_;
};

// ---------------------------------------------------------------- sumcheck round body
// Device-resident state of one ProverLayers::layer call (W hands, QW, HQUAD).  The host keeps
// the transcript: per round-hand it asks for (a0, a2) and pushes the challenge back.
template <class Field>
class GpuSumcheckRound {
  using Elt = typename Field::Elt;

 public:
  explicit GpuSumcheckRound(const Context& c) : c_(c) {}
  // a0 = sum QW[2i] W[2i], a2 = sum (QW[2i+1]-QW[2i])(W[2i+1]-W[2i]) (prover_layers.h:365-388)
  void partials(size_t n, const void* d_QW, const void* d_W, Elt& a0, Elt& a2) const {
    uint64_t x[2], y[2];
    check(c_.get(), lfgpu_sumcheck_partials(c_.get(), field_id<Field>(), n, d_QW, d_W, x, y), "sumcheck_partials");
    std::memcpy(&a0, x, 16);
    std::memcpy(&a2, y, 16);
  }
  void qw_scatter(size_t n, const void* d_hc, const void* d_vc, int hand, const void* d_Wother, size_t nqw, void* d_QW) const {
    check(c_.get(), lfgpu_qw_scatter(c_.get(), field_id<Field>(), n, d_hc, d_vc, hand, d_Wother, nqw, d_QW), "qw_scatter");
  }
  void bind_dense(size_t n0, const Elt& r, const void* d_in, void* d_out) const {
    uint64_t rr[2];
    std::memcpy(rr, &r, 16);
    check(c_.get(), lfgpu_dense_bind(c_.get(), field_id<Field>(), n0, rr, d_in, d_out), "dense_bind");
  }
  size_t bind_hquad(size_t n, const void* d_hc, const void* d_vc, const Elt& r, int hand, void* d_hc_out, void* d_vc_out) const {
    uint64_t rr[2];
    std::memcpy(rr, &r, 16);
    size_t n_out = 0;
    check(c_.get(), lfgpu_hquad_bind_h(c_.get(), field_id<Field>(), n, d_hc, d_vc, rr, hand, d_hc_out, d_vc_out, &n_out),
          "hquad_bind_h");
    return n_out;
  }

 private:
  const Context& c_;
};

}  // namespace lfgpu
#endif  // LFGPU_ADAPTERS_H_
