// ref_shim.cc -- thin extern "C" driver around the REAL reference templates.
//
// TEST INFRASTRUCTURE ONLY (see lf_oracle.h).  This file contains no reference
// source: it #includes the reference headers where they lie under
// /root/reference/lib and is compiled by oracle/Makefile into oracle/_ref/
// (git-ignored; travels to the GPU box as a built .so for cpu_baseline
// "reference").  It is used to (1) validate the C restatement in lf_oracle.c,
// (2) generate tests/golden/*.bin, (3) time the reference CPU path.
//
// All element buffers are raw 16-byte Elt images (memcpy in/out).

#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <utility>
#include <vector>

#include "algebra/blas.h"
#include "algebra/bogorng.h"
#include "algebra/convolution.h"
#include "algebra/fft.h"
#include "algebra/fp.h"
#include "algebra/fp2.h"
#include "algebra/fp_p128.h"
#include "algebra/fp_p256.h"
#include "algebra/rfft.h"
#include "algebra/poly.h"
#include "algebra/reed_solomon.h"
#include "arrays/affine.h"
#include "arrays/dense.h"
#include "arrays/eqs.h"
#include "gf2k/gf2_128.h"
#include "gf2k/lch14.h"
#include "gf2k/lch14_reed_solomon.h"
#include "ligero/ligero_param.h"
#include "merkle/merkle_commitment.h"
#include "merkle/merkle_tree.h"
#include "random/random.h"
#include "sumcheck/hquad.h"
#include "sumcheck/quad.h"
// ProverLayers::evaluations is private; the shim needs to call it directly.
#define private public
#include "sumcheck/prover_layers.h"
#undef private

using namespace proofs;

namespace {
using GF4 = GF2_128<4>;
using GF5 = GF2_128<5>;
using FP = Fp128<>;

const GF4& gf4() {
  static const GF4 f;
  return f;
}
const GF5& gf5() {
  static const GF5 f;
  return f;
}
const FP& fp() {
  static const FP f;
  return f;
}

static_assert(sizeof(GF4::Elt) == 16, "GF2_128 Elt is 16 bytes");
static_assert(sizeof(FP::Elt) == 16, "Fp128 Elt is 16 bytes");

template <class E>
E ld(const void* p) {
  E e;
  memcpy(&e, p, 16);
  return e;
}
template <class E>
void st(void* p, const E& e) {
  memcpy(p, &e, 16);
}

FP::Elt omega32() {
  return fp().of_string("164956748514267535023998284330560247862");
}

// RandomEngine that replays a caller-supplied byte stream.
class BufferRng : public RandomEngine {
 public:
  BufferRng(const uint8_t* p, size_t n) : p_(p), n_(n), pos_(0) {}
  void bytes(uint8_t* buf, size_t n) override {
    check(pos_ + n <= n_, "BufferRng exhausted");
    memcpy(buf, p_ + pos_, n);
    pos_ += n;
  }

 private:
  const uint8_t* p_;
  size_t n_, pos_;
};

template <class Field>
void gf_lch14_fft(const Field& F, int dir, size_t l, size_t coset, void* B) {
  LCH14<Field> fft(F);
  auto* b = reinterpret_cast<typename Field::Elt*>(B);
  if (dir == 0)
    fft.FFT(l, coset, b);
  else if (dir == 1)
    fft.IFFT(l, coset, b);
  else
    fft.BidirectionalFFT(l, coset /* = k */, b);
}
}  // namespace

extern "C" {

// ---------------------------------------------------------------- GF(2^128)
void ref_gf_mul(const void* a, const void* b, void* out) {
  st(out, gf4().mulf(ld<GF4::Elt>(a), ld<GF4::Elt>(b)));
}
void ref_gf_inv(const void* a, void* out) { st(out, gf4().invertf(ld<GF4::Elt>(a))); }
void ref_gf_of_scalar(int k, uint64_t u, void* out) {
  if (k == 4)
    st(out, gf4().of_scalar(u));
  else
    st(out, gf5().of_scalar(u));
}
void ref_gf_beta(int k, size_t i, void* out) {
  if (k == 4)
    st(out, gf4().beta(i));
  else
    st(out, gf5().beta(i));
}
void ref_gf_poly_evaluation_point(int k, size_t i, void* out) {
  if (k == 4)
    st(out, gf4().poly_evaluation_point(i));
  else
    st(out, gf5().poly_evaluation_point(i));
}
void ref_lch14_twiddle(int k, size_t i, size_t u, void* out) {
  if (k == 4) {
    LCH14<GF4> f(gf4());
    st(out, f.twiddle(i, u));
  } else {
    LCH14<GF5> f(gf5());
    st(out, f.twiddle(i, u));
  }
}
// dir: 0 = FFT, 1 = IFFT, 2 = BidirectionalFFT (coset := k)
void ref_lch14_fft(int k, int dir, size_t l, size_t coset, void* B) {
  if (k == 4)
    gf_lch14_fft(gf4(), dir, l, coset, B);
  else
    gf_lch14_fft(gf5(), dir, l, coset, B);
}
void ref_lch14_rs_interpolate(int k, size_t n, size_t m, void* y) {
  if (k == 4) {
    LCH14ReedSolomonFactory<GF4> fac(gf4());
    fac.make(n, m)->interpolate(reinterpret_cast<GF4::Elt*>(y));
  } else {
    LCH14ReedSolomonFactory<GF5> fac(gf5());
    fac.make(n, m)->interpolate(reinterpret_cast<GF5::Elt*>(y));
  }
}
// nrow rows, stride ld elements; same (n, m) for all rows.  Used for timing.
void ref_lch14_rs_encode_rows(int k, size_t nrow, size_t n, size_t m, void* T, size_t ld_) {
  if (k == 4) {
    LCH14ReedSolomonFactory<GF4> fac(gf4());
    auto rs = fac.make(n, m);
    for (size_t r = 0; r < nrow; ++r) rs->interpolate(reinterpret_cast<GF4::Elt*>(T) + r * ld_);
  } else {
    LCH14ReedSolomonFactory<GF5> fac(gf5());
    auto rs = fac.make(n, m);
    for (size_t r = 0; r < nrow; ++r) rs->interpolate(reinterpret_cast<GF5::Elt*>(T) + r * ld_);
  }
}

// ---------------------------------------------------------------- Fp128
void ref_fp_mul(const void* a, const void* b, void* out) {
  st(out, fp().mulf(ld<FP::Elt>(a), ld<FP::Elt>(b)));
}
void ref_fp_add(const void* a, const void* b, void* out) {
  st(out, fp().addf(ld<FP::Elt>(a), ld<FP::Elt>(b)));
}
void ref_fp_sub(const void* a, const void* b, void* out) {
  st(out, fp().subf(ld<FP::Elt>(a), ld<FP::Elt>(b)));
}
void ref_fp_inv(const void* a, void* out) { st(out, fp().invertf(ld<FP::Elt>(a))); }
void ref_fp_of_scalar(uint64_t u, void* out) { st(out, fp().of_scalar(u)); }
void ref_fp_from_mont(const void* a, void* out) {
  auto n = fp().from_montgomery(ld<FP::Elt>(a));
  memcpy(out, &n, 16);
}
void ref_fp_omega32(void* out) { st(out, omega32()); }
void ref_fp_bogorng_fill(uint64_t seed, size_t n, void* out) {
  Bogorng<FP> rng(&fp(), seed);
  auto* o = reinterpret_cast<FP::Elt*>(out);
  for (size_t i = 0; i < n; ++i) o[i] = rng.next();
}
// backward (dir 0) / forward (dir 1) FFT with the true 2^32-order root
void ref_fp_fft(int dir, size_t n, void* A) {
  auto* a = reinterpret_cast<FP::Elt*>(A);
  if (dir == 0)
    FFT<FP>::fftb(a, n, omega32(), uint64_t(1) << 32, fp());
  else
    FFT<FP>::fftf(a, n, omega32(), uint64_t(1) << 32, fp());
}
// ---- F64_2 = Fp2<Fp<1>> (lib/algebra/fft_test.cc:205-229)
using F64 = Fp<1>;
using F64_2 = Fp2<F64>;
static const F64& f64() {
  static const F64 f("18446744069414584321");
  return f;
}
static const F64_2& f64_2() {
  static const F64_2 f(f64());
  return f;
}
static F64_2::Elt f64_2_omega32() {
  static constexpr char kSmallRoot[] = "2752994695033296049";
  return f64_2().of_string(kSmallRoot);
}
void ref_f64_2_binop(int op, const void* a, const void* b, void* out) {  // 0 add, 1 sub, 2 mul, 3 inverse of a
  const auto x = ld<F64_2::Elt>(a), y = ld<F64_2::Elt>(b);
  st(out, op == 0 ? f64_2().addf(x, y) : op == 1 ? f64_2().subf(x, y) : op == 2 ? f64_2().mulf(x, y) : f64_2().invertf(x));
}
void ref_f64_2_of_scalar(uint64_t re, uint64_t im, void* out) { st(out, f64_2().of_scalar_field(re, im)); }
void ref_f64_2_omega32(void* out) { st(out, f64_2_omega32()); }
// imag = 0: real elements as in BM_FFT_F64_2; imag = 1: a second generator fills the imaginary parts
void ref_f64_2_bogorng_fill(uint64_t seed, int imag, size_t n, void* out) {
  Bogorng<F64> re(&f64(), seed), im(&f64(), seed + 17);
  auto* o = reinterpret_cast<F64_2::Elt*>(out);
  for (size_t i = 0; i < n; ++i) {
    o[i] = f64_2().of_scalar(re.next());
    if (imag) o[i].im = im.next();
  }
}
// omega == nullptr: the real root of order 2^32; else any root of that order (an Fp2 element)
void ref_f64_2_fft(int dir, size_t n, const void* omega, void* A) {
  static_assert(sizeof(F64_2::Elt) == 16, "F64_2 element is two 64-bit limbs");
  auto* a = reinterpret_cast<F64_2::Elt*>(A);
  const F64_2::Elt w = omega ? ld<F64_2::Elt>(omega) : f64_2_omega32();
  if (dir == 0)
    FFT<F64_2>::fftb(a, n, w, uint64_t(1) << 32, f64_2());
  else
    FFT<F64_2>::fftf(a, n, w, uint64_t(1) << 32, f64_2());
}

void ref_fp_rs_interpolate(size_t n, size_t m, void* y) {
  FFTConvolutionFactory<FP> cf(fp(), omega32(), uint64_t(1) << 32);
  ReedSolomonFactory<FP, FFTConvolutionFactory<FP>> rsf(cf, fp());
  rsf.make(n, m)->interpolate(reinterpret_cast<FP::Elt*>(y));
}

// ---------------------------------------------------------------- Merkle
// leaves[n][32] -> layers[2n][32]
void ref_merkle_build_tree(size_t n, const uint8_t* leaves, uint8_t* layers) {
  MerkleTree mt(n);
  for (size_t i = 0; i < n; ++i) {
    Digest d;
    memcpy(d.data, leaves + 32 * i, 32);
    mt.set_leaf(i, d);
  }
  mt.build_tree();
  memset(layers, 0, 32);
  for (size_t i = 1; i < 2 * n; ++i) memcpy(layers + 32 * i, mt.layers_[i].data, 32);
}

// MerkleCommitment::commit over the columns [col0, col0+ncols) of a tableau,
// with nonces replayed from `nonces` (ncols*32 bytes).  field: 4 = GF2_128, 6 = Fp128.
void ref_column_commit(int field, size_t nrow, size_t ld_, size_t col0, size_t ncols,
                       const void* tableau, const uint8_t* nonces, uint8_t* root_out) {
  BufferRng rng(nonces, 32 * ncols);
  MerkleCommitment mc(ncols);
  Digest root;
  if (field == 4) {
    const auto* T = reinterpret_cast<const GF4::Elt*>(tableau);
    auto upd = [&](size_t j, proofs::SHA256& sha) {
      LigeroCommon<GF4>::column_hash(nrow, &T[j + col0], ld_, sha, gf4());
    };
    root = mc.commit(upd, rng);
  } else {
    const auto* T = reinterpret_cast<const FP::Elt*>(tableau);
    auto upd = [&](size_t j, proofs::SHA256& sha) {
      LigeroCommon<FP>::column_hash(nrow, &T[j + col0], ld_, sha, fp());
    };
    root = mc.commit(upd, rng);
  }
  memcpy(root_out, root.data, 32);
}

}  // extern "C"

// ---------------------------------------------------------------- sumcheck pieces
template <class Field>
static void evals_t(const Field& F, size_t n, const void* eq0, const void* QW, const void* W,
                    const void* sum, void* evals) {
  using Elt = typename Field::Elt;
  ProverLayers<Field> pl(F);
  auto e = pl.evaluations(n, ld<Elt>(eq0), reinterpret_cast<const Elt*>(QW),
                          reinterpret_cast<const Elt*>(W), ld<Elt>(sum), F);
  for (int k = 0; k < 3; ++k) st(reinterpret_cast<uint8_t*>(evals) + 16 * k, e[k]);
}
extern "C" void ref_sumcheck_evaluations(int field, size_t n, const void* eq0, const void* QW, const void* W,
                              const void* sum, void* evals) {
  if (field == 4)
    evals_t(gf4(), n, eq0, QW, W, sum, evals);
  else
    evals_t(fp(), n, eq0, QW, W, sum, evals);
}

template <class Field>
static size_t dense_bind_t(const Field& F, size_t n0, const void* r, void* v) {
  using Elt = typename Field::Elt;
  Dense<Field> d(n0, 1);
  memcpy(&d.v_[0], v, 16 * n0);
  d.bind(ld<Elt>(r), F);
  memcpy(v, &d.v_[0], 16 * d.n0_);
  return d.n0_;
}
extern "C" size_t ref_dense_bind(int field, size_t n0, const void* r, void* v) {
  return field == 4 ? dense_bind_t(gf4(), n0, r, v) : dense_bind_t(fp(), n0, r, v);
}

template <class Field>
static size_t hquad_bind_t(const Field& F, size_t n, uint32_t* hc, void* vc, const void* r, int hand) {
  using Elt = typename Field::Elt;
  using HQ = HQuad<Field>;
  HQ h(n);
  for (size_t i = 0; i < n; ++i) {
    h.hc_[i].h[0] = typename HQ::quad_corner_t(hc[2 * i]);
    h.hc_[i].h[1] = typename HQ::quad_corner_t(hc[2 * i + 1]);
    h.vc_[i].v = ld<Elt>(reinterpret_cast<uint8_t*>(vc) + 16 * i);
  }
  h.bind_h(ld<Elt>(r), hand, F);
  for (size_t i = 0; i < h.n_; ++i) {
    hc[2 * i] = static_cast<uint32_t>(size_t(h.hc_[i].h[0]));
    hc[2 * i + 1] = static_cast<uint32_t>(size_t(h.hc_[i].h[1]));
    st(reinterpret_cast<uint8_t*>(vc) + 16 * i, h.vc_[i].v);
  }
  return h.n_;
}
extern "C" size_t ref_hquad_bind_h(int field, size_t n, uint32_t* hc, void* vc, const void* r, int hand) {
  return field == 4 ? hquad_bind_t(gf4(), n, hc, vc, r, hand) : hquad_bind_t(fp(), n, hc, vc, r, hand);
}

// ---------------------------------------------------------------- quad: eval_quad / bind_g / raw_eq2
template <class Field>
static std::unique_ptr<Quad<Field>> make_quad(size_t n, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                                              const uint32_t* vi, size_t nk, const void* kvec) {
  using Elt = typename Field::Elt;
  using Q = Quad<Field>;
  using qc = typename Q::quad_corner_t;
  auto kv = std::make_shared<std::vector<Elt>>(nk);
  memcpy(kv->data(), kvec, 16 * nk);
  auto dt = std::make_shared<typename Q::delta_table_t>(n);
  auto q = std::make_unique<Q>(n, kv, dt);
  uint32_t pg = 0, p0 = 0, p1 = 0;
  for (size_t i = 0; i < n; ++i) {
    (*dt)[i].dg = qc(uint32_t(g[i] - pg));
    (*dt)[i].dh[0] = qc(uint32_t(h0[i] - p0));
    (*dt)[i].dh[1] = qc(uint32_t(h1[i] - p1));
    (*dt)[i].vi = vi[i];
    q->assign(i, static_cast<uint32_t>(i));
    pg = g[i];
    p0 = h0[i];
    p1 = h1[i];
  }
  return q;
}
template <class Field>
static int eval_quad_t(const Field& F, size_t n, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                       const uint32_t* vi, size_t nk, const void* kvec, size_t nv, size_t nw, const void* W, void* V) {
  auto q = make_quad<Field>(n, g, h0, h1, vi, nk, kvec);
  Dense<Field> dW(1, nw), dV(1, nv);
  memcpy(&dW.v_[0], W, 16 * nw);
  ProverLayers<Field> pl(F);
  bool ok = pl.eval_quad(q.get(), &dV, &dW, F);
  memcpy(V, &dV.v_[0], 16 * nv);
  return ok ? 1 : 0;
}
extern "C" int ref_eval_quad(int field, size_t n, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                             const uint32_t* vi, size_t nk, const void* kvec, size_t nv, size_t nw, const void* W, void* V) {
  return field == 4 ? eval_quad_t(gf4(), n, g, h0, h1, vi, nk, kvec, nv, nw, W, V)
                    : eval_quad_t(fp(), n, g, h0, h1, vi, nk, kvec, nv, nw, W, V);
}
template <class Field>
static size_t bind_g_t(const Field& F, size_t n, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                       const uint32_t* vi, size_t nk, const void* kvec, size_t logv, const void* G0, const void* G1,
                       const void* alpha, const void* beta, uint32_t* hc, void* vc) {
  using Elt = typename Field::Elt;
  auto q = make_quad<Field>(n, g, h0, h1, vi, nk, kvec);
  auto hq = q->bind_g(logv, reinterpret_cast<const Elt*>(G0), reinterpret_cast<const Elt*>(G1), ld<Elt>(alpha),
                      ld<Elt>(beta), F);
  for (size_t i = 0; i < hq->n_; ++i) {
    hc[2 * i] = static_cast<uint32_t>(size_t(hq->hc_[i].h[0]));
    hc[2 * i + 1] = static_cast<uint32_t>(size_t(hq->hc_[i].h[1]));
    st(reinterpret_cast<uint8_t*>(vc) + 16 * i, hq->vc_[i].v);
  }
  return hq->n_;
}
extern "C" size_t ref_quad_bind_g(int field, size_t n, const uint32_t* g, const uint32_t* h0, const uint32_t* h1,
                                  const uint32_t* vi, size_t nk, const void* kvec, size_t logv, const void* G0,
                                  const void* G1, const void* alpha, const void* beta, uint32_t* hc, void* vc) {
  return field == 4 ? bind_g_t(gf4(), n, g, h0, h1, vi, nk, kvec, logv, G0, G1, alpha, beta, hc, vc)
                    : bind_g_t(fp(), n, g, h0, h1, vi, nk, kvec, logv, G0, G1, alpha, beta, hc, vc);
}
template <class Field>
static void raw_eq2_t(const Field& F, size_t logn, size_t n, const void* G0, const void* G1, const void* alpha, void* out) {
  using Elt = typename Field::Elt;
  auto v = Eqs<Field>::raw_eq2(logn, n, reinterpret_cast<const Elt*>(G0), reinterpret_cast<const Elt*>(G1), ld<Elt>(alpha), F);
  memcpy(out, v.data(), 16 * n);
}
extern "C" void ref_raw_eq2(int field, size_t logn, size_t n, const void* G0, const void* G1, const void* alpha, void* out) {
  if (field == 4) raw_eq2_t(gf4(), logn, n, G0, G1, alpha, out); else raw_eq2_t(fp(), logn, n, G0, G1, alpha, out);
}

// ---------------------------------------------------------------- P-256 base field (BASELINE config 5: the mdoc
// signature circuit's ZkProver<Fp256Base, ReedSolomonFactory<Fp256Base, FFTExtConvolutionFactory>>,
// lib/circuits/mdoc/mdoc_zk.cc:70-88,485-500)
namespace {
using F256 = Fp256<true>;  // = Fp256Base (lib/ec/p256.h:42)
using F256_2 = Fp2<F256>;
static_assert(sizeof(F256::Elt) == 32, "Fp256Base Elt is 32 bytes");
const F256& f256() {
  static const F256 f;
  return f;
}
const F256_2& f256_2() {
  static const F256_2 f(f256());
  return f;
}
F256_2::Elt omega31() {
  return f256_2().of_string("112649224146410281873500457609690258373018840430489408729223714171582664680802",
                            "84087994358540907695740461427818660560182168997182378749313018254450460212908");
}
F256::Elt ld32(const void* p) {
  F256::Elt e;
  memcpy(&e, p, 32);
  return e;
}
void st32(void* p, const F256::Elt& e) { memcpy(p, &e, 32); }
}  // namespace

extern "C" {
void ref_p256_mul(const void* a, const void* b, void* out) { st32(out, f256().mulf(ld32(a), ld32(b))); }
void ref_p256_add(const void* a, const void* b, void* out) { st32(out, f256().addf(ld32(a), ld32(b))); }
void ref_p256_sub(const void* a, const void* b, void* out) { st32(out, f256().subf(ld32(a), ld32(b))); }
void ref_p256_inv(const void* a, void* out) { st32(out, f256().invertf(ld32(a))); }
void ref_p256_of_scalar(uint64_t u, void* out) { st32(out, f256().of_scalar(u)); }
void ref_p256_to_bytes(const void* a, uint8_t out[32]) { f256().to_bytes_field(out, ld32(a)); }
void ref_p256_omega(void* re, void* im) {
  auto w = omega31();
  st32(re, w.re);
  st32(im, w.im);
}
void ref_p256_rfft(int dir, size_t n, void* A) {  // dir 0: r2hc, 1: hc2r
  auto* a = reinterpret_cast<F256::Elt*>(A);
  if (dir == 0)
    RFFT<F256_2>::r2hc(a, n, omega31(), uint64_t(1) << 31, f256_2());
  else
    RFFT<F256_2>::hc2r(a, n, omega31(), uint64_t(1) << 31, f256_2());
}
void ref_p256_rs_interpolate(size_t n, size_t m, void* y) {
  using CF = FFTExtConvolutionFactory<F256, F256_2>;
  CF cf(f256(), f256_2(), omega31(), uint64_t(1) << 31);
  ReedSolomonFactory<F256, CF> rsf(cf, f256());
  rsf.make(n, m)->interpolate(reinterpret_cast<F256::Elt*>(y));
}
void ref_p256_rs_encode_rows(size_t nrow, size_t n, size_t m, void* T, size_t ld_) {
  using CF = FFTExtConvolutionFactory<F256, F256_2>;
  CF cf(f256(), f256_2(), omega31(), uint64_t(1) << 31);
  ReedSolomonFactory<F256, CF> rsf(cf, f256());
  auto rs = rsf.make(n, m);
  auto* t = reinterpret_cast<F256::Elt*>(T);
  for (size_t r = 0; r < nrow; ++r) rs->interpolate(t + r * ld_);
}
void ref_p256_column_commit(size_t nrow, size_t ld_, size_t col0, size_t ncols, const void* tableau, const uint8_t* nonces,
                            uint8_t* root_out) {
  BufferRng rng(nonces, 32 * ncols);
  MerkleCommitment mc(ncols);
  const auto* T = reinterpret_cast<const F256::Elt*>(tableau);
  auto upd = [&](size_t j, proofs::SHA256& sha) { LigeroCommon<F256>::column_hash(nrow, &T[j + col0], ld_, sha, f256()); };
  Digest root = mc.commit(upd, rng);
  memcpy(root_out, root.data, 32);
}
}  // extern "C"
