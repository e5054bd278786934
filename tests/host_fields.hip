// host_fields.hip -- compiles longfellow-zk_amd/csrc/fields.h (the device arithmetic)
// for the HOST, so the limb-level logic (Montgomery carries, Karatsuba clmul, SHA-256
// rounds) is unit-tested against the oracle in the CPU-only container.
#include "../longfellow-zk_amd/csrc/fields.h"

extern "C" {
void hf_fp_mul(const u64* a, const u64* b, u64* o) { elt_t r = fp_mul(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
void hf_fp_add(const u64* a, const u64* b, u64* o) { elt_t r = fp_add(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
void hf_fp_sub(const u64* a, const u64* b, u64* o) { elt_t r = fp_sub(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
void hf_fp_reduce_limbs(const u64* a, u64* o) { elt_t r = fp_reduce_limbs(a[0], a[1], a[2], a[3]); o[0] = r.lo; o[1] = r.hi; }
void hf_gf_mul(const u64* a, const u64* b, u64* o) { elt_t r = gf_mul(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
// F64 (p = 2^64 - 2^32 + 1, shift-only Montgomery reduction) and F64_2 = Fp2<F64>: n pairs at once
void hf_f64_mul_many(size_t n, const u64* a, const u64* b, u64* o) { for (size_t i = 0; i < n; ++i) o[i] = f64_mul(a[i], b[i]); }
void hf_f64_add_many(size_t n, const u64* a, const u64* b, u64* o) { for (size_t i = 0; i < n; ++i) o[i] = f64_add(a[i], b[i]); }
void hf_f64_sub_many(size_t n, const u64* a, const u64* b, u64* o) { for (size_t i = 0; i < n; ++i) o[i] = f64_sub(a[i], b[i]); }
void hf_f64x2_mul(const u64* a, const u64* b, u64* o) { elt_t r = f64x2_mul(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
void hf_f64x2_mul_real(const u64* a, const u64* b, u64* o) { elt_t r = f64x2_mul_real(elt_t{a[0], a[1]}, b[0]); o[0] = r.lo; o[1] = r.hi; }
// SHA-256 of nblk whole 64-byte blocks (no padding): returns raw state words
void hf_sha_blocks(const unsigned char* p, unsigned nblk, u32* h) {
  sha_state s;
  sha_init(s);
  for (unsigned b = 0; b < nblk; ++b) {
    u32 w[16];
    for (int i = 0; i < 16; ++i)
      w[i] = ((u32)p[64 * b + 4 * i] << 24) | ((u32)p[64 * b + 4 * i + 1] << 16) | ((u32)p[64 * b + 4 * i + 2] << 8) | p[64 * b + 4 * i + 3];
    sha_compress(s, w);
  }
  for (int i = 0; i < 8; ++i) h[i] = s.h[i];
}
}

// ---- bit-sliced tower arithmetic (csrc/bitslice.h), host compilation
#include "../longfellow-zk_amd/csrc/bitslice.h"

template <int K>
static void bs_roundtrip_mul(const u64* t_poly, const u64* in, u64* out, int mode) {
  constexpr int M = Tower<K>::M, D = Tower<K>::D;
  constexpr u32 MU = Tower<K>::MU_LOW;
  // rows -> poly planes
  u32 P[128], T[128], R[128], Q[128];
  for (int w = 0; w < 4; ++w) {
    u32 x[32];
    for (int r = 0; r < 32; ++r) x[r] = (u32)(in[2 * r + (w >> 1)] >> (32 * (w & 1)));
    bs_transpose32(x);
    for (int b = 0; b < 32; ++b) P[32 * w + b] = x[b];
  }
#define IN_(i) P[i]
  if (K == 4) {
#define O0(o) T[0 * 16 + o]
#define O1(o) T[1 * 16 + o]
#define O2(o) T[2 * 16 + o]
#define O3(o) T[3 * 16 + o]
#define O4(o) T[4 * 16 + o]
#define O5(o) T[5 * 16 + o]
#define O6(o) T[6 * 16 + o]
#define O7(o) T[7 * 16 + o]
    TOWER_K4_P2T_Q0(IN_, O0); TOWER_K4_P2T_Q1(IN_, O1); TOWER_K4_P2T_Q2(IN_, O2); TOWER_K4_P2T_Q3(IN_, O3);
    TOWER_K4_P2T_Q4(IN_, O4); TOWER_K4_P2T_Q5(IN_, O5); TOWER_K4_P2T_Q6(IN_, O6); TOWER_K4_P2T_Q7(IN_, O7);
  } else {
#define U0(o) T[0 * 32 + o]
#define U1(o) T[1 * 32 + o]
#define U2(o) T[2 * 32 + o]
#define U3(o) T[3 * 32 + o]
    TOWER_K5_P2T_Q0(IN_, U0); TOWER_K5_P2T_Q1(IN_, U1); TOWER_K5_P2T_Q2(IN_, U2); TOWER_K5_P2T_Q3(IN_, U3);
  }
  u32 t = tower_twiddle_bits<K>(t_poly[0], t_poly[1]);
  for (int q = 0; q < D; ++q) {
    u32 b[M], dst[M];
    for (int j = 0; j < M; ++j) { b[j] = T[q * M + j]; dst[j] = 0; }
    if (mode == 0) bs_mac_uniform<M, MU>(t, b, dst); else bs_mac_lane<M, MU>(t, b, dst);
    for (int j = 0; j < M; ++j) R[q * M + j] = dst[j];
  }
#define RIN(i) R[i]
#define Q0(o) Q[0 + o]
#define Q1(o) Q[32 + o]
#define Q2(o) Q[64 + o]
#define Q3(o) Q[96 + o]
  if (K == 4) { TOWER_K4_T2P_W0(RIN, Q0); TOWER_K4_T2P_W1(RIN, Q1); TOWER_K4_T2P_W2(RIN, Q2); TOWER_K4_T2P_W3(RIN, Q3); }
  else { TOWER_K5_T2P_W0(RIN, Q0); TOWER_K5_T2P_W1(RIN, Q1); TOWER_K5_T2P_W2(RIN, Q2); TOWER_K5_T2P_W3(RIN, Q3); }
  for (int r = 0; r < 64; ++r) out[r] = 0;
  for (int w = 0; w < 4; ++w) {
    u32 x[32];
    for (int b = 0; b < 32; ++b) x[b] = Q[32 * w + b];
    bs_transpose32(x);
    for (int r = 0; r < 32; ++r) out[2 * r + (w >> 1)] |= (u64)x[r] << (32 * (w & 1));
  }
}
extern "C" void hf_bs_mul(int k, const u64* t_poly, const u64* in, u64* out, int mode) {
  if (k == 4) bs_roundtrip_mul<4>(t_poly, in, out, mode); else bs_roundtrip_mul<5>(t_poly, in, out, mode);
}
extern "C" void hf_transpose32(u32* x) {
  u32 y[32];
  for (int i = 0; i < 32; ++i) y[i] = x[i];
  bs_transpose32(y);
  for (int i = 0; i < 32; ++i) x[i] = y[i];
}

// ---- fp256.h (P-256 base field, 32-byte elements)
#include "../longfellow-zk_amd/csrc/fp256.h"
extern "C" void hf_p256_mul(const elt32_t* a, const elt32_t* b, elt32_t* o) { *o = fp256_mul(*a, *b); }
extern "C" void hf_p256_add(const elt32_t* a, const elt32_t* b, elt32_t* o) { *o = fp256_add(*a, *b); }
extern "C" void hf_p256_sub(const elt32_t* a, const elt32_t* b, elt32_t* o) { *o = fp256_sub(*a, *b); }
extern "C" void hf_p256_canon(const elt32_t* a, elt32_t* o) { *o = fp256_canon(*a); }
extern "C" void hf_p256_c2mul(const fp2_t* a, const fp2_t* b, fp2_t* o) { *o = fp2_mul(*a, *b); }
// limb accumulators of the P-256 sumcheck sums (zk256.hip) and the host helpers next to them
extern "C" void hf_p256_reduce_limbs(const u64* acc, elt32_t* o) { *o = fp256_reduce_limbs(acc, h256_rsq()); }
extern "C" void hf_p256_of_scalar(u64 u, elt32_t* o) { *o = h256_of_scalar(u); }
extern "C" void hf_p256_inv(const elt32_t* a, elt32_t* o) { *o = h256_inv(*a); }
extern "C" int hf_p256_of_bytes(const uint8_t* b, elt32_t* o) { return h256_of_bytes(b, *o) ? 1 : 0; }
// bulk sampling vs one attempt at a time over the same byte stream (FpGeneric::sample, fp_generic.h:360-371)
struct HfStream {
  const uint8_t* p;
  size_t left, calls;
};
static void hf_take(HfStream& st, uint8_t* b, size_t n) {
  if (n > st.left) n = st.left;  // the tests size the stream generously; running dry would show as a mismatch
  memcpy(b, st.p, n);
  st.p += n;
  st.left -= n;
  ++st.calls;
}
extern "C" size_t hf_p256_sample_both(const uint8_t* stream, size_t len, size_t n, elt32_t* bulk, elt32_t* single, size_t* used) {
  HfStream a{stream, len, 0}, b{stream, len, 0};
  h256_sample_many(bulk, n, [&](uint8_t* o, size_t k) { hf_take(a, o, k); });
  for (size_t i = 0; i < n; ++i) single[i] = h256_sample([&](uint8_t* o, size_t k) { hf_take(b, o, k); });
  used[0] = len - a.left;
  used[1] = len - b.left;
  return a.calls;
}
