// fp256.h -- the P-256 base field Fp256Base = FpGeneric<4, true, Fp256Reduce> (reference lib/algebra/fp_p256.h:32-63,
// lib/algebra/fp_generic.h) and its quadratic extension Fp2<Fp256Base> (lib/algebra/fp2.h), host + device.
//
// Element = the reference's in-memory Elt image: 4 x u64 little-endian limbs of the Montgomery form x * 2^256 mod p,
// 32 bytes.  p = 2^256 - 2^224 + 2^192 + 2^96 - 1, so p = -1 (mod 2^96): -p^-1 mod 2^32 = 1 and a reduction step needs no
// multiplication -- with m = the low limb, m * p = m * 2^256 - m * 2^224 + m * 2^192 + m * 2^96 - m (fp_p256.h:43-62).
#pragma once
#include <string.h>

#include <vector>

#include "fields.h"

struct alignas(16) elt32_t {
  u64 l[4];
};
struct alignas(16) fp2_t {  // Fp2<Fp256Base>::Elt {re, im}
  elt32_t re, im;
};

#define FIELD_P256 1  // FieldID P256_ID (lib/proto/circuit_io.h:24-36)

LF_HD elt32_t e32_zero() { return elt32_t{{0, 0, 0, 0}}; }
LF_HD bool e32_eq(const elt32_t& a, const elt32_t& b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }

// p as 8 x u32 limbs
#define P256_W0 0xFFFFFFFFu
#define P256_W1 0xFFFFFFFFu
#define P256_W2 0xFFFFFFFFu
#define P256_W3 0x00000000u
#define P256_W4 0x00000000u
#define P256_W5 0x00000000u
#define P256_W6 0x00000001u
#define P256_W7 0xFFFFFFFFu

namespace fp256_detail {
LF_HD void to_w(const elt32_t& a, u32 (&w)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w[2 * i] = (u32)a.l[i];
    w[2 * i + 1] = (u32)(a.l[i] >> 32);
  }
}
LF_HD elt32_t from_w(const u32 (&w)[8]) {
  elt32_t r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.l[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32);
  return r;
}
LF_HD u32 pw(int i) {
  return i == 0 ? P256_W0 : i == 1 ? P256_W1 : i == 2 ? P256_W2 : i == 3 ? P256_W3 : i == 4 ? P256_W4 : i == 5 ? P256_W5 : i == 6 ? P256_W6 : P256_W7;
}
// r = a - p if a >= p (a < 2p given with its carry-out bit `hi`)
LF_HD void cond_sub_p(u32 (&t)[8], u32 hi) {
  u32 d[8];
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 x = (u64)t[i] - pw(i) - br;
    d[i] = (u32)x;
    br = (x >> 32) & 1u;
  }
  const bool ge = hi != 0 || br == 0;  // a >= p
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = ge ? d[i] : t[i];
}
}  // namespace fp256_detail

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 versions: 32-bit limb carry chains through VCC (see the Fp128 ones in fields.h for the wait states).  hipcc's lowering
// of the portable code below is 122 / 96 instructions per add / sub (64-bit carries rebuilt with moves and v_lshl_add_u64);
// these are 26 / 18 -- and a round-hand of the Fp256Base sumcheck is a chain of ~20 of them on one wave.
#define P256_NOP "s_nop 1\n\t"
__device__ __forceinline__ elt32_t fp256_add(const elt32_t& a, const elt32_t& b) {
  u32 x[8], y[8], d[8], c;
  fp256_detail::to_w(a, x);
  fp256_detail::to_w(b, y);
  asm("v_add_co_u32 %0, vcc, %0, %9\n\t" P256_NOP
      "v_addc_co_u32 %1, vcc, %1, %10, vcc\n\t" P256_NOP
      "v_addc_co_u32 %2, vcc, %2, %11, vcc\n\t" P256_NOP
      "v_addc_co_u32 %3, vcc, %3, %12, vcc\n\t" P256_NOP
      "v_addc_co_u32 %4, vcc, %4, %13, vcc\n\t" P256_NOP
      "v_addc_co_u32 %5, vcc, %5, %14, vcc\n\t" P256_NOP
      "v_addc_co_u32 %6, vcc, %6, %15, vcc\n\t" P256_NOP
      "v_addc_co_u32 %7, vcc, %7, %16, vcc\n\t" P256_NOP
      "v_addc_co_u32 %8, vcc, 0, 0, vcc"
      : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "=&v"(c)
      : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
      : "vcc");
  // (c : x) - p, p = {-1, -1, -1, 0, 0, 0, 1, -1}; the final borrow is set iff the sum is below p
  asm("v_subrev_co_u32 %0, vcc, -1, %8\n\t" P256_NOP
      "v_subbrev_co_u32 %1, vcc, -1, %9, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %2, vcc, -1, %10, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %3, vcc, 0, %11, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %4, vcc, 0, %12, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %5, vcc, 0, %13, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %6, vcc, 1, %14, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %7, vcc, -1, %15, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %16, vcc, 0, %16, vcc\n\t" P256_NOP
      "v_cndmask_b32 %8, %0, %8, vcc\n\t"
      "v_cndmask_b32 %9, %1, %9, vcc\n\t"
      "v_cndmask_b32 %10, %2, %10, vcc\n\t"
      "v_cndmask_b32 %11, %3, %11, vcc\n\t"
      "v_cndmask_b32 %12, %4, %12, vcc\n\t"
      "v_cndmask_b32 %13, %5, %13, vcc\n\t"
      "v_cndmask_b32 %14, %6, %14, vcc\n\t"
      "v_cndmask_b32 %15, %7, %15, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+v"(x[0]), "+v"(x[1]),
        "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(c)
      :
      : "vcc");
  return fp256_detail::from_w(x);
}
__device__ __forceinline__ elt32_t fp256_sub(const elt32_t& a, const elt32_t& b) {
  u32 x[8], y[8], m;
  fp256_detail::to_w(a, x);
  fp256_detail::to_w(b, y);
  asm("v_sub_co_u32 %0, vcc, %0, %9\n\t" P256_NOP
      "v_subb_co_u32 %1, vcc, %1, %10, vcc\n\t" P256_NOP
      "v_subb_co_u32 %2, vcc, %2, %11, vcc\n\t" P256_NOP
      "v_subb_co_u32 %3, vcc, %3, %12, vcc\n\t" P256_NOP
      "v_subb_co_u32 %4, vcc, %4, %13, vcc\n\t" P256_NOP
      "v_subb_co_u32 %5, vcc, %5, %14, vcc\n\t" P256_NOP
      "v_subb_co_u32 %6, vcc, %6, %15, vcc\n\t" P256_NOP
      "v_subb_co_u32 %7, vcc, %7, %16, vcc\n\t" P256_NOP
      "v_cndmask_b32 %8, 0, -1, vcc"
      : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "=&v"(m)
      : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7])
      : "vcc");
  const u32 one = m & 1u;  // borrow: add p = {m, m, m, 0, 0, 0, m & 1, m} back
  asm("v_add_co_u32 %0, vcc, %0, %8\n\t" P256_NOP
      "v_addc_co_u32 %1, vcc, %1, %8, vcc\n\t" P256_NOP
      "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t" P256_NOP
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t" P256_NOP
      "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t" P256_NOP
      "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t" P256_NOP
      "v_addc_co_u32 %6, vcc, %6, %9, vcc\n\t" P256_NOP
      "v_addc_co_u32 %7, vcc, %7, %8, vcc"
      : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
      : "v"(m), "v"(one)
      : "vcc");
  return fp256_detail::from_w(x);
}
// x = (c : x) - p if that is not negative (x < 2p with its carry bit c): the second half of fp256_add
__device__ __forceinline__ void p256_cond_sub(u32 (&x)[8], u32 c) {
  u32 d[8];
  // (c : x) - p, p = {-1, -1, -1, 0, 0, 0, 1, -1}; the final borrow is set iff the sum is below p
  asm("v_subrev_co_u32 %0, vcc, -1, %8\n\t" P256_NOP
      "v_subbrev_co_u32 %1, vcc, -1, %9, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %2, vcc, -1, %10, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %3, vcc, 0, %11, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %4, vcc, 0, %12, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %5, vcc, 0, %13, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %6, vcc, 1, %14, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %7, vcc, -1, %15, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %16, vcc, 0, %16, vcc\n\t" P256_NOP
      "v_cndmask_b32 %8, %0, %8, vcc\n\t"
      "v_cndmask_b32 %9, %1, %9, vcc\n\t"
      "v_cndmask_b32 %10, %2, %10, vcc\n\t"
      "v_cndmask_b32 %11, %3, %11, vcc\n\t"
      "v_cndmask_b32 %12, %4, %12, vcc\n\t"
      "v_cndmask_b32 %13, %5, %13, vcc\n\t"
      "v_cndmask_b32 %14, %6, %14, vcc\n\t"
      "v_cndmask_b32 %15, %7, %15, vcc"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "+v"(x[0]), "+v"(x[1]),
        "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(c)
      :
      : "vcc");
}
// Montgomery product on the device: 8 x 8 product scanning (64 v_mad_u64_u32 into a 64-bit column accumulator + carry count, as
// fp_mul in fields.h), then the reduction in FOUR 64-bit steps -- p = -1 (mod 2^64), so the quotient digit m = (T[i+1] : T[i]) is
// the two low limbs themselves and m p = m 2^256 - m 2^224 + m 2^192 + m 2^96 - m only adds and subtracts m at limb offsets
// 3, 6, 8 and 7: one add chain over limbs i+3 .. i+9 and one subtract chain over i+7 .. i+9 per step; what would ripple beyond
// limb i+9 cannot reach a later quotient digit (those are limbs <= 7), so the four carry and four borrow bits are applied
// together at the end.  255 instead of hipcc's 507 instructions for the portable code below (the limb algorithm is restated and
// checked against Python integers in tests/test_p256_cpu.py::test_device_montgomery_limb_algorithm; the instructions themselves by
// the GPU parity tests of the field, tests/test_p256_gpu.py::test_p256_binops with its edge values, and the proof-byte tests).
__device__ __forceinline__ elt32_t fp256_mul(const elt32_t& a, const elt32_t& b) {
  u32 x[8], y[8], T[17];
  fp256_detail::to_w(a, x);
  fp256_detail::to_w(b, y);
  u64 acc = 0;
  u32 ov = 0;
  // one column of the product per statement: its products first, each with the carry out of the 64-bit accumulator in an SGPR
  // pair of its own, then the carries into the count -- a carry link is then separated from its producer by the column's other
  // instructions instead of by the two wait states (s_nop 1) gfx950 wants between a VALU that writes a carry and its consumer
  u64 sc[8];
  FP_MAD(acc, x[0], y[0]);
  FP_COL(T[0], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"
      "v_mad_u64_u32 %0, %3, %6, %7, %0\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1])
      : "v"(x[0]), "v"(y[1]), "v"(x[1]), "v"(y[0])
      : "vcc");
  FP_COL(T[1], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
      "v_mad_u64_u32 %0, %3, %7, %8, %0\n\t"
      "v_mad_u64_u32 %0, %4, %9, %10, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2])
      : "v"(x[0]), "v"(y[2]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[0])
      : "vcc");
  FP_COL(T[2], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %6, %7, %0\n\t"
      "v_mad_u64_u32 %0, %3, %8, %9, %0\n\t"
      "v_mad_u64_u32 %0, %4, %10, %11, %0\n\t"
      "v_mad_u64_u32 %0, %5, %12, %13, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3])
      : "v"(x[0]), "v"(y[3]), "v"(x[1]), "v"(y[2]), "v"(x[2]), "v"(y[1]), "v"(x[3]), "v"(y[0])
      : "vcc");
  FP_COL(T[3], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %7, %8, %0\n\t"
      "v_mad_u64_u32 %0, %3, %9, %10, %0\n\t"
      "v_mad_u64_u32 %0, %4, %11, %12, %0\n\t"
      "v_mad_u64_u32 %0, %5, %13, %14, %0\n\t"
      "v_mad_u64_u32 %0, %6, %15, %16, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %6"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3]), "=&s"(sc[4])
      : "v"(x[0]), "v"(y[4]), "v"(x[1]), "v"(y[3]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[1]), "v"(x[4]), "v"(y[0])
      : "vcc");
  FP_COL(T[4], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %8, %9, %0\n\t"
      "v_mad_u64_u32 %0, %3, %10, %11, %0\n\t"
      "v_mad_u64_u32 %0, %4, %12, %13, %0\n\t"
      "v_mad_u64_u32 %0, %5, %14, %15, %0\n\t"
      "v_mad_u64_u32 %0, %6, %16, %17, %0\n\t"
      "v_mad_u64_u32 %0, %7, %18, %19, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %7"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3]), "=&s"(sc[4]), "=&s"(sc[5])
      : "v"(x[0]), "v"(y[5]), "v"(x[1]), "v"(y[4]), "v"(x[2]), "v"(y[3]), "v"(x[3]), "v"(y[2]), "v"(x[4]), "v"(y[1]), "v"(x[5]), "v"(y[0])
      : "vcc");
  FP_COL(T[5], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %9, %10, %0\n\t"
      "v_mad_u64_u32 %0, %3, %11, %12, %0\n\t"
      "v_mad_u64_u32 %0, %4, %13, %14, %0\n\t"
      "v_mad_u64_u32 %0, %5, %15, %16, %0\n\t"
      "v_mad_u64_u32 %0, %6, %17, %18, %0\n\t"
      "v_mad_u64_u32 %0, %7, %19, %20, %0\n\t"
      "v_mad_u64_u32 %0, %8, %21, %22, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %7\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %8"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3]), "=&s"(sc[4]), "=&s"(sc[5]), "=&s"(sc[6])
      : "v"(x[0]), "v"(y[6]), "v"(x[1]), "v"(y[5]), "v"(x[2]), "v"(y[4]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[2]), "v"(x[5]), "v"(y[1]), "v"(x[6]), "v"(y[0])
      : "vcc");
  FP_COL(T[6], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %10, %11, %0\n\t"
      "v_mad_u64_u32 %0, %3, %12, %13, %0\n\t"
      "v_mad_u64_u32 %0, %4, %14, %15, %0\n\t"
      "v_mad_u64_u32 %0, %5, %16, %17, %0\n\t"
      "v_mad_u64_u32 %0, %6, %18, %19, %0\n\t"
      "v_mad_u64_u32 %0, %7, %20, %21, %0\n\t"
      "v_mad_u64_u32 %0, %8, %22, %23, %0\n\t"
      "v_mad_u64_u32 %0, %9, %24, %25, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %7\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %8\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %9"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3]), "=&s"(sc[4]), "=&s"(sc[5]), "=&s"(sc[6]), "=&s"(sc[7])
      : "v"(x[0]), "v"(y[7]), "v"(x[1]), "v"(y[6]), "v"(x[2]), "v"(y[5]), "v"(x[3]), "v"(y[4]), "v"(x[4]), "v"(y[3]), "v"(x[5]), "v"(y[2]), "v"(x[6]), "v"(y[1]), "v"(x[7]), "v"(y[0])
      : "vcc");
  FP_COL(T[7], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %9, %10, %0\n\t"
      "v_mad_u64_u32 %0, %3, %11, %12, %0\n\t"
      "v_mad_u64_u32 %0, %4, %13, %14, %0\n\t"
      "v_mad_u64_u32 %0, %5, %15, %16, %0\n\t"
      "v_mad_u64_u32 %0, %6, %17, %18, %0\n\t"
      "v_mad_u64_u32 %0, %7, %19, %20, %0\n\t"
      "v_mad_u64_u32 %0, %8, %21, %22, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %7\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %8"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3]), "=&s"(sc[4]), "=&s"(sc[5]), "=&s"(sc[6])
      : "v"(x[1]), "v"(y[7]), "v"(x[2]), "v"(y[6]), "v"(x[3]), "v"(y[5]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[3]), "v"(x[6]), "v"(y[2]), "v"(x[7]), "v"(y[1])
      : "vcc");
  FP_COL(T[8], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %8, %9, %0\n\t"
      "v_mad_u64_u32 %0, %3, %10, %11, %0\n\t"
      "v_mad_u64_u32 %0, %4, %12, %13, %0\n\t"
      "v_mad_u64_u32 %0, %5, %14, %15, %0\n\t"
      "v_mad_u64_u32 %0, %6, %16, %17, %0\n\t"
      "v_mad_u64_u32 %0, %7, %18, %19, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %7"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3]), "=&s"(sc[4]), "=&s"(sc[5])
      : "v"(x[2]), "v"(y[7]), "v"(x[3]), "v"(y[6]), "v"(x[4]), "v"(y[5]), "v"(x[5]), "v"(y[4]), "v"(x[6]), "v"(y[3]), "v"(x[7]), "v"(y[2])
      : "vcc");
  FP_COL(T[9], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %7, %8, %0\n\t"
      "v_mad_u64_u32 %0, %3, %9, %10, %0\n\t"
      "v_mad_u64_u32 %0, %4, %11, %12, %0\n\t"
      "v_mad_u64_u32 %0, %5, %13, %14, %0\n\t"
      "v_mad_u64_u32 %0, %6, %15, %16, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %6"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3]), "=&s"(sc[4])
      : "v"(x[3]), "v"(y[7]), "v"(x[4]), "v"(y[6]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[4]), "v"(x[7]), "v"(y[3])
      : "vcc");
  FP_COL(T[10], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %6, %7, %0\n\t"
      "v_mad_u64_u32 %0, %3, %8, %9, %0\n\t"
      "v_mad_u64_u32 %0, %4, %10, %11, %0\n\t"
      "v_mad_u64_u32 %0, %5, %12, %13, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %5"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2]), "=&s"(sc[3])
      : "v"(x[4]), "v"(y[7]), "v"(x[5]), "v"(y[6]), "v"(x[6]), "v"(y[5]), "v"(x[7]), "v"(y[4])
      : "vcc");
  FP_COL(T[11], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
      "v_mad_u64_u32 %0, %3, %7, %8, %0\n\t"
      "v_mad_u64_u32 %0, %4, %9, %10, %0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1]), "=&s"(sc[2])
      : "v"(x[5]), "v"(y[7]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[5])
      : "vcc");
  FP_COL(T[12], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"
      "v_mad_u64_u32 %0, %3, %6, %7, %0\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %3"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0]), "=&s"(sc[1])
      : "v"(x[6]), "v"(y[7]), "v"(x[7]), "v"(y[6])
      : "vcc");
  FP_COL(T[13], acc, ov);
  asm(
      "v_mad_u64_u32 %0, %2, %3, %4, %0\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %2"
      : "+v"(acc), "+v"(ov), "=&s"(sc[0])
      : "v"(x[7]), "v"(y[7])
      : "vcc");
  FP_COL(T[14], acc, ov);
  T[15] = (u32)acc;
  T[16] = 0;
  u32 cc[4], bb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int i = 2 * s;
    const u32 m0 = T[i], m1 = T[i + 1];
    asm(
      "v_add_co_u32 %0, vcc, %0, %8\n\t" P256_NOP
      "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t" P256_NOP
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t" P256_NOP
      "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t" P256_NOP
      "v_addc_co_u32 %4, vcc, %4, %9, vcc\n\t" P256_NOP
      "v_addc_co_u32 %5, vcc, %5, %8, vcc\n\t" P256_NOP
      "v_addc_co_u32 %6, vcc, %6, %9, vcc\n\t" P256_NOP
      "v_addc_co_u32 %7, vcc, 0, 0, vcc"
        : "+v"(T[i + 3]), "+v"(T[i + 4]), "+v"(T[i + 5]), "+v"(T[i + 6]), "+v"(T[i + 7]), "+v"(T[i + 8]), "+v"(T[i + 9]), "=&v"(cc[s])
        : "v"(m0), "v"(m1)
        : "vcc");
    asm(
      "v_sub_co_u32 %0, vcc, %0, %4\n\t" P256_NOP
      "v_subb_co_u32 %1, vcc, %1, %5, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %2, vcc, 0, %2, vcc\n\t" P256_NOP
      "v_addc_co_u32 %3, vcc, 0, 0, vcc"
        : "+v"(T[i + 7]), "+v"(T[i + 8]), "+v"(T[i + 9]), "=&v"(bb[s])
        : "v"(m0), "v"(m1)
        : "vcc");
  }
  asm(
      "v_add_co_u32 %0, vcc, %0, %7\n\t" P256_NOP
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t" P256_NOP
      "v_addc_co_u32 %2, vcc, %2, %8, vcc\n\t" P256_NOP
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t" P256_NOP
      "v_addc_co_u32 %4, vcc, %4, %9, vcc\n\t" P256_NOP
      "v_addc_co_u32 %5, vcc, 0, %5, vcc\n\t" P256_NOP
      "v_addc_co_u32 %6, vcc, %6, %10, vcc"
      : "+v"(T[10]), "+v"(T[11]), "+v"(T[12]), "+v"(T[13]), "+v"(T[14]), "+v"(T[15]), "+v"(T[16])
      : "v"(cc[0]), "v"(cc[1]), "v"(cc[2]), "v"(cc[3])
      : "vcc");
  asm(
      "v_sub_co_u32 %0, vcc, %0, %7\n\t" P256_NOP
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t" P256_NOP
      "v_subb_co_u32 %2, vcc, %2, %8, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %3, vcc, 0, %3, vcc\n\t" P256_NOP
      "v_subb_co_u32 %4, vcc, %4, %9, vcc\n\t" P256_NOP
      "v_subbrev_co_u32 %5, vcc, 0, %5, vcc\n\t" P256_NOP
      "v_subb_co_u32 %6, vcc, %6, %10, vcc"
      : "+v"(T[10]), "+v"(T[11]), "+v"(T[12]), "+v"(T[13]), "+v"(T[14]), "+v"(T[15]), "+v"(T[16])
      : "v"(bb[0]), "v"(bb[1]), "v"(bb[2]), "v"(bb[3])
      : "vcc");
  u32 r[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) r[k] = T[8 + k];
  p256_cond_sub(r, T[16]);
  return fp256_detail::from_w(r);
}
__host__ inline elt32_t fp256_mul(const elt32_t& a, const elt32_t& b);
// host functions parsed during the device pass resolve to these overloads
__host__ inline elt32_t fp256_add(const elt32_t& a, const elt32_t& b);
__host__ inline elt32_t fp256_sub(const elt32_t& a, const elt32_t& b);
#define P256_HOSTDEV __host__ inline
#else
#define P256_HOSTDEV LF_HD
#endif
P256_HOSTDEV elt32_t fp256_add(const elt32_t& a, const elt32_t& b) {
  u32 x[8], y[8], t[8];
  fp256_detail::to_w(a, x);
  fp256_detail::to_w(b, y);
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (u64)x[i] + y[i];
    t[i] = (u32)c;
    c >>= 32;
  }
  fp256_detail::cond_sub_p(t, (u32)c);
  return fp256_detail::from_w(t);
}
P256_HOSTDEV elt32_t fp256_sub(const elt32_t& a, const elt32_t& b) {
  u32 x[8], y[8], t[8];
  fp256_detail::to_w(a, x);
  fp256_detail::to_w(b, y);
  u64 br = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 d = (u64)x[i] - y[i] - br;
    t[i] = (u32)d;
    br = (d >> 32) & 1u;
  }
  const u32 mask = 0u - (u32)br;  // borrow: add p back
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    c += (u64)t[i] + (fp256_detail::pw(i) & mask);
    t[i] = (u32)c;
    c >>= 32;
  }
  return fp256_detail::from_w(t);
}
LF_HD elt32_t fp256_neg(const elt32_t& a) { return fp256_sub(e32_zero(), a); }

// Montgomery product a * b / 2^256 mod p: operand scanning over 32-bit limbs, one multiplication-free reduction step per
// limb (m = t[0]: t += m * p, then drop the zero limb).
P256_HOSTDEV elt32_t fp256_mul(const elt32_t& a, const elt32_t& b) {
  u32 x[8], y[8];
  fp256_detail::to_w(a, x);
  fp256_detail::to_w(b, y);
  u32 t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // t += x[i] * y
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      c += (u64)x[i] * y[j] + t[j];
      t[j] = (u32)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (u32)c;
    t[9] = (u32)(c >> 32);
    // t += m * p with m = t[0]:  m*p = -m + m*2^96 + m*2^192 - m*2^224 + m*2^256
    const u32 m = t[0];
    // add m at limbs 3, 6, 8 and subtract m at limbs 0, 7 (signed carry chain)
    long long s;
    s = (long long)t[0] - m;  // = 0
    s >>= 32;
    s += t[1]; t[0] = (u32)s; s >>= 32;                       // shifted down by one limb as we go
    s += t[2]; t[1] = (u32)s; s >>= 32;
    s += (long long)t[3] + m; t[2] = (u32)s; s >>= 32;
    s += t[4]; t[3] = (u32)s; s >>= 32;
    s += t[5]; t[4] = (u32)s; s >>= 32;
    s += (long long)t[6] + m; t[5] = (u32)s; s >>= 32;
    s += (long long)t[7] - m; t[6] = (u32)s; s >>= 32;
    s += (long long)t[8] + m; t[7] = (u32)s; s >>= 32;
    s += t[9]; t[8] = (u32)s;
    t[9] = 0;
  }
  u32 r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = t[i];
  fp256_detail::cond_sub_p(r, t[8]);
  return fp256_detail::from_w(r);
}

// canonical value (to_bytes_field = from_montgomery, little-endian bytes): x * 1 / R
LF_HD elt32_t fp256_canon(const elt32_t& a) { return fp256_mul(a, elt32_t{{1, 0, 0, 0}}); }

// ---- Fp2<Fp256Base>, i^2 = -1 (lib/algebra/fp2.h:77-95: Karatsuba, 3 products)
LF_HD fp2_t fp2_add(const fp2_t& a, const fp2_t& b) { return fp2_t{fp256_add(a.re, b.re), fp256_add(a.im, b.im)}; }
LF_HD fp2_t fp2_sub(const fp2_t& a, const fp2_t& b) { return fp2_t{fp256_sub(a.re, b.re), fp256_sub(a.im, b.im)}; }
LF_HD fp2_t fp2_mul(const fp2_t& a, const fp2_t& y) {
  const elt32_t p0 = fp256_mul(a.re, y.re), p1 = fp256_mul(a.im, y.im);
  const elt32_t a01 = fp256_add(a.re, a.im), y01 = fp256_add(y.re, y.im);
  fp2_t r;
  r.re = fp256_sub(p0, p1);
  r.im = fp256_sub(fp256_sub(fp256_mul(a01, y01), p0), p1);
  return r;
}

// ---- device loads / stores (two 16-byte accesses per element)
#if defined(__HIPCC__)
__device__ inline elt32_t ld32(const elt32_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  elt32_t r;
  r.l[0] = (u64)a.x | ((u64)a.y << 32);
  r.l[1] = (u64)a.z | ((u64)a.w << 32);
  r.l[2] = (u64)b.x | ((u64)b.y << 32);
  r.l[3] = (u64)b.z | ((u64)b.w << 32);
  return r;
}
__device__ inline void st32(elt32_t* p, const elt32_t& v) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4((u32)v.l[0], (u32)(v.l[0] >> 32), (u32)v.l[1], (u32)(v.l[1] >> 32));
  q[1] = make_uint4((u32)v.l[2], (u32)(v.l[2] >> 32), (u32)v.l[3], (u32)(v.l[3] >> 32));
}
#endif

LF_HD bool e32_is_zero(const elt32_t& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }

// sum_k acc[k] 2^(32k) mod p for eight 64-bit limb accumulators (sums of 32-bit limbs of canonical residues; images add, so
// the result of summing Montgomery images is the image of the sum).  S = lo + 2^256 hi with hi < 2^40, and
// 2^256 = D = 2^224 - 2^192 - 2^96 + 1 (mod p): two folds of the overflow word with shifts and signed carries
// (lo + hi D < 2^265, then < 2^256 + 2^233 < 2p) and one conditional subtraction -- no product.  (`rsq` is unused; the
// first version computed hi R as the Montgomery image of hi with a full product, and the sumcheck rounds are chains of
// dependent 256-bit products.)
namespace fp256_detail {
LF_HD void fold_overflow(u32 (&t)[8], u32 h0, u32 h1, u32& top) {  // t + (h1 2^32 + h0) D -> t, top (9th limb)
  long long s;
  s = (long long)t[0] + h0;               t[0] = (u32)s; s >>= 32;
  s += (long long)t[1] + h1;              t[1] = (u32)s; s >>= 32;
  s += (long long)t[2];                   t[2] = (u32)s; s >>= 32;
  s += (long long)t[3] - h0;              t[3] = (u32)s; s >>= 32;
  s += (long long)t[4] - h1;              t[4] = (u32)s; s >>= 32;
  s += (long long)t[5];                   t[5] = (u32)s; s >>= 32;
  s += (long long)t[6] - h0;              t[6] = (u32)s; s >>= 32;
  s += (long long)t[7] - h1 + h0;         t[7] = (u32)s; s >>= 32;
  s += (long long)h1;
  top = (u32)s;  // the sum is non-negative and below 2^265
}
}  // namespace fp256_detail
LF_HD elt32_t fp256_reduce_limbs(const u64 acc[8], const elt32_t& rsq) {
  (void)rsq;
  u32 w[8];
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    c += (u64)(u32)acc[k];
    if (k) c += acc[k - 1] >> 32;
    w[k] = (u32)c;
    c >>= 32;
  }
  c += acc[7] >> 32;  // the part above 2^256: < 2^40
  u32 top;
  fp256_detail::fold_overflow(w, (u32)c, (u32)(c >> 32), top);  // < 2^265: top < 2^9
  fp256_detail::fold_overflow(w, top, 0u, top);                  // < 2^256 + 2^233 < 2p: top is 0 or 1
  fp256_detail::cond_sub_p(w, top);
  return fp256_detail::from_w(w);
}

// ---- host-side helpers (FpGeneric: to_montgomery, of_scalar, invertf, of_bytes_field, sample)
inline elt32_t h256_rsq() {  // R^2 mod p
  static const elt32_t v = [] {
    elt32_t x{{1, 0, 0, 0}};
    for (int i = 0; i < 512; ++i) x = fp256_add(x, x);
    return x;
  }();
  return v;
}
inline elt32_t h256_to_mont(const elt32_t& raw) { return fp256_mul(raw, h256_rsq()); }
inline elt32_t h256_of_scalar(u64 u) { return h256_to_mont(elt32_t{{u, 0, 0, 0}}); }
inline elt32_t h256_inv(const elt32_t& x) {  // x^(p-2)
  const u64 e[4] = {0xFFFFFFFFFFFFFFFDull, 0x00000000FFFFFFFFull, 0, 0xFFFFFFFF00000001ull};
  elt32_t r = h256_of_scalar(1), b = x;
  for (int i = 0; i < 256; ++i) {
    if ((e[i / 64] >> (i % 64)) & 1) r = fp256_mul(r, b);
    b = fp256_mul(b, b);
  }
  return r;
}
inline bool h256_fits(const elt32_t& raw) {  // raw < p
  const u64 P[4] = {0xFFFFFFFFFFFFFFFFull, 0x00000000FFFFFFFFull, 0, 0xFFFFFFFF00000001ull};
  for (int i = 3; i >= 0; --i) {
    if (raw.l[i] < P[i]) return true;
    if (raw.l[i] > P[i]) return false;
  }
  return false;
}
// of_bytes_field (fp_generic.h:344-351): 32 little-endian bytes of the canonical value -> Montgomery; false if >= p
inline bool h256_of_bytes(const uint8_t in[32], elt32_t& e) {
  elt32_t raw;
  memcpy(&raw, in, 32);
  if (!h256_fits(raw)) return false;
  e = h256_to_mont(raw);
  return true;
}
inline void h256_to_bytes(const elt32_t& e, uint8_t out[32]) {  // to_bytes_field (:378-380)
  const elt32_t r = fp256_canon(e);
  memcpy(out, &r, 32);
}
// FpGeneric::sample (:360-371): exact_bits = 256, so 32 bytes per attempt, rejected when >= p
template <class Fill>
inline elt32_t h256_sample(Fill fill) {
  for (;;) {
    uint8_t b[32];
    fill(b, 32);
    elt32_t e;
    if (h256_of_bytes(b, e)) return e;
  }
}
// n consecutive samples with few calls of the byte source: the reference draws 32 bytes per attempt and a rejected
// attempt is followed by the next 32 bytes of the same stream, so for every byte-stream RandomEngine (LCG test engines,
// Transcript/FSPRF, SecureRandomEngine) the accepted chunks of one long draw are the same elements in the same order.
template <class Fill>
inline void h256_sample_many(elt32_t* out, size_t n, Fill fill) {
  std::vector<uint8_t> buf;
  size_t have = 0;
  while (have < n) {
    const size_t want = n - have;
    buf.resize(32 * want);
    fill(buf.data(), buf.size());
    for (size_t i = 0; i < want; ++i)
      if (h256_of_bytes(&buf[32 * i], out[have])) ++have;
  }
}
