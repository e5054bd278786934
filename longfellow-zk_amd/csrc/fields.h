// fields.h -- device-side field arithmetic for gfx950 (CDNA4).
//
// Element layout in memory (both fields): 16 bytes = 2 little-endian u64, the
// reference's in-memory Elt:
//   GF2_128: polynomial basis mod x^128+x^7+x^2+x+1, limb 0 = bits 0..63
//            (/root/reference/lib/gf2k/sysdep.h:23-44, gf2_128.h:64-89)
//   Fp128  : p = 2^128 - 2^108 + 1, Montgomery form R = 2^128, canonical (< p)
//            (/root/reference/lib/algebra/fp_p128.h:61-88, fp_generic.h:161-201,484-519)
//
// CDNA4 has no carry-less multiply and 32-bit integer multipliers, so both
// products are built from v_mad_u64_u32 / v_mul_u32_u24-class VALU ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned char u8;

// The arithmetic below is also compiled for the host by tests/host_fields.cc, so that
// the limb-level logic can be unit-tested against the oracle without a GPU.
#define LF_HD __host__ __device__ __forceinline__

struct __attribute__((aligned(16))) elt_t {
  u64 lo, hi;
};

LF_HD elt_t elt_zero() { return elt_t{0ull, 0ull}; }
LF_HD bool elt_eq(elt_t a, elt_t b) { return a.lo == b.lo && a.hi == b.hi; }

// 16-byte vector load/store (global or LDS)
LF_HD elt_t ld16(const elt_t* p) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  return elt_t{((u64)v.y << 32) | v.x, ((u64)v.w << 32) | v.z};
}
LF_HD void st16(elt_t* p, elt_t e) {
  uint4 v;
  v.x = (u32)e.lo;
  v.y = (u32)(e.lo >> 32);
  v.z = (u32)e.hi;
  v.w = (u32)(e.hi >> 32);
  *reinterpret_cast<uint4*>(p) = v;
}

// Payload-before-sequence ordering for words a RUNNING kernel posts to pinned host memory with relaxed system-scope stores:
// the stores are separate vector-memory instructions (the post spans several 64-byte lines) and gfx950 gives no order among
// them, and a workgroup-scope release fence emits NO wait here.  s_waitcnt vmcnt(0) holds the lane until every store it has
// issued is acknowledged (gfx9 counts stores in vmcnt), without the L2 write-back an agent / system release would cost; the
// sequence word is stored only after it.  (ISA pinned by tests/test_abi_and_adapters.py::test_post_publish_waits_for_payload.)
__device__ __forceinline__ void lf_wait_stores_before_publish() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ===================================================================== Fp128
#define FP_P_LO 0x0000000000000001ull
#define FP_P_HI 0xFFFFF00000000000ull

LF_HD elt_t fp_add_c(elt_t a, elt_t b) {
  u64 lo = a.lo + b.lo;
  u64 c = lo < a.lo;
  u64 hi = a.hi + b.hi;
  u64 c2 = hi < a.hi;
  hi += c;
  c2 |= (hi < c);
  // subtract p if overflow or >= p
  u64 slo = lo - FP_P_LO;
  u64 bw = lo < FP_P_LO;
  u64 shi = hi - FP_P_HI - bw;
  bool ge = c2 || (hi > FP_P_HI) || (hi == FP_P_HI && lo >= FP_P_LO);
  return ge ? elt_t{slo, shi} : elt_t{lo, hi};
}

LF_HD elt_t fp_sub_c(elt_t a, elt_t b) {
  u64 lo = a.lo - b.lo;
  u64 bw = a.lo < b.lo;
  u64 hi = a.hi - b.hi - bw;
  bool neg = (a.hi < b.hi) || (a.hi == b.hi && bw);
  if (neg) {
    u64 l2 = lo + FP_P_LO;
    u64 c = l2 < lo;
    hi = hi + FP_P_HI + c;
    lo = l2;
  }
  return elt_t{lo, hi};
}

// 64x64 -> 128
LF_HD void mul64(u64 a, u64 b, u64& lo, u64& hi) {
  lo = a * b;
#if defined(__HIP_DEVICE_COMPILE__)
  hi = __umul64hi(a, b);
#else
  hi = (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// Montgomery product a*b/2^128 mod p.  -p^-1 mod 2^64 = 2^64 - 1 (p = 1 mod 2^64),
// and m*p = m*2^128 - m*2^108 + m needs shifts only (the reference's
// Fp128Reduce::reduction_step, fp_p128.h:68-75).
LF_HD elt_t fp_mul_c(elt_t a, elt_t b) {
  u64 t0, t1, t2, t3, t4 = 0;
  u64 l, h, c;
  // schoolbook
  mul64(a.lo, b.lo, t0, t1);
  mul64(a.hi, b.hi, t2, t3);
  mul64(a.lo, b.hi, l, h);
  t1 += l; c = t1 < l;
  t2 += c; c = t2 < c;
  t2 += h; c += t2 < h;
  t3 += c;
  mul64(a.hi, b.lo, l, h);
  t1 += l; c = t1 < l;
  t2 += c; c = t2 < c;
  t2 += h; c += t2 < h;
  t3 += c;
  // REDC step 1: m = -t0; t += m * p
  {
    u64 m = 0ull - t0;
    // + m at limb 0: t0 + m = 0 with carry (m != 0)
    c = (m != 0);
    // limb 1: add carry, subtract (m << 44)
    u64 s_lo = m << 44, s_hi = m >> 20;
    u64 x = t1 + c; u64 c1 = x < c;
    u64 y = x - s_lo; u64 b1 = x < s_lo;
    t1 = y;
    // limb 2: + m + c1 - s_hi - b1
    u64 x2 = t2 + m; u64 c2 = x2 < m;
    x2 += c1; c2 += x2 < c1;
    u64 y2 = x2 - s_hi; u64 b2 = x2 < s_hi;
    u64 y3 = y2 - b1; b2 += y2 < b1;
    t2 = y3;
    // limb 3: + c2 - b2
    u64 x3 = t3 + c2; u64 c3 = x3 < c2;
    u64 y4 = x3 - b2; u64 b3 = x3 < b2;
    t3 = y4;
    t4 = t4 + c3 - b3;
  }
  // REDC step 2: m = -t1; t += m * p * 2^64
  {
    u64 m = 0ull - t1;
    c = (m != 0);
    u64 s_lo = m << 44, s_hi = m >> 20;
    u64 x = t2 + c; u64 c1 = x < c;
    u64 y = x - s_lo; u64 b1 = x < s_lo;
    t2 = y;
    u64 x2 = t3 + m; u64 c2 = x2 < m;
    x2 += c1; c2 += x2 < c1;
    u64 y2 = x2 - s_hi; u64 b2 = x2 < s_hi;
    u64 y3 = y2 - b1; b2 += y2 < b1;
    t3 = y3;
    t4 = t4 + c2 - b2;
  }
  // result = t4:t3:t2, < 2p
  u64 lo = t2, hi = t3;
  u64 slo = lo - FP_P_LO;
  u64 bw = lo < FP_P_LO;
  u64 shi = hi - FP_P_HI - bw;
  bool ge = (t4 != 0) || (hi > FP_P_HI) || (hi == FP_P_HI && lo >= FP_P_LO);
  return ge ? elt_t{slo, shi} : elt_t{lo, hi};
}


#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 versions: explicit 32-bit limb carry chains (v_add_co/v_addc_co through VCC) and
// v_mad_u64_u32 with its carry-out.  hipcc's own lowering of the portable code above recomputes
// every carry with 64-bit compares (183 VALU instructions per product); these are ~100.
// Each asm statement keeps VCC live only inside itself.  gfx950 needs 2 wait states between a VALU
// that writes VCC and a VALU that reads it (hipcc emits `s_nop 1` in compiled code but never inside an
// asm statement), so every carry link carries its own `s_nop 1`; other waves issue in those slots.
#define FP_W(e, w0, w1, w2, w3) \
  u32 w0 = (u32)(e).lo, w1 = (u32)((e).lo >> 32), w2 = (u32)(e).hi, w3 = (u32)((e).hi >> 32)
#define FP_PACK(w0, w1, w2, w3) elt_t{((u64)(w1) << 32) | (w0), ((u64)(w3) << 32) | (w2)}

__device__ __forceinline__ elt_t fp_add(elt_t a, elt_t b) {
  FP_W(a, a0, a1, a2, a3);
  FP_W(b, b0, b1, b2, b3);
  u32 s0, s1, s2, s3, c, d0, d1, d2, d3;
  asm("v_add_co_u32 %0, vcc, %9, %13\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %1, vcc, %10, %14, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %2, vcc, %11, %15, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %3, vcc, %12, %16, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %4, vcc, 0, 0, vcc\n\t"
      // (c:s) - p, p = {1, 0, 0, 0xfffff000}; final borrow set <=> s < p
      "v_subrev_co_u32 %5, vcc, 1, %0\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %6, vcc, 0, %1, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %7, vcc, 0, %2, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %8, vcc, %3, %17, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %0, %5, %0, vcc\n\t"
      "v_cndmask_b32 %1, %6, %1, vcc\n\t"
      "v_cndmask_b32 %2, %7, %2, vcc\n\t"
      "v_cndmask_b32 %3, %8, %3, vcc"
      : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3), "=&v"(c), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(0xfffff000u)
      : "vcc");
  return FP_PACK(s0, s1, s2, s3);
}

__device__ __forceinline__ elt_t fp_sub(elt_t a, elt_t b) {
  FP_W(a, a0, a1, a2, a3);
  FP_W(b, b0, b1, b2, b3);
  u32 d0, d1, d2, d3, e0, e3;
  asm("v_sub_co_u32 %0, vcc, %6, %10\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %7, %11, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %2, vcc, %8, %12, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %3, vcc, %9, %13, vcc\n\t"
      // borrow => add p back
      "s_nop 1\n\t"
      "v_cndmask_b32 %4, 0, 1, vcc\n\t"
      "v_cndmask_b32 %5, 0, %14, vcc\n\t"
      "v_add_co_u32 %0, vcc, %0, %4\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %3, vcc, %5, %3, vcc"
      : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(e0), "=&v"(e3)
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(0xfffff000u)
      : "vcc");
  return FP_PACK(d0, d1, d2, d3);
}

// acc(64) += x*y, carry-out accumulated into ov
#define FP_MADC(acc, ov, x, y) \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ov) : "v"(x), "v"(y) : "vcc")
#define FP_MAD(acc, x, y) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc")
#define FP_COL(tk, acc, ov)                 \
  tk = (u32)(acc);                          \
  acc = ((acc) >> 32) | ((u64)(ov) << 32);  \
  ov = 0

__device__ __forceinline__ elt_t fp_mul(elt_t a, elt_t b) {
  FP_W(a, a0, a1, a2, a3);
  FP_W(b, b0, b1, b2, b3);
  u32 t0, t1, t2, t3, t4, t5, t6, t7, t8, ov = 0;
  u64 acc = 0;
  // product scanning, 16 x v_mad_u64_u32
  FP_MAD(acc, a0, b0);
  FP_COL(t0, acc, ov);
  FP_MAD(acc, a0, b1);
  FP_MADC(acc, ov, a1, b0);
  FP_COL(t1, acc, ov);
  FP_MADC(acc, ov, a0, b2);
  FP_MADC(acc, ov, a1, b1);
  FP_MADC(acc, ov, a2, b0);
  FP_COL(t2, acc, ov);
  FP_MADC(acc, ov, a0, b3);
  FP_MADC(acc, ov, a1, b2);
  FP_MADC(acc, ov, a2, b1);
  FP_MADC(acc, ov, a3, b0);
  FP_COL(t3, acc, ov);
  FP_MADC(acc, ov, a1, b3);
  FP_MADC(acc, ov, a2, b2);
  FP_MADC(acc, ov, a3, b1);
  FP_COL(t4, acc, ov);
  FP_MADC(acc, ov, a2, b3);
  FP_MADC(acc, ov, a3, b2);
  FP_COL(t5, acc, ov);
  FP_MAD(acc, a3, b3);
  t6 = (u32)acc;
  t7 = (u32)(acc >> 32);
#ifndef LF_FP_REDC2
  // REDC in ONE 128-bit step.  p = 1 - 2^108 (mod 2^128) and 2^216 = 0 (mod 2^128), so p^-1 = 1 + 2^108 (mod 2^128):
  //   m = -T_lo p^-1 = -(T_lo + (T_lo << 108)) = -(t0, t1, t2, t3 + (t0 << 12))   (mod 2^128)
  //   (T + m p) / 2^128 = T_hi + m - (m >> 20) + delta,   delta = carry-out of T_lo + m
  // (m p = m 2^128 - m 2^108 + m; the low halves cancel up to that carry).  32 instructions instead of the 46 of two
  // 64-bit steps; the result is < 2p as before.
  const u32 u3 = t3 + (t0 << 12);
  u32 m0, m1, m2, m3;
  asm("v_sub_co_u32 %0, vcc, 0, %4\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %2, vcc, 0, %6, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %3, vcc, 0, %7, vcc"
      : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
      : "v"(t0), "v"(t1), "v"(t2), "v"(u3)
      : "vcc");
  const u32 s0 = __builtin_amdgcn_alignbit(m1, m0, 20), s1 = __builtin_amdgcn_alignbit(m2, m1, 20), s2 = __builtin_amdgcn_alignbit(m3, m2, 20),
            s3 = m3 >> 20;
  u32 q0, q1, q2, q3, x;
  asm("v_sub_co_u32 %0, vcc, %10, %14\n\t"        // q = m - (m >> 20): never borrows out
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %11, %15, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %2, vcc, %12, %16, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %3, vcc, %13, %17, vcc\n\t"
      "v_add_co_u32 %4, vcc, %18, %10\n\t"         // delta = carry-out of T_lo + m (sums discarded)
      "s_nop 1\n\t"
      "v_addc_co_u32 %4, vcc, %19, %11, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %4, vcc, %20, %12, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %4, vcc, %21, %13, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %5, vcc, %5, %0, vcc\n\t"     // (t8:t7..t4) = T_hi + q + delta
      "s_nop 1\n\t"
      "v_addc_co_u32 %6, vcc, %6, %1, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %7, vcc, %7, %2, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %8, vcc, %8, %3, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %9, vcc, 0, 0, vcc"
      : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(x), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7), "=&v"(t8)
      : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(t0), "v"(t1), "v"(t2), "v"(t3)
      : "vcc");
  u32 d0, d1, d2, d3;
  // (t8:t7..t4) - p ; final borrow <=> value < p
  asm("v_subrev_co_u32 %0, vcc, 1, %5\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %6, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %2, vcc, 0, %7, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %3, vcc, %8, %9, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %0, %0, %5, vcc\n\t"
      "v_cndmask_b32 %1, %1, %6, vcc\n\t"
      "v_cndmask_b32 %2, %2, %7, vcc\n\t"
      "v_cndmask_b32 %3, %3, %8, vcc"
      : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "+v"(t8)
      : "v"(t4), "v"(t5), "v"(t6), "v"(t7), "v"(0xfffff000u)
      : "vcc");
  return FP_PACK(d0, d1, d2, d3);
}
#else
  // REDC, two 64-bit steps: m = -(t[k+1]:t[k]);  t += m*2^(32k) * (2^128 - 2^108 + 1)
  u32 m0, m1, s3, s4, s5;
  asm("v_sub_co_u32 %0, vcc, 0, %9\n\t"          // m = 0 - (t1:t0); borrow-out = (t != 0) = carry into limb 2
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, 0, %10, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %4, vcc, %0, %4, vcc\n\t"   // + m * 2^128
      "s_nop 1\n\t"
      "v_addc_co_u32 %5, vcc, %1, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %7, vcc, 0, %7, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %8, vcc, 0, 0, vcc"
      : "=&v"(m0), "=&v"(m1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7), "=&v"(t8)
      : "v"(t0), "v"(t1)
      : "vcc");
  s3 = m0 << 12;
  s4 = __builtin_amdgcn_alignbit(m1, m0, 20);
  s5 = m1 >> 20;
  asm("v_sub_co_u32 %0, vcc, %0, %6\n\t"           // - m * 2^108
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %1, %7, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %2, vcc, %2, %8, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %5, vcc, 0, %5, vcc"
      : "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7), "+v"(t8)
      : "v"(s3), "v"(s4), "v"(s5)
      : "vcc");
  asm("v_sub_co_u32 %0, vcc, 0, %7\n\t"           // m = 0 - (t3:t2)
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, 0, %8, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %4, vcc, %0, %4, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %5, vcc, %1, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %6, vcc, 0, %6, vcc"
      : "=&v"(m0), "=&v"(m1), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7), "+v"(t8)
      : "v"(t2), "v"(t3)
      : "vcc");
  s3 = m0 << 12;
  s4 = __builtin_amdgcn_alignbit(m1, m0, 20);
  s5 = m1 >> 20;
  u32 d0, d1, d2, d3;
  asm("v_sub_co_u32 %0, vcc, %0, %8\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %2, vcc, %2, %10, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %3, vcc, 0, %3, vcc\n\t"
      // (t8:t7..t4) - p ; final borrow <=> value < p
      "v_subrev_co_u32 %4, vcc, 1, %11\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %5, vcc, 0, %0, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %6, vcc, 0, %1, vcc\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %7, vcc, %2, %12, vcc\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %4, %4, %11, vcc\n\t"
      "v_cndmask_b32 %5, %5, %0, vcc\n\t"
      "v_cndmask_b32 %6, %6, %1, vcc\n\t"
      "v_cndmask_b32 %7, %7, %2, vcc"
      : "+v"(t5), "+v"(t6), "+v"(t7), "+v"(t8), "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
      : "v"(s3), "v"(s4), "v"(s5), "v"(t4), "v"(0xfffff000u)
      : "vcc");
  return FP_PACK(d0, d1, d2, d3);
}
#endif
// host functions parsed during the device pass resolve to these overloads
__host__ inline elt_t fp_add(elt_t a, elt_t b) { return fp_add_c(a, b); }
__host__ inline elt_t fp_sub(elt_t a, elt_t b) { return fp_sub_c(a, b); }
__host__ inline elt_t fp_mul(elt_t a, elt_t b) { return fp_mul_c(a, b); }
#else
LF_HD elt_t fp_add(elt_t a, elt_t b) { return fp_add_c(a, b); }
LF_HD elt_t fp_sub(elt_t a, elt_t b) { return fp_sub_c(a, b); }
LF_HD elt_t fp_mul(elt_t a, elt_t b) { return fp_mul_c(a, b); }
#endif

LF_HD elt_t fp_from_mont(elt_t a) { return fp_mul(a, elt_t{1ull, 0ull}); }

// sum_k a_k 2^(32k) mod p for four u64 limb accumulators (sums of 32-bit limbs of residues: the integer-limb
// accumulators of the Fp128 scatters).  Plain integer folding with 2^128 = 2^108 - 1 (mod p): the value is < 2^160,
// so one fold leaves < 2^141 and a second one < 2^128; three residues are then added.  ~40 integer ops instead of
// the eight Montgomery products of a limb-by-limb recombination.  (A sum of Montgomery images is the image of the sum.)
LF_HD elt_t fp_canon128(elt_t x) {  // x < 2^128 < 2p  ->  x mod p
  const bool ge = x.hi > FP_P_HI || (x.hi == FP_P_HI && x.lo >= FP_P_LO);
  if (ge) {
    const u64 lo = x.lo - FP_P_LO;
    const u64 bw = x.lo < FP_P_LO;
    x = elt_t{lo, x.hi - FP_P_HI - bw};
  }
  return x;
}
LF_HD elt_t fp_reduce_limbs(u64 a0, u64 a1, u64 a2, u64 a3) {
  // S = (w2 : w1 : w0) in 64-bit words
  const u64 w0 = a0 + (a1 << 32);
  const u64 c0 = w0 < a0;
  u64 w1 = a2 + ((a1 >> 32) + c0);
  u64 c1 = w1 < a2;
  const u64 t = a3 << 32;
  w1 += t;
  c1 += w1 < t;
  const u64 w2 = (a3 >> 32) + c1;  // < 2^32 + 2
  const elt_t x = fp_canon128(elt_t{w0, w1});
  // w2 * 2^128 = w2 * (2^108 - 1) = ytop 2^128 + (yh 2^64) - w2
  const u64 yh = w2 << 44;
  u64 ytop = w2 >> 20;
  const u64 br = w2 != 0;
  if (yh < br) ytop -= 1;  // borrow out of the low 128 bits (then w2 >= 2^20, so ytop >= 1)
  const elt_t y = fp_canon128(elt_t{0ull - w2, yh - br});
  // ytop * 2^128 = ytop * (2^108 - 1) < 2^121
  const elt_t z{0ull - ytop, (ytop << 44) - (u64)(ytop != 0)};
  return fp_add(fp_add(x, y), z);
}

// ===================================================================== F64 and F64_2
// F64 = Fp<1> with p = 2^64 - 2^32 + 1 (lib/algebra/fp_generic.h; the prime of lib/algebra/fft_test.cc:205-229), values
// in Montgomery form with R = 2^64, canonical in [0, p).  F64_2 = Fp2<F64> with i^2 = -1 (lib/algebra/fp2.h:36-52):
// elt_t{lo = re, hi = im}, the memory image of Fp2<Fp<1>>::Elt.
#define F64_P 0xFFFFFFFF00000001ull
LF_HD u64 f64_add_c(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a || s >= F64_P) s -= F64_P;
  return s;
}
LF_HD u64 f64_sub_c(u64 a, u64 b) {
  u64 d = a - b;
  if (a < b) d += F64_P;
  return d;
}
// Montgomery product a b 2^-64 mod p.  p^-1 = 2^32 + 1 (mod 2^64), so m = lo (2^32 + 1) has m p = lo (mod 2^64) and
// (a b - m p) / 2^64 = hi - floor(m p / 2^64) exactly; with m p = m 2^64 - m (2^32 - 1) the floor needs shifts only.
LF_HD u64 f64_mul_c(u64 a, u64 b) {
  u64 lo, hi;
  mul64(a, b, lo, hi);
  const u64 m = lo + (lo << 32);
  const u64 s = m << 32;                                         // m (2^32 - 1) = ((m >> 32) - (s < m)) 2^64 + (s - m)
  const u64 mh = m - (m >> 32) + (u64)(s < m) - (u64)(lo != 0);  // floor(m p / 2^64); s - m = -lo (mod 2^64)
  u64 t = hi - mh;
  if (hi < mh) t += F64_P;
  return t;
}
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 versions on 32-bit limbs (see the Fp128 ones above for the VCC wait states).  hipcc's lowering of the portable code
// takes ~45 VALU instructions per product (64-bit compares and selects for every carry); these take 14, 5 and 7.
__device__ __forceinline__ u64 f64_sub(u64 a, u64 b) {  // a - b, then - (2^32 - 1) (= + p mod 2^64) on borrow
  u32 d0, d1, m;
  asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %2, 0, -1, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %2\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(d0), "=&v"(d1), "=&v"(m)
      : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
      : "vcc");
  return ((u64)d1 << 32) | d0;
}
__device__ __forceinline__ u64 f64_add(u64 a, u64 b) {  // a - (p - b): p - b needs no reduction and b = 0 comes out right
  u32 n0, n1;
  asm("v_sub_co_u32 %0, vcc, 1, %2\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, -1, %3, vcc"
      : "=&v"(n0), "=&v"(n1)
      : "v"((u32)b), "v"((u32)(b >> 32))
      : "vcc");
  return f64_sub(a, ((u64)n1 << 32) | n0);
}
__device__ __forceinline__ u64 f64_mul(u64 a, u64 b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 t0 = 0;
  FP_MAD(t0, a0, b0);
  u64 t1 = t0 >> 32;
  FP_MAD(t1, a0, b1);  // < 2^64: (2^32 - 1)^2 + 2^32 - 1
  u64 t2 = (u32)t1;
  FP_MAD(t2, a1, b0);
  u64 t3 = (t1 >> 32) + (t2 >> 32);
  FP_MAD(t3, a1, b1);
  // x = (l0, l1, h0, h1).  With a = lo + (lo << 32) (carry e) and b = a - (a >> 32) - e the result is hi - b, plus p on borrow:
  // the same m and floor(m p / 2^64) as f64_mul_c, the carry e standing for its two comparisons.
  u32 r0, r1, x, y, z;
  asm("v_add_co_u32 %2, vcc, %6, %5\n\t"       // x = a1 = l1 + l0, e
      "s_nop 1\n\t"
      "v_subb_co_u32 %3, vcc, %5, %2, vcc\n\t"  // y = b0 = l0 - a1 - e
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %2, vcc, 0, %2, vcc\n\t"  // x = b1 = a1 - borrow
      "v_sub_co_u32 %0, vcc, %7, %3\n\t"        // r = hi - b
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %8, %2, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %4, 0, -1, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %4\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(x), "=&v"(y), "=&v"(z)
      : "v"((u32)t0), "v"((u32)t2), "v"((u32)t3), "v"((u32)(t3 >> 32))
      : "vcc");
  return ((u64)r1 << 32) | r0;
}
// host functions parsed during the device pass resolve to these overloads
__host__ inline u64 f64_add(u64 a, u64 b) { return f64_add_c(a, b); }
__host__ inline u64 f64_sub(u64 a, u64 b) { return f64_sub_c(a, b); }
__host__ inline u64 f64_mul(u64 a, u64 b) { return f64_mul_c(a, b); }
#else
LF_HD u64 f64_add(u64 a, u64 b) { return f64_add_c(a, b); }
LF_HD u64 f64_sub(u64 a, u64 b) { return f64_sub_c(a, b); }
LF_HD u64 f64_mul(u64 a, u64 b) { return f64_mul_c(a, b); }
#endif
LF_HD elt_t f64x2_add(elt_t a, elt_t b) { return elt_t{f64_add(a.lo, b.lo), f64_add(a.hi, b.hi)}; }
LF_HD elt_t f64x2_sub(elt_t a, elt_t b) { return elt_t{f64_sub(a.lo, b.lo), f64_sub(a.hi, b.hi)}; }
LF_HD elt_t f64x2_mul(elt_t a, elt_t b) {  // Fp2::mul (fp2.h:87-101): three base-field products
  const u64 p0 = f64_mul(a.lo, b.lo), p1 = f64_mul(a.hi, b.hi);
  const u64 x = f64_mul(f64_add(a.lo, a.hi), f64_add(b.lo, b.hi));
  return elt_t{f64_sub(p0, p1), f64_sub(f64_sub(x, p0), p1)};
}
LF_HD elt_t f64x2_mul_real(elt_t a, u64 y) { return elt_t{f64_mul(a.lo, y), f64_mul(a.hi, y)}; }  // Fp2::mul(Elt, Scalar) (:102-105)

// ===================================================================== GF(2^128)
LF_HD elt_t gf_add(elt_t a, elt_t b) { return elt_t{a.lo ^ b.lo, a.hi ^ b.hi}; }

// 32x32 -> 64 carry-less product by Kronecker substitution with 4-bit holes
// (8 bits per masked operand, at most 8 partial products per hole < 16): same
// idea as the reference's portable clmul64_lo (lib/gf2k/sysdep.h:348-358),
// resized to the 32-bit multiplier of CDNA4 (v_mad_u64_u32).
#if defined(__HIP_DEVICE_COMPILE__)
// gfx950: the sums of four partial products and the final interleave through the three-input boolean instruction
// (v_bitop3_b32: 0x96 = a ^ b ^ c, 0xCA = bit select src0 ? src1 : src2) -- 46 instead of 62 instructions per 32 x 32 product
__device__ __forceinline__ u64 clmul_x4(u64 p, u64 q, u64 r, u64 s) {
  const u32 lo = __builtin_amdgcn_bitop3_b32((u32)p, (u32)q, (u32)r, 0x96) ^ (u32)s;
  const u32 hi = __builtin_amdgcn_bitop3_b32((u32)(p >> 32), (u32)(q >> 32), (u32)(r >> 32), 0x96) ^ (u32)(s >> 32);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 clmul_sel4(u32 z0, u32 z1, u32 z2, u32 z3) {  // bit i of every nibble from z_i
  const u32 t01 = __builtin_amdgcn_bitop3_b32(0x55555555u, z0, z1, 0xCA), t23 = __builtin_amdgcn_bitop3_b32(0x55555555u, z2, z3, 0xCA);
  return __builtin_amdgcn_bitop3_b32(0x33333333u, t01, t23, 0xCA);
}
__device__ __forceinline__ u64 clmul32(u32 x, u32 y) {
  const u32 m0 = 0x11111111u, m1 = 0x22222222u, m2 = 0x44444444u, m3 = 0x88888888u;
  const u32 x0 = x & m0, x1 = x & m1, x2 = x & m2, x3 = x & m3;
  const u32 y0 = y & m0, y1 = y & m1, y2 = y & m2, y3 = y & m3;
  const u64 z0 = clmul_x4((u64)x0 * y0, (u64)x1 * y3, (u64)x2 * y2, (u64)x3 * y1);
  const u64 z1 = clmul_x4((u64)x0 * y1, (u64)x1 * y0, (u64)x2 * y3, (u64)x3 * y2);
  const u64 z2 = clmul_x4((u64)x0 * y2, (u64)x1 * y1, (u64)x2 * y0, (u64)x3 * y3);
  const u64 z3 = clmul_x4((u64)x0 * y3, (u64)x1 * y2, (u64)x2 * y1, (u64)x3 * y0);
  const u32 lo = clmul_sel4((u32)z0, (u32)z1, (u32)z2, (u32)z3);
  const u32 hi = clmul_sel4((u32)(z0 >> 32), (u32)(z1 >> 32), (u32)(z2 >> 32), (u32)(z3 >> 32));
  return ((u64)hi << 32) | lo;
}
// host functions parsed during the device pass resolve to this overload
__host__ inline u64 clmul32(u32 x, u32 y) {
  const u32 m0 = 0x11111111u, m1 = 0x22222222u, m2 = 0x44444444u, m3 = 0x88888888u;
  u32 x0 = x & m0, x1 = x & m1, x2 = x & m2, x3 = x & m3;
  u32 y0 = y & m0, y1 = y & m1, y2 = y & m2, y3 = y & m3;
  u64 z0 = ((u64)x0 * y0) ^ ((u64)x1 * y3) ^ ((u64)x2 * y2) ^ ((u64)x3 * y1);
  u64 z1 = ((u64)x0 * y1) ^ ((u64)x1 * y0) ^ ((u64)x2 * y3) ^ ((u64)x3 * y2);
  u64 z2 = ((u64)x0 * y2) ^ ((u64)x1 * y1) ^ ((u64)x2 * y0) ^ ((u64)x3 * y3);
  u64 z3 = ((u64)x0 * y3) ^ ((u64)x1 * y2) ^ ((u64)x2 * y1) ^ ((u64)x3 * y0);
  return (z0 & 0x1111111111111111ull) | (z1 & 0x2222222222222222ull) | (z2 & 0x4444444444444444ull) |
         (z3 & 0x8888888888888888ull);
}
#else
LF_HD u64 clmul32(u32 x, u32 y) {
  const u32 m0 = 0x11111111u, m1 = 0x22222222u, m2 = 0x44444444u, m3 = 0x88888888u;
  u32 x0 = x & m0, x1 = x & m1, x2 = x & m2, x3 = x & m3;
  u32 y0 = y & m0, y1 = y & m1, y2 = y & m2, y3 = y & m3;
  u64 z0 = ((u64)x0 * y0) ^ ((u64)x1 * y3) ^ ((u64)x2 * y2) ^ ((u64)x3 * y1);
  u64 z1 = ((u64)x0 * y1) ^ ((u64)x1 * y0) ^ ((u64)x2 * y3) ^ ((u64)x3 * y2);
  u64 z2 = ((u64)x0 * y2) ^ ((u64)x1 * y1) ^ ((u64)x2 * y0) ^ ((u64)x3 * y3);
  u64 z3 = ((u64)x0 * y3) ^ ((u64)x1 * y2) ^ ((u64)x2 * y1) ^ ((u64)x3 * y0);
  return (z0 & 0x1111111111111111ull) | (z1 & 0x2222222222222222ull) | (z2 & 0x4444444444444444ull) |
         (z3 & 0x8888888888888888ull);
}
#endif

// a ^ b ^ c on 64-bit words: two v_bitop3_b32 on the device
LF_HD u64 gf_x3(u64 a, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  const u32 lo = __builtin_amdgcn_bitop3_b32((u32)a, (u32)b, (u32)c, 0x96);
  const u32 hi = __builtin_amdgcn_bitop3_b32((u32)(a >> 32), (u32)(b >> 32), (u32)(c >> 32), 0x96);
  return ((u64)hi << 32) | lo;
#else
  return a ^ b ^ c;
#endif
}

// 64x64 -> 128 (Karatsuba over 32-bit halves)
LF_HD void clmul64(u64 x, u64 y, u64& lo, u64& hi) {
  u32 xl = (u32)x, xh = (u32)(x >> 32), yl = (u32)y, yh = (u32)(y >> 32);
  u64 z0 = clmul32(xl, yl);
  u64 z2 = clmul32(xh, yh);
  u64 z1 = gf_x3(clmul32(xl ^ xh, yl ^ yh), z0, z2);
  lo = z0 ^ (z1 << 32);
  hi = z2 ^ (z1 >> 32);
}

// fold a 256-bit carry-less product: x^128 = x^7 + x^2 + x + 1
LF_HD elt_t gf_reduce256(u64 t0, u64 t1, u64 t2, u64 t3) {
  t1 = gf_x3(gf_x3(t1, t3, t3 << 1), t3 << 2, t3 << 7);
  t2 ^= gf_x3(t3 >> 63, t3 >> 62, t3 >> 57);
  t0 = gf_x3(gf_x3(t0, t2, t2 << 1), t2 << 2, t2 << 7);
  t1 ^= gf_x3(t2 >> 63, t2 >> 62, t2 >> 57);
  return elt_t{t0, t1};
}

// generic 128x128 product (Karatsuba over 64-bit halves: 9 clmul32)
LF_HD elt_t gf_mul(elt_t a, elt_t b) {
  u64 z0l, z0h, z2l, z2h, z1l, z1h;
  clmul64(a.lo, b.lo, z0l, z0h);
  clmul64(a.hi, b.hi, z2l, z2h);
  clmul64(a.lo ^ a.hi, b.lo ^ b.hi, z1l, z1h);
  z1l = gf_x3(z1l, z0l, z2l);
  z1h = gf_x3(z1h, z0h, z2h);
  return gf_reduce256(z0l, z0h ^ z1l, z2l ^ z1h, z2h);
}

// ===================================================================== field tag dispatch
enum { FIELD_GF2_128 = 4, FIELD_FP128 = 6 };  // FieldID, lib/proto/circuit_io.h:24-36

template <int F>
struct Fld;
template <>
struct Fld<FIELD_GF2_128> {
  static LF_HD elt_t add(elt_t a, elt_t b) { return gf_add(a, b); }
  static LF_HD elt_t sub(elt_t a, elt_t b) { return gf_add(a, b); }
  static LF_HD elt_t mul(elt_t a, elt_t b) { return gf_mul(a, b); }
  static LF_HD elt_t canon(elt_t a) { return a; }  // to_bytes_field image
};
template <>
struct Fld<FIELD_FP128> {
  static LF_HD elt_t add(elt_t a, elt_t b) { return fp_add(a, b); }
  static LF_HD elt_t sub(elt_t a, elt_t b) { return fp_sub(a, b); }
  static LF_HD elt_t mul(elt_t a, elt_t b) { return fp_mul(a, b); }
  static LF_HD elt_t canon(elt_t a) { return fp_from_mont(a); }
};

// ===================================================================== SHA-256
LF_HD u32 ror32(u32 x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, n);
#else
  return (x >> n) | (x << (32 - n));
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
static __constant__ u32 kSha256K[64] = {
#else
static const u32 kSha256K[64] = {
#endif
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

struct sha_state {
  u32 h[8];
};
LF_HD void sha_init(sha_state& s) {
  s.h[0] = 0x6a09e667; s.h[1] = 0xbb67ae85; s.h[2] = 0x3c6ef372; s.h[3] = 0xa54ff53a;
  s.h[4] = 0x510e527f; s.h[5] = 0x9b05688c; s.h[6] = 0x1f83d9ab; s.h[7] = 0x5be0cd19;
}
// one compression; w[16] is the big-endian-decoded message block (clobbered)
// three-input boolean operations: gfx950 has one instruction for any of them (v_bitop3_b32); hipcc finds it for Ch but
// spends two v_xor_b32 on every three-way XOR and an extra v_and_b32 on Maj
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u32 sha_xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 sha_maj(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
#else
inline u32 sha_xor3(u32 a, u32 b, u32 c) { return a ^ b ^ c; }
inline u32 sha_maj(u32 a, u32 b, u32 c) { return (a & b) ^ (a & c) ^ (b & c); }
#endif
LF_HD void sha_compress(sha_state& s, u32 w[16]) {
  u32 a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    u32 wi;
    if (i < 16) {
      wi = w[i];
    } else {
      u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      u32 s0 = sha_xor3(ror32(w15, 7), ror32(w15, 18), w15 >> 3);
      u32 s1 = sha_xor3(ror32(w2, 17), ror32(w2, 19), w2 >> 10);
      wi = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
      w[i & 15] = wi;
    }
    u32 S1 = sha_xor3(ror32(e, 6), ror32(e, 11), ror32(e, 25));
    u32 ch = (e & f) ^ (~e & g);
    u32 t1 = h + S1 + ch + kSha256K[i] + wi;
    u32 S0 = sha_xor3(ror32(a, 2), ror32(a, 13), ror32(a, 22));
    u32 mj = sha_maj(a, b, c);
    u32 t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d;
  s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
}
LF_HD u32 bswap32(u32 x) { return __builtin_bswap32(x); }
