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

// ===================================================================== Fp128
#define FP_P_LO 0x0000000000000001ull
#define FP_P_HI 0xFFFFF00000000000ull

LF_HD elt_t fp_add(elt_t a, elt_t b) {
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

LF_HD elt_t fp_sub(elt_t a, elt_t b) {
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
LF_HD elt_t fp_mul(elt_t a, elt_t b) {
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

LF_HD elt_t fp_from_mont(elt_t a) { return fp_mul(a, elt_t{1ull, 0ull}); }

// ===================================================================== GF(2^128)
LF_HD elt_t gf_add(elt_t a, elt_t b) { return elt_t{a.lo ^ b.lo, a.hi ^ b.hi}; }

// 32x32 -> 64 carry-less product by Kronecker substitution with 4-bit holes
// (8 bits per masked operand, at most 8 partial products per hole < 16): same
// idea as the reference's portable clmul64_lo (lib/gf2k/sysdep.h:348-358),
// resized to the 32-bit multiplier of CDNA4 (v_mad_u64_u32).
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

// 64x64 -> 128 (Karatsuba over 32-bit halves)
LF_HD void clmul64(u64 x, u64 y, u64& lo, u64& hi) {
  u32 xl = (u32)x, xh = (u32)(x >> 32), yl = (u32)y, yh = (u32)(y >> 32);
  u64 z0 = clmul32(xl, yl);
  u64 z2 = clmul32(xh, yh);
  u64 z1 = clmul32(xl ^ xh, yl ^ yh) ^ z0 ^ z2;
  lo = z0 ^ (z1 << 32);
  hi = z2 ^ (z1 >> 32);
}

// fold a 256-bit carry-less product: x^128 = x^7 + x^2 + x + 1
LF_HD elt_t gf_reduce256(u64 t0, u64 t1, u64 t2, u64 t3) {
  t1 ^= t3 ^ (t3 << 1) ^ (t3 << 2) ^ (t3 << 7);
  t2 ^= (t3 >> 63) ^ (t3 >> 62) ^ (t3 >> 57);
  t0 ^= t2 ^ (t2 << 1) ^ (t2 << 2) ^ (t2 << 7);
  t1 ^= (t2 >> 63) ^ (t2 >> 62) ^ (t2 >> 57);
  return elt_t{t0, t1};
}

// generic 128x128 product (Karatsuba over 64-bit halves: 9 clmul32)
LF_HD elt_t gf_mul(elt_t a, elt_t b) {
  u64 z0l, z0h, z2l, z2h, z1l, z1h;
  clmul64(a.lo, b.lo, z0l, z0h);
  clmul64(a.hi, b.hi, z2l, z2h);
  clmul64(a.lo ^ a.hi, b.lo ^ b.hi, z1l, z1h);
  z1l ^= z0l ^ z2l;
  z1h ^= z0h ^ z2h;
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
LF_HD void sha_compress(sha_state& s, u32 w[16]) {
  u32 a = s.h[0], b = s.h[1], c = s.h[2], d = s.h[3], e = s.h[4], f = s.h[5], g = s.h[6], h = s.h[7];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    u32 wi;
    if (i < 16) {
      wi = w[i];
    } else {
      u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
      u32 s0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3);
      u32 s1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
      wi = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
      w[i & 15] = wi;
    }
    u32 S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25);
    u32 ch = (e & f) ^ (~e & g);
    u32 t1 = h + S1 + ch + kSha256K[i] + wi;
    u32 S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22);
    u32 mj = (a & b) ^ (a & c) ^ (b & c);
    u32 t2 = S0 + mj;
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
  }
  s.h[0] += a; s.h[1] += b; s.h[2] += c; s.h[3] += d;
  s.h[4] += e; s.h[5] += f; s.h[6] += g; s.h[7] += h;
}
LF_HD u32 bswap32(u32 x) { return __builtin_bswap32(x); }
