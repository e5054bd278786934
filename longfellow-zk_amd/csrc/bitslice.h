// bitslice.h -- bit-sliced tower arithmetic for GF(2^128) over the LCH14 subfield GF(2^m).
//
// Layout: a "unit" is one tower coordinate (m planes) of one tableau column for 32 batch rows:
// word j of the unit holds, in bit r, the coefficient of h^j of that coordinate for row r.
// Multiplying 32 rows by the same subfield constant t (an LCH14 twiddle, lch14.h:81-100) is then
//     acc ^= cur   for every set bit k of t,   cur <- cur * h  (3 XORs: mu is a pentanomial)
// i.e. ~m/2 * m + 3m full-width XORs per 32 element-coordinates -- no multiplier, no carry-less
// multiply, exact.  Basis and conversion programs: tools/gen_tower.py -> tower_k{4,5}.h.
#pragma once
#include "fields.h"
#include "tower_k4.h"
#include "tower_k5.h"

template <int K>
struct Tower;
template <>
struct Tower<4> {
  static constexpr int M = TOWER_K4_M, D = TOWER_K4_D;
  static constexpr u32 MU_LOW = TOWER_K4_MU_LOW;
};
template <>
struct Tower<5> {
  static constexpr int M = TOWER_K5_M, D = TOWER_K5_D;
  static constexpr u32 MU_LOW = TOWER_K5_MU_LOW;
};

// cur <- cur * h in GF(2)[h]/mu  (plane renaming + one XOR per middle tap of mu)
template <int M, u32 MU_LOW>
LF_HD void bs_mulh(u32 (&cur)[M]) {
  const u32 top = cur[M - 1];
#pragma unroll
  for (int j = M - 1; j > 0; --j) cur[j] = ((MU_LOW >> j) & 1u) ? (cur[j - 1] ^ top) : cur[j - 1];
  cur[0] = top;
}

// dst ^= t * b,  t the same for every lane of the wave (scalar branches on its bits).  Two bits of t per step: when both
// are set, b h^k and b h^(k+1) join dst in ONE three-input XOR per plane (v_bitop3_b32 on gfx950; hipcc does not form it
// from two XORs) -- 24 instead of 32 plane operations per pair of bits on average.
template <int M, u32 MU_LOW>
LF_HD void bs_mac_uniform(u32 t, const u32 (&b)[M], u32 (&dst)[M]) {
  static_assert(M % 2 == 0, "two bits per step");
  u32 cur[M];
#pragma unroll
  for (int j = 0; j < M; ++j) cur[j] = b[j];
#pragma unroll
  for (int k = 0; k < M; k += 2) {
    const u32 two = (t >> k) & 3u;
    if (two == 3u) {
      u32 nxt[M];
#pragma unroll
      for (int j = 0; j < M; ++j) nxt[j] = cur[j];
      bs_mulh<M, MU_LOW>(nxt);
#pragma unroll
      for (int j = 0; j < M; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        dst[j] = __builtin_amdgcn_bitop3_b32(dst[j], cur[j], nxt[j], 0x96);
#else
        dst[j] ^= cur[j] ^ nxt[j];
#endif
      }
#pragma unroll
      for (int j = 0; j < M; ++j) cur[j] = nxt[j];
    } else {
      if (two == 1u) {
#pragma unroll
        for (int j = 0; j < M; ++j) dst[j] ^= cur[j];
      }
      bs_mulh<M, MU_LOW>(cur);
      if (two == 2u) {
#pragma unroll
        for (int j = 0; j < M; ++j) dst[j] ^= cur[j];
      }
    }
    if (k + 2 < M) bs_mulh<M, MU_LOW>(cur);
  }
}

// dst ^= t * b with a per-lane t (no branches: every step is masked)
template <int M, u32 MU_LOW>
LF_HD void bs_mac_lane(u32 t, const u32 (&b)[M], u32 (&dst)[M]) {
  u32 cur[M];
#pragma unroll
  for (int j = 0; j < M; ++j) cur[j] = b[j];
#pragma unroll
  for (int k = 0; k < M; ++k) {
    const u32 mask = 0u - ((t >> k) & 1u);
#pragma unroll
    for (int j = 0; j < M; ++j) dst[j] ^= cur[j] & mask;
    if (k + 1 < M) bs_mulh<M, MU_LOW>(cur);
  }
}

#if defined(__HIPCC__)
// dst ^= t * b where t is constant on aligned groups of 2^glog lanes (glog <= 6): one scalar-branched pass per
// group with the other lanes masked off.  glog = 6 is the wave-uniform case (one pass); smaller groups cost
// (64 >> glog) passes, still ~2x cheaper than masking every plane XOR per lane (bs_mac_lane).
template <int M, u32 MU_LOW>
__device__ inline void bs_mac_groups(u32 t, u32 glog, const u32 (&b)[M], u32 (&dst)[M]) {
  const u32 lane = __lane_id();
#pragma nounroll
  for (u32 g = 0; g < (64u >> glog); ++g) {
    const u32 tg = __builtin_amdgcn_readlane(t, g << glog);
    if ((lane >> glog) == g) bs_mac_uniform<M, MU_LOW>(tg, b, dst);
  }
}
#endif

// 32x32 bit-matrix transpose: on return bit r of x[b] = bit b of the original x[r].
LF_HD void bs_transpose32(u32 (&x)[32]) {
#pragma unroll
  for (int s = 16; s >= 1; s >>= 1) {
    const u32 m = s == 16 ? 0x0000FFFFu : s == 8 ? 0x00FF00FFu : s == 4 ? 0x0F0F0F0Fu : s == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (k & s) continue;
      // swap the (row-low, col-high) block with the (row-high, col-low) block
      const u32 a = x[k], b = x[k + s];
      x[k] = (a & m) | ((b & m) << s);
      x[k + s] = ((a >> s) & m) | (b & ~m);
    }
  }
}

// subfield coordinates (bits of t in the basis h^j) of a twiddle given in the polynomial basis
template <int K>
inline u32 tower_twiddle_bits(u64 lo, u64 hi);
template <>
inline u32 tower_twiddle_bits<4>(u64 lo, u64 hi) {
  u32 t = 0;
  for (int j = 0; j < TOWER_K4_M; ++j)
    t |= (u32)((__builtin_popcountll(kTowerK4TwMask[j][0] & lo) + __builtin_popcountll(kTowerK4TwMask[j][1] & hi)) & 1) << j;
  return t;
}
template <>
inline u32 tower_twiddle_bits<5>(u64 lo, u64 hi) {
  u32 t = 0;
  for (int j = 0; j < TOWER_K5_M; ++j)
    t |= (u32)((__builtin_popcountll(kTowerK5TwMask[j][0] & lo) + __builtin_popcountll(kTowerK5TwMask[j][1] & hi)) & 1) << j;
  return t;
}
