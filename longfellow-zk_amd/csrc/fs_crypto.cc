// fs_crypto.cc -- see fs_crypto.h.  Host-only translation unit (no device code).
#include "fs_crypto.h"

#if defined(__x86_64__)
#include <cpuid.h>
#include <immintrin.h>
#define FS_X86 1
#else
#define FS_X86 0
#endif

namespace {
const uint32_t kK256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

int g_force_portable = 0;
struct HwCaps {
  bool sha = false, aes = false, clmul = false;
  HwCaps() {
#if FS_X86
    unsigned a, b, c, d;
    if (__get_cpuid(1, &a, &b, &c, &d)) {
      const bool ssse3 = c & (1u << 9), sse41 = c & (1u << 19);
      aes = (c & (1u << 25)) && sse41;
      clmul = (c & (1u << 1)) && sse41;
      if (__get_cpuid_count(7, 0, &a, &b, &c, &d)) sha = (b & (1u << 29)) && ssse3 && sse41;
    }
#endif
  }
};
const HwCaps& caps() {
  static const HwCaps c;
  return c;
}

inline uint32_t rr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void sha256_blocks_portable(uint32_t h[8], const uint8_t* p, size_t nblocks) {
  for (; nblocks; --nblocks, p += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rr(w[i - 15], 7) ^ rr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      const uint32_t s1 = rr(w[i - 2], 17) ^ rr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t t1 = hh + (rr(e, 6) ^ rr(e, 11) ^ rr(e, 25)) + ((e & f) ^ (~e & g)) + kK256[i] + w[i];
      const uint32_t t2 = (rr(a, 2) ^ rr(a, 13) ^ rr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
}

#if FS_X86
// SHA extensions: two rounds per sha256rnds2 on the (ABEF, CDGH) state split; message schedule by sha256msg1/2.
__attribute__((target("sha,sse4.1,ssse3"))) void sha256_blocks_ni(uint32_t h[8], const uint8_t* p, size_t nblocks) {
  const __m128i bswap = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
  __m128i t = _mm_loadu_si128((const __m128i*)&h[0]);   // DCBA
  __m128i s1 = _mm_loadu_si128((const __m128i*)&h[4]);  // HGFE
  t = _mm_shuffle_epi32(t, 0xB1);                       // CDAB
  s1 = _mm_shuffle_epi32(s1, 0x1B);                     // EFGH
  __m128i s0 = _mm_alignr_epi8(t, s1, 8);               // ABEF
  s1 = _mm_blend_epi16(s1, t, 0xF0);                    // CDGH
  for (; nblocks; --nblocks, p += 64) {
    const __m128i save0 = s0, save1 = s1;
    __m128i w[16];
    for (int i = 0; i < 16; ++i) {
      if (i < 4) {
        w[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(p + 16 * i)), bswap);
      } else {
        __m128i x = _mm_sha256msg1_epu32(w[i - 4], w[i - 3]);
        x = _mm_add_epi32(x, _mm_alignr_epi8(w[i - 1], w[i - 2], 4));
        w[i] = _mm_sha256msg2_epu32(x, w[i - 1]);
      }
      __m128i m = _mm_add_epi32(w[i], _mm_loadu_si128((const __m128i*)&kK256[4 * i]));
      s1 = _mm_sha256rnds2_epu32(s1, s0, m);
      m = _mm_shuffle_epi32(m, 0x0E);
      s0 = _mm_sha256rnds2_epu32(s0, s1, m);
    }
    s0 = _mm_add_epi32(s0, save0);
    s1 = _mm_add_epi32(s1, save1);
  }
  t = _mm_shuffle_epi32(s0, 0x1B);        // FEBA
  s1 = _mm_shuffle_epi32(s1, 0xB1);       // DCHG
  s0 = _mm_blend_epi16(t, s1, 0xF0);      // DCBA
  s1 = _mm_alignr_epi8(s1, t, 8);         // HGFE
  _mm_storeu_si128((__m128i*)&h[0], s0);
  _mm_storeu_si128((__m128i*)&h[4], s1);
}

__attribute__((target("aes,sse4.1"))) void aes256_encrypt_ni(const uint8_t rk[15][16], const uint8_t* in, uint8_t* out, size_t nblocks) {
  __m128i k[15];
  for (int i = 0; i < 15; ++i) k[i] = _mm_load_si128((const __m128i*)rk[i]);
  for (size_t b = 0; b < nblocks; ++b) {
    __m128i s = _mm_xor_si128(_mm_loadu_si128((const __m128i*)(in + 16 * b)), k[0]);
    for (int r = 1; r < 14; ++r) s = _mm_aesenc_si128(s, k[r]);
    s = _mm_aesenclast_si128(s, k[14]);
    _mm_storeu_si128((__m128i*)(out + 16 * b), s);
  }
}
#endif

void sha256_blocks(uint32_t h[8], const uint8_t* p, size_t nblocks) {
#if FS_X86
  if (caps().sha && !g_force_portable) return sha256_blocks_ni(h, p, nblocks);
#endif
  sha256_blocks_portable(h, p, nblocks);
}

struct AesTables {
  uint8_t sbox[256];
  AesTables() {  // multiplicative inverse in GF(2^8) followed by the affine map
    uint8_t ex[256], lg[256];
    uint32_t x = 1;
    for (int i = 0; i < 255; ++i) {
      ex[i] = (uint8_t)x;
      lg[x] = (uint8_t)i;
      x ^= (x << 1) ^ ((x & 0x80) ? 0x11B : 0);  // multiply by 3 (a generator)
      x &= 0xFF;
    }
    lg[0] = 0;
    for (int a = 0; a < 256; ++a) {
      const uint8_t inv = a ? ex[(255 - lg[a]) % 255] : 0;
      uint8_t s = inv;
      for (int sh = 1; sh <= 4; ++sh) s ^= (uint8_t)((inv << sh) | (inv >> (8 - sh)));
      sbox[a] = s ^ 0x63;
    }
  }
};
const AesTables& aes_tables() {
  static const AesTables t;
  return t;
}
inline uint8_t xt(uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1B : 0)); }

void aes256_encrypt_portable(const uint8_t rk[15][16], const uint8_t in[16], uint8_t out[16]) {
  const uint8_t* sb = aes_tables().sbox;
  uint8_t s[16], t[16];
  for (int i = 0; i < 16; ++i) s[i] = in[i] ^ rk[0][i];
  for (int rnd = 1; rnd <= 14; ++rnd) {
    for (int c = 0; c < 4; ++c)  // SubBytes + ShiftRows (column-major state: s[4c + r])
      for (int r = 0; r < 4; ++r) t[4 * c + r] = sb[s[4 * ((c + r) & 3) + r]];
    if (rnd != 14) {
      for (int c = 0; c < 4; ++c) {
        const uint8_t* a = &t[4 * c];
        const uint8_t x = a[0] ^ a[1] ^ a[2] ^ a[3];
        s[4 * c + 0] = a[0] ^ x ^ xt(a[0] ^ a[1]);
        s[4 * c + 1] = a[1] ^ x ^ xt(a[1] ^ a[2]);
        s[4 * c + 2] = a[2] ^ x ^ xt(a[2] ^ a[3]);
        s[4 * c + 3] = a[3] ^ x ^ xt(a[3] ^ a[0]);
      }
    } else {
      memcpy(s, t, 16);
    }
    for (int i = 0; i < 16; ++i) s[i] ^= rk[rnd][i];
  }
  memcpy(out, s, 16);
}
}  // namespace

#if FS_X86
__attribute__((target("pclmul,sse4.1"))) static void gf128_mul_clmul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
  const __m128i x = _mm_loadu_si128((const __m128i*)a), y = _mm_loadu_si128((const __m128i*)b);
  const __m128i t0 = _mm_clmulepi64_si128(x, y, 0x00), t3 = _mm_clmulepi64_si128(x, y, 0x11);
  const __m128i mid = _mm_xor_si128(_mm_clmulepi64_si128(x, y, 0x10), _mm_clmulepi64_si128(x, y, 0x01));
  uint64_t p0 = (uint64_t)_mm_extract_epi64(t0, 0), p1 = (uint64_t)_mm_extract_epi64(t0, 1) ^ (uint64_t)_mm_extract_epi64(mid, 0);
  uint64_t p2 = (uint64_t)_mm_extract_epi64(t3, 0) ^ (uint64_t)_mm_extract_epi64(mid, 1), p3 = (uint64_t)_mm_extract_epi64(t3, 1);
  // x^128 = x^7 + x^2 + x + 1: fold the two high words down, top word first
  const __m128i r = _mm_set_epi64x(0, 0x87);
  __m128i f = _mm_clmulepi64_si128(_mm_set_epi64x(0, (long long)p3), r, 0x00);
  p1 ^= (uint64_t)_mm_extract_epi64(f, 0);
  p2 ^= (uint64_t)_mm_extract_epi64(f, 1);
  f = _mm_clmulepi64_si128(_mm_set_epi64x(0, (long long)p2), r, 0x00);
  p0 ^= (uint64_t)_mm_extract_epi64(f, 0);
  p1 ^= (uint64_t)_mm_extract_epi64(f, 1);
  out[0] = p0;
  out[1] = p1;
}
#endif
bool fs_gf128_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
#if FS_X86
  if (caps().clmul && !g_force_portable) {
    gf128_mul_clmul(a, b, out);
    return true;
  }
#endif
  (void)a; (void)b; (void)out;
  return false;
}

int fs_crypto_hw() { return (caps().sha && caps().aes && !g_force_portable) ? 1 : 0; }
void fs_crypto_force_portable(int on) { g_force_portable = on; }

Sha256::Sha256() {
  static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  memcpy(h, iv, sizeof(h));
}
void Sha256::update(const uint8_t* p, size_t n) {
  len += n;
  if (fill) {
    const size_t take = n < 64 - fill ? n : 64 - fill;
    memcpy(buf + fill, p, take);
    fill += take; p += take; n -= take;
    if (fill == 64) { sha256_blocks(h, buf, 1); fill = 0; }
  }
  if (n >= 64) {
    sha256_blocks(h, p, n / 64);
    p += n & ~(size_t)63;
    n &= 63;
  }
  if (n) { memcpy(buf, p, n); fill = n; }
}
void Sha256::digest(uint8_t out[32]) const {
  Sha256 s = *this;
  const uint64_t bits = s.len * 8;
  uint8_t pad[72] = {0x80};
  const size_t padn = (s.fill < 56 ? 56 : 120) - s.fill;
  for (int i = 0; i < 8; ++i) pad[padn + i] = (uint8_t)(bits >> (56 - 8 * i));
  s.update(pad, padn + 8);
  for (int i = 0; i < 8; ++i) {
    out[4 * i] = (uint8_t)(s.h[i] >> 24); out[4 * i + 1] = (uint8_t)(s.h[i] >> 16);
    out[4 * i + 2] = (uint8_t)(s.h[i] >> 8); out[4 * i + 3] = (uint8_t)s.h[i];
  }
}

void Aes256::set_key(const uint8_t key[32]) {
  const uint8_t* sb = aes_tables().sbox;
  uint8_t w[60][4];
  memcpy(w, key, 32);
  uint8_t rcon = 1;
  for (int i = 8; i < 60; ++i) {
    uint8_t t[4] = {w[i - 1][0], w[i - 1][1], w[i - 1][2], w[i - 1][3]};
    if (i % 8 == 0) {
      const uint8_t t0 = t[0];
      t[0] = sb[t[1]] ^ rcon; t[1] = sb[t[2]]; t[2] = sb[t[3]]; t[3] = sb[t0];
      rcon = xt(rcon);
    } else if (i % 8 == 4) {
      for (int j = 0; j < 4; ++j) t[j] = sb[t[j]];
    }
    for (int j = 0; j < 4; ++j) w[i][j] = w[i - 8][j] ^ t[j];
  }
  memcpy(rk, w, sizeof(rk));
}
void Aes256::encrypt(const uint8_t in[16], uint8_t out[16]) const {
#if FS_X86
  if (caps().aes && !g_force_portable) return aes256_encrypt_ni(rk, in, out, 1);
#endif
  aes256_encrypt_portable(rk, in, out);
}
void Aes256::ctr_blocks(uint64_t ctr0, size_t nblocks, uint8_t* out) const {
  while (nblocks) {
    uint8_t ctr[8][16];
    const size_t nb = nblocks < 8 ? nblocks : 8;
    memset(ctr, 0, sizeof(ctr));
    for (size_t b = 0; b < nb; ++b)
      for (int i = 0; i < 8; ++i) ctr[b][i] = (uint8_t)((ctr0 + b) >> (8 * i));
#if FS_X86
    if (caps().aes && !g_force_portable) {
      aes256_encrypt_ni(rk, &ctr[0][0], out, nb);
    } else
#endif
      for (size_t b = 0; b < nb; ++b) aes256_encrypt_portable(rk, ctr[b], out + 16 * b);
    ctr0 += nb; out += 16 * nb; nblocks -= nb;
  }
}
