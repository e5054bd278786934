// fs_crypto.h -- host-side SHA-256 (FIPS 180-4) and AES-256 encryption (FIPS-197) for the built-in Fiat-Shamir
// transcript (reference lib/random/transcript.h:33-190 uses OpenSSL for both, lib/util/crypto.h:30-103).
// Portable C++ with SHA-NI / AES-NI fast paths selected once by cpuid (fs_crypto.cc).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

struct Sha256 {
  uint32_t h[8];
  uint64_t len = 0;
  uint8_t buf[64];
  size_t fill = 0;
  Sha256();
  void update(const uint8_t* p, size_t n);
  void digest(uint8_t out[32]) const;  // of a copy: the running state is not disturbed
};

struct Aes256 {
  alignas(16) uint8_t rk[15][16];
  void set_key(const uint8_t key[32]);
  void encrypt(const uint8_t in[16], uint8_t out[16]) const;
  // out[16*i..] = AES(LE64(ctr0 + i) || 0^8), i < nblocks  (FSPRF::refill, transcript.h:53-60)
  void ctr_blocks(uint64_t ctr0, size_t nblocks, uint8_t* out) const;
};

// GF(2^128) product mod x^128 + x^7 + x^2 + x + 1 (GF2_128::mulf, lib/gf2k/gf2_128.h:233-246) on the host with
// PCLMULQDQ; returns false (out untouched) when the instruction is missing or the portable paths are forced, so
// the caller falls back to the portable product of fields.h.  Operands: two little-endian u64 each.
bool fs_gf128_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]);

// 1 when the SHA-NI / AES-NI paths are in use (for the tests; 0 = portable code)
int fs_crypto_hw();
// force the portable paths (tests compare both)
void fs_crypto_force_portable(int on);
