// host_fields.hip -- compiles longfellow-zk_amd/csrc/fields.h (the device arithmetic)
// for the HOST, so the limb-level logic (Montgomery carries, Karatsuba clmul, SHA-256
// rounds) is unit-tested against the oracle in the CPU-only container.
#include "../longfellow-zk_amd/csrc/fields.h"

extern "C" {
void hf_fp_mul(const u64* a, const u64* b, u64* o) { elt_t r = fp_mul(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
void hf_fp_add(const u64* a, const u64* b, u64* o) { elt_t r = fp_add(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
void hf_fp_sub(const u64* a, const u64* b, u64* o) { elt_t r = fp_sub(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
void hf_gf_mul(const u64* a, const u64* b, u64* o) { elt_t r = gf_mul(elt_t{a[0], a[1]}, elt_t{b[0], b[1]}); o[0] = r.lo; o[1] = r.hi; }
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
