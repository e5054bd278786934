// ctx.h -- context shared by the C-ABI translation units (host side).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/lfgpu.h"
#include "fields.h"

struct GfHostCtx {  // GF2_128<k> constants (lib/gf2k/gf2_128.h:97-116, lch14.h:45-77)
  bool init = false;
  unsigned k = 0, sub_bits = 0;
  elt_t g{};
  elt_t beta[32];
  elt_t w_hat[32][32];
};

struct lfgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  // scratch (grown on demand, never shrunk)
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  void* scratch2 = nullptr;
  size_t scratch2_bytes = 0;
  void* scratch3 = nullptr;  // Ligero-level temporaries (never used by the kernels' own launchers)
  size_t scratch3_bytes = 0;
  // cached device tables keyed by a string
  std::map<std::string, void*> tables;
  // cached host-side POD plans (e.g. RS op-list descriptors) keyed by a string
  std::map<std::string, std::string> blobs;
  GfHostCtx gf[2];  // [0] k = 4, [1] k = 5
  // small pinned host mailbox for results read back every call (roots, partial sums)
  void* mailbox_h = nullptr;
  void* mailbox_d = nullptr;
  int num_cu = 256;
};

int lf_fail(lfgpu_ctx* c, int code, const char* fmt, ...);
#define LF_HIP(c, call)                                                                 \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess)                                                               \
      return lf_fail((c), LFGPU_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                               \
  } while (0)
#define LF_TRY(expr)               \
  do {                             \
    int rc_ = (expr);              \
    if (rc_ != LFGPU_OK) return rc_; \
  } while (0)

int lf_scratch(lfgpu_ctx* c, size_t bytes, void** out);
int lf_scratch2(lfgpu_ctx* c, size_t bytes, void** out);
int lf_scratch3(lfgpu_ctx* c, size_t bytes, void** out);
// upload (and cache under `key`) a host table; returns device pointer
int lf_table(lfgpu_ctx* c, const std::string& key, const void* host, size_t bytes, void** out);
bool lf_table_lookup(lfgpu_ctx* c, const std::string& key, void** out);

// host-side field helpers (use the LF_HD arithmetic of fields.h compiled for the host)
elt_t h_gf_inv(elt_t a);
elt_t h_fp_inv(elt_t a);
elt_t h_fp_of_scalar(u64 u);
const GfHostCtx* lf_gf_ctx(lfgpu_ctx* c, int k);
elt_t h_lch14_twiddle(const GfHostCtx* g, unsigned i, u64 u);

static inline unsigned lf_log2(size_t n) {
  unsigned l = 0;
  while (((size_t)1 << l) < n) ++l;
  return l;
}
