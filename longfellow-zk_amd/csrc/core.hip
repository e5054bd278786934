// core.hip -- context, memory helpers, host-side field constants.
#include <algorithm>

#include "ctx.h"

int lf_fail(lfgpu_ctx* c, int code, const char* fmt, ...) {
  if (c) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

static int grow(lfgpu_ctx* c, void** buf, size_t* cap, size_t bytes, void** out) {
  if (bytes > *cap) {
    if (*buf) {
      LF_HIP(c, hipStreamSynchronize(c->stream));
      LF_HIP(c, hipFree(*buf));
      *buf = nullptr;
      *cap = 0;
    }
    hipError_t e = hipMalloc(buf, bytes);
    if (e != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    *cap = bytes;
  }
  *out = *buf;
  return LFGPU_OK;
}
int lf_scratch(lfgpu_ctx* c, size_t bytes, void** out) { return grow(c, &c->scratch, &c->scratch_bytes, bytes, out); }
int lf_scratch2(lfgpu_ctx* c, size_t bytes, void** out) { return grow(c, &c->scratch2, &c->scratch2_bytes, bytes, out); }
int lf_scratch3(lfgpu_ctx* c, size_t bytes, void** out) { return grow(c, &c->scratch3, &c->scratch3_bytes, bytes, out); }
int lf_scratch4(lfgpu_ctx* c, size_t bytes, void** out) { return grow(c, &c->scratch4, &c->scratch4_bytes, bytes, out); }

int lf_pool_get(lfgpu_ctx* c, size_t bytes, void** out) {
  for (size_t i = 0; i < c->pool.size(); ++i)
    if (c->pool[i].bytes == bytes) {
      *out = c->pool[i].p;
      c->pool.erase(c->pool.begin() + i);
      return LFGPU_OK;
    }
  if (hipMalloc(out, bytes ? bytes : 16) != hipSuccess) {
    (void)hipGetLastError();
    return lf_fail(c, LFGPU_ERR_NOMEM, "hipMalloc(%zu) failed", bytes);
  }
  return LFGPU_OK;
}
void lf_pool_put(lfgpu_ctx* c, void* p, size_t bytes) {
  if (!p) return;
  if (bytes > LF_POOL_MAX_BYTES) {  // a one-off giant (a 16 GiB synthetic tableau): not worth holding on to
    (void)hipFree(p);
    return;
  }
  size_t held = bytes;
  for (const auto& e : c->pool) held += e.bytes;
  while (!c->pool.empty() && (c->pool.size() >= LF_POOL_MAX || held > LF_POOL_MAX_BYTES)) {  // oldest first
    held -= c->pool.front().bytes;
    (void)hipFree(c->pool.front().p);
    c->pool.erase(c->pool.begin());
  }
  c->pool.push_back(lfgpu_ctx::PoolEntry{p, bytes});
}

int lf_stage_upload(lfgpu_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
  const unsigned slot = c->stage_next & 3;
  if (!c->stage_h || !c->stage_ev[slot] || bytes > LF_STAGE_SLOT) {  // no ring: plain copy, the source must outlive it
    LF_HIP(c, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    LF_HIP(c, hipStreamSynchronize(c->stream));
    return LFGPU_OK;
  }
  ++c->stage_next;
  LF_HIP(c, hipEventSynchronize(c->stage_ev[slot]));  // the copy that last used this slot (normally long done)
  void* h = (uint8_t*)c->stage_h + (size_t)slot * LF_STAGE_SLOT;
  memcpy(h, h_src, bytes);
  LF_HIP(c, hipMemcpyAsync(d_dst, h, bytes, hipMemcpyHostToDevice, c->stream));
  LF_HIP(c, hipEventRecord(c->stage_ev[slot], c->stream));
  return LFGPU_OK;
}

bool lf_table_lookup(lfgpu_ctx* c, const std::string& key, void** out) {
  auto it = c->tables.find(key);
  if (it == c->tables.end()) return false;
  *out = it->second;
  return true;
}
int lf_table(lfgpu_ctx* c, const std::string& key, const void* host, size_t bytes, void** out) {
  if (lf_table_lookup(c, key, out)) return LFGPU_OK;
  void* d = nullptr;
  hipError_t e = hipMalloc(&d, bytes ? bytes : 16);
  if (e != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "hipMalloc(table %zu) failed: %s", bytes, hipGetErrorString(e));
  // synchronous copy: `host` is usually a temporary
  LF_HIP(c, hipMemcpy(d, host, bytes, hipMemcpyHostToDevice));
  c->tables[key] = d;
  *out = d;
  return LFGPU_OK;
}

// ------------------------------------------------------------------ host field helpers
elt_t h_gf_inv(elt_t a) {  // a^(2^128-2)
  elt_t r{1, 0}, s = a;
  for (int i = 1; i < 128; ++i) {
    s = gf_mul(s, s);
    r = gf_mul(r, s);
  }
  return r;
}
static elt_t h_fp_rsq() {
  static const elt_t rsq = [] {  // thread-safe one-time initialisation (C++11)
    elt_t r{1, 0};
    for (int i = 0; i < 256; ++i) r = fp_add(r, r);
    return r;
  }();
  return rsq;
}
elt_t h_fp_of_scalar(u64 u) { return fp_mul(elt_t{u, 0}, h_fp_rsq()); }
elt_t h_fp_to_mont(elt_t raw) { return fp_mul(raw, h_fp_rsq()); }
bool h_fp_fits(elt_t raw) { return raw.hi < FP_P_HI || (raw.hi == FP_P_HI && raw.lo < FP_P_LO); }
elt_t h_fp_inv(elt_t x) {  // x^(p-2)
  unsigned __int128 e = (((unsigned __int128)FP_P_HI) << 64 | FP_P_LO) - 2;
  elt_t r = h_fp_of_scalar(1), b = x;
  while (e) {
    if (e & 1) r = fp_mul(r, b);
    b = fp_mul(b, b);
    e >>= 1;
  }
  return r;
}

// GF2_128<k> ctor + LCH14 ctor constants (lib/gf2k/gf2_128.h:97-116,369-391; lch14.h:45-77)
bool lf_gf_ctx_build(GfHostCtx* g, int k) {
  if (k != 4 && k != 5) return false;
  if (g->init) return true;
  g->k = k;
  g->sub_bits = 1u << k;
  elt_t r{2, 0};
  for (unsigned i = k; i < 7; ++i) {
    elt_t s = r;
    for (unsigned j = 0; j < (1u << i); ++j) s = gf_mul(s, s);
    r = gf_mul(r, s);
  }
  g->g = r;
  g->beta[0] = elt_t{1, 0};
  for (unsigned i = 1; i < g->sub_bits; ++i) g->beta[i] = gf_mul(g->beta[i - 1], r);
  unsigned sb = g->sub_bits;
  for (unsigned j = 0; j < sb; ++j) g->w_hat[0][j] = g->beta[j];
  for (unsigned i = 0; i + 1 < sb; ++i)
    for (unsigned j = 0; j < sb; ++j)
      g->w_hat[i + 1][j] = gf_mul(g->w_hat[i][j], gf_add(g->w_hat[i][j], g->w_hat[i][i]));
  for (unsigned i = 0; i < sb; ++i) {
    elt_t sc = h_gf_inv(g->w_hat[i][i]);
    for (unsigned j = 0; j < sb; ++j) g->w_hat[i][j] = gf_mul(sc, g->w_hat[i][j]);
  }
  for (unsigned b = 0; b < 4; ++b)
    for (unsigned v = 0; v < 256; ++v) {
      elt_t t{0, 0};
      for (unsigned i = 0; i < 8; ++i)
        if (((v >> i) & 1) && 8 * b + i < sb) t = gf_add(t, g->beta[8 * b + i]);
      g->sub_tab[b][v] = t;
    }
  g->init = true;
  return true;
}
const GfHostCtx* lf_gf_ctx(lfgpu_ctx* c, int k) {
  if (k != 4 && k != 5) return nullptr;
  GfHostCtx* g = &c->gf[k - 4];
  return lf_gf_ctx_build(g, k) ? g : nullptr;
}
// LCH14::twiddle (lch14.h:81-89)
elt_t h_lch14_twiddle(const GfHostCtx* g, unsigned i, u64 u) {
  elt_t t{0, 0};
  for (unsigned k = 0; u != 0 && k < g->sub_bits; ++k, u >>= 1)
    if (u & 1) t = gf_add(t, g->w_hat[i][k]);
  return t;
}

// ------------------------------------------------------------------ CU budget of the resident kernels (ctx.h)
#include <atomic>
#include <mutex>
namespace {
struct CuBudget {
  std::mutex m;
  int used[64] = {0};
  int sharers[64] = {0};  // contexts with their own stream per device
} g_cu;
int cu_limit(const lfgpu_ctx* c) {
  static const int env = getenv("LFGPU_CU_BUDGET") ? atoi(getenv("LFGPU_CU_BUDGET")) : -1;
  return env >= 0 ? env : c->num_cu;
}
}  // namespace
bool lf_cu_acquire(lfgpu_ctx* c, int n) {
  if (n <= 0 || c->device < 0 || c->device >= 64) return false;
  std::lock_guard<std::mutex> lk(g_cu.m);
  if (g_cu.used[c->device] + n > cu_limit(c)) return false;
  g_cu.used[c->device] += n;
  c->cu_held += n;
  return true;
}
int lf_cu_sharers(const lfgpu_ctx* c) {
  if (c->device < 0 || c->device >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_cu.m);
  return g_cu.sharers[c->device];
}
static void cu_sharer_add(const lfgpu_ctx* c, int d) {
  if (c->device < 0 || c->device >= 64) return;
  std::lock_guard<std::mutex> lk(g_cu.m);
  g_cu.sharers[c->device] += d;
}
int lf_cu_available(const lfgpu_ctx* c) {
  if (c->device < 0 || c->device >= 64) return 0;
  std::lock_guard<std::mutex> lk(g_cu.m);
  return std::max(0, cu_limit(c) - g_cu.used[c->device]);
}
void lf_cu_release(lfgpu_ctx* c, int n) {
  if (n < 0 || n > c->cu_held) n = c->cu_held;
  if (n == 0) return;
  std::lock_guard<std::mutex> lk(g_cu.m);
  g_cu.used[c->device] -= n;
  c->cu_held -= n;
}

// first launch of any kernel of this library loads its code object onto the device (~150 ms for the 2 MiB of gfx950 code):
// lfgpu_init pays that, so that the first FFT / circuit upload / proof of a process is not the one that does
__global__ void lf_warm_kernel(u32* p) {
  if (threadIdx.x == 0 && p) p[0] = 0;
}

// ------------------------------------------------------------------ C ABI: context
extern "C" {

int lfgpu_init(int device, lfgpu_ctx** out) {
  if (!out) return LFGPU_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return LFGPU_ERR_HIP;
  if (device < 0 || device >= ndev) return LFGPU_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return LFGPU_ERR_HIP;
  lfgpu_ctx* c = new lfgpu_ctx();
  c->device = device;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_khz = (u64)khz;
  if (hipHostMalloc(&c->mailbox_h, 4096) != hipSuccess || hipMalloc(&c->mailbox_d, 4096) != hipSuccess) {
    delete c;
    return LFGPU_ERR_NOMEM;
  }
  void* ph = nullptr;
  if (hipHostMalloc(&ph, 4096, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    hipHostFree(c->mailbox_h);
    hipFree(c->mailbox_d);
    delete c;
    return LFGPU_ERR_NOMEM;
  }
  memset(ph, 0, 4096);
  c->poll_h = (volatile u64*)ph;
  if (hipHostMalloc(&c->stage_h, 4 * LF_STAGE_SLOT) != hipSuccess) c->stage_h = nullptr;  // optional: uploads then synchronise
  for (int i = 0; i < 4 && c->stage_h; ++i)
    if (hipEventCreateWithFlags(&c->stage_ev[i], hipEventDisableTiming) != hipSuccess) c->stage_ev[i] = nullptr;
  hipLaunchKernelGGL(lf_warm_kernel, dim3(1), dim3(64), 0, c->stream, (u32*)c->mailbox_d);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
    lfgpu_shutdown(c);
    return LFGPU_ERR_HIP;
  }
  *out = c;
  return LFGPU_OK;
}

int lfgpu_shutdown(lfgpu_ctx* c) {
  if (!c) return LFGPU_ERR_ARG;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  lf_cu_release(c, -1);
  if (c->own_stream) {
    cu_sharer_add(c, -1);
    if (c->stream == c->own_stream) c->stream = nullptr;
    hipStreamDestroy(c->own_stream);
  }
  for (auto& kv : c->tables) hipFree(kv.second);
  if (c->scratch) hipFree(c->scratch);
  if (c->scratch2) hipFree(c->scratch2);
  if (c->scratch3) hipFree(c->scratch3);
  if (c->scratch4) hipFree(c->scratch4);
  for (auto& e : c->pool) hipFree(e.p);
  if (c->zk_eq) hipFree(c->zk_eq);
  if (c->mailbox_h) hipHostFree(c->mailbox_h);
  if (c->poll_h) hipHostFree((void*)c->poll_h);
  for (int i = 0; i < 4; ++i)
    if (c->stage_ev[i]) hipEventDestroy(c->stage_ev[i]);
  if (c->stage_h) hipHostFree(c->stage_h);
  if (c->mailbox_d) hipFree(c->mailbox_d);
  delete c;
  return LFGPU_OK;
}

const char* lfgpu_last_error(const lfgpu_ctx* c) { return c ? c->err : "null context"; }

int lfgpu_set_stream(lfgpu_ctx* c, void* s) {
  if (!c) return LFGPU_ERR_ARG;
  c->stream = (hipStream_t)s;
  return LFGPU_OK;
}
int lfgpu_own_stream(lfgpu_ctx* c) {
  if (!c) return LFGPU_ERR_ARG;
  LF_HIP(c, hipSetDevice(c->device));
  if (!c->own_stream) {
    LF_HIP(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    cu_sharer_add(c, +1);
    // HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment said otherwise BEFORE the first HIP
    // call of the process).  Streams that share a queue run in order: one prover's kernels then wait behind another prover's
    // resident grid, for a millisecond each time (measured: 16 provers fall from 385 to below 200 proofs/s).  Said once.
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const int hwq = e && atoi(e) > 0 ? atoi(e) : 4;
    static std::atomic<bool> warned{false};
    if (lf_cu_sharers(c) > hwq && !getenv("LFGPU_QUIET") && !warned.exchange(true))
      fprintf(stderr, "lfgpu: %d contexts with streams of their own on device %d but GPU_MAX_HW_QUEUES = %d: set GPU_MAX_HW_QUEUES >= the number of "
                      "concurrent provers before the first HIP call (INTEGRATION.md, throughput mode)\n", lf_cu_sharers(c), c->device, hwq);
  }
  LF_HIP(c, hipStreamSynchronize(c->stream));  // what was enqueued on the old stream is complete before the switch
  c->stream = c->own_stream;
  return LFGPU_OK;
}
int lfgpu_set_rng_exact_calls(lfgpu_ctx* c, int exact) {
  if (!c) return LFGPU_ERR_ARG;
  c->rng_exact = exact ? 1 : 0;
  return LFGPU_OK;
}
int lfgpu_sync(lfgpu_ctx* c) {
  if (!c) return LFGPU_ERR_ARG;
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}
int lfgpu_malloc(lfgpu_ctx* c, size_t bytes, void** d_out) {
  if (!c || !d_out) return LFGPU_ERR_ARG;
  hipError_t e = hipMalloc(d_out, bytes ? bytes : 16);
  if (e != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return LFGPU_OK;
}
int lfgpu_free(lfgpu_ctx* c, void* d) {
  if (!c) return LFGPU_ERR_ARG;
  LF_HIP(c, hipFree(d));
  return LFGPU_OK;
}
int lfgpu_memcpy_h2d(lfgpu_ctx* c, void* d, const void* h, size_t bytes) {
  if (!c) return LFGPU_ERR_ARG;
  LF_HIP(c, hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}
int lfgpu_memcpy_d2h(lfgpu_ctx* c, void* h, const void* d, size_t bytes) {
  if (!c) return LFGPU_ERR_ARG;
  LF_HIP(c, hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}

}  // extern "C"
