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
  elt_t sub_tab[4][256];  // of_scalar by bytes: sub_tab[b][v] = sum of beta[8 b + i] over the set bits i of v (gf2_128.h:192-214)
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
  void* scratch4 = nullptr;  // caller-level tableaux that must survive RS / FFT / gather helpers (the verifier's A rows)
  size_t scratch4_bytes = 0;
  // cached device tables keyed by a string
  std::map<std::string, void*> tables;
  // cached host-side POD plans (e.g. RS op-list descriptors) keyed by a string
  std::map<std::string, std::string> blobs;
  GfHostCtx gf[2];  // [0] k = 4, [1] k = 5
  // small pinned host mailbox for results read back every call (roots, partial sums)
  void* mailbox_h = nullptr;
  void* mailbox_d = nullptr;
  // one-entry cache of the last freed Ligero tableau / Merkle buffers (a prover that commits repeatedly with the same
  // parameters does not pay hipMalloc + hipFree per proof)
  // (a small pool keyed by size: hipFree waits for EVERY stream of the device, so in the steady state of a prover -- or of K
  // provers sharing the device -- nothing may be freed per proof; lf_pool_get / lf_pool_put)
  struct PoolEntry {
    void* p;
    size_t bytes;
  };
  std::vector<PoolEntry> pool;
  // EQ table over the circuit inputs of the last verifier_constraints run (the dense block of the Ligero inner-product
  // matrix is built from it on the device)
  void* zk_eq = nullptr;
  size_t zk_eq_bytes = 0;
  // pinned staging ring for small host tables that are uploaded without a stream synchronisation: slot i may be
  // rewritten once stage_ev[i] (recorded after its copy) has completed
  void* stage_h = nullptr;
  hipEvent_t stage_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  unsigned stage_next = 0;
  // coherent (fine-grained) pinned words a running kernel writes and the host polls: results of the fused
  // sumcheck steps come back without a stream synchronisation
  volatile u64* poll_h = nullptr;
  u64 poll_seq = 0, poll_next = 0;
  int resident_state = 0;  // 0 untested, 1 the resident sumcheck kernel may be used, -1 it may not
  // per-context (= per-device) launch configuration: nothing below the C ABI is process-global, so one process may hold
  // contexts on several devices (hipFuncSetAttribute is applied once per context)
  unsigned attr_done = 0;  // bit 0 FFT tile kernels, 1 RS row kernel, 2 bit-sliced butterflies, 3 sumcheck grid tail
  int tile_log = 12;       // FFT tile = 2^tile_log elements (LFGPU_TILE_LOG = 12 | 13)
  unsigned bs_rlog = 5;    // bit-sliced butterfly tile: 2^bs_rlog combos (LFGPU_BS_RLOG = 5..7)
  int sc_tail_ok = -1;     // the shrinking-grid kernel may use its LDS tail (-1 undecided)
  int num_cu = 256;
  u64 wall_khz = 100000;  // rate of wall_clock64() (hipDeviceAttributeWallClockRate)
  hipStream_t own_stream = nullptr;  // lfgpu_own_stream: created by the library, destroyed with the context
  // resident (spin-barrier) kernels: CUs this context holds of its device's budget (lf_cu_acquire), and how often a grid
  // could not be placed although the budget said so (another process on the device): after two strikes the context stops
  // launching resident grids
  int cu_held = 0;
  int grid_strikes = 0;
  // lfgpu_set_rng_exact_calls: 1 = every RandomEngine draw is its own call of the caller's hook with the reference's size (16 /
  // 32 bytes per element, kSubFieldBytes per subfield element, 32 per Merkle nonce); 0 = consecutive draws may be merged into
  // one call, which is the same bytes for every engine that is a byte stream
  int rng_exact = 0;
};
// Per-device CU budget of the kernels whose workgroups wait for each other (sc_grid_layer_kernel, grid256_layer_kernel, the
// single-workgroup resident kernel).  Such a grid is only safe when ALL its workgroups are placed together; two grids each
// half placed would wait for CUs the other holds until their timeouts.  Every launch therefore first takes its workgroup
// count from a process-wide counter per device (budget = the device's CU count, LFGPU_CU_BUDGET overrides) and gives it back
// as the grid shrinks / ends: the sum over all live grids never exceeds the CUs, so every one of them gets placed whatever
// else (ordinary kernels, which always finish) shares the device.  All or nothing; false = take another driver this time.
bool lf_cu_acquire(lfgpu_ctx* c, int n);
int lf_cu_sharers(const lfgpu_ctx* c);  // contexts with a stream of their own on this device (lfgpu_own_stream): >= 2 = throughput mode
int lf_cu_available(const lfgpu_ctx* c);  // CUs of the device's budget nobody holds right now (a hint: it may change at once)
void lf_cu_release(lfgpu_ctx* c, int n);  // n < 0: everything the context holds

int lf_fail(lfgpu_ctx* c, int code, const char* fmt, ...);
// Zeroes a host container when its scope ends.  The witness, the pads, the blinding rows of a Ligero layout and a recorded
// RandomEngine stream are secrets of the prover; the allocator does not clear what it takes back.
template <class V>
struct LfScrubHost {
  V& v;
  explicit LfScrubHost(V& x) : v(x) {}
  ~LfScrubHost() {
    if (!v.empty()) explicit_bzero((void*)v.data(), v.size() * sizeof(v[0]));
  }
  LfScrubHost(const LfScrubHost&) = delete;
  LfScrubHost& operator=(const LfScrubHost&) = delete;
};
#define LF_SCRUB_ON_EXIT(vec) LfScrubHost<decltype(vec)> lf_scrub_##vec(vec)
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

// device buffers that outlive one call but recur with the same size (Ligero tableaux, Merkle layers): taken from / returned
// to the context's pool instead of hipMalloc / hipFree.  A returned buffer may still be in use by work queued on the
// context's stream (in order: whatever a later user enqueues comes after it).  At most LF_POOL_MAX entries and LF_POOL_MAX_BYTES
// in all, oldest evicted.
#define LF_POOL_MAX 8
#define LF_POOL_MAX_BYTES ((size_t)2 << 30)  // all pooled buffers of a context together; larger single buffers are freed at once
int lf_pool_get(lfgpu_ctx* c, size_t bytes, void** out);
void lf_pool_put(lfgpu_ctx* c, void* p, size_t bytes);
int lf_scratch(lfgpu_ctx* c, size_t bytes, void** out);
int lf_scratch2(lfgpu_ctx* c, size_t bytes, void** out);
int lf_scratch3(lfgpu_ctx* c, size_t bytes, void** out);
// Ownership: `scratch` = FFT pass buffers (fft.hip, lch_bs.hip), Merkle staging and the sumcheck layer state; `scratch2` = RS
// work rows, small per-call tables (indices, partial sums); `scratch3` = Ligero-level temporaries; `scratch4` = a caller's
// tableau that is handed to the RS / FFT / gather helpers and therefore must not alias any of the three above.
int lf_scratch4(lfgpu_ctx* c, size_t bytes, void** out);
// upload (and cache under `key`) a host table; returns device pointer
int lf_table(lfgpu_ctx* c, const std::string& key, const void* host, size_t bytes, void** out);
bool lf_table_lookup(lfgpu_ctx* c, const std::string& key, void** out);
// asynchronous upload of <= LF_STAGE_SLOT bytes through the pinned staging ring (no stream synchronisation)
#define LF_STAGE_SLOT 4096
int lf_stage_upload(lfgpu_ctx* c, void* d_dst, const void* h_src, size_t bytes);
// Quad::bind_gh_all without the read-back: enqueue on the stream into 4 device words, fold them on the host later
struct lfgpu_quad;
int lf_quad_bind_gh_all_enqueue(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2],
                                const uint64_t beta[2], size_t logw, size_t nw, const void* h_H0, const void* h_H1, u64* d_acc);
void lf_quad_bind_gh_all_fold(int field, const u64 w[4], uint64_t out[2]);
// Quad::bind_g, enqueue only (nothing is read back: the HQUAD size is computed once at lfgpu_quad_upload)
int lf_quad_bind_g(lfgpu_quad* q, size_t logv, const void* h_G0, const void* h_G1, const uint64_t alpha[2], const uint64_t beta[2],
                   void* d_hc_out, void* d_vc_out, size_t* n_out, void* d_zero2 = nullptr, size_t zero2_bytes = 0,
                   void* d_zero3 = nullptr, size_t zero3_bytes = 0);
#define LF_GH_BATCH_MAX 96  // layers whose sums fit the device mailbox (32 bytes each from offset 512)

// host-side field helpers (use the LF_HD arithmetic of fields.h compiled for the host)
elt_t h_gf_inv(elt_t a);
elt_t h_fp_inv(elt_t a);
elt_t h_fp_of_scalar(u64 u);
elt_t h_fp_to_mont(elt_t raw);  // raw < p -> Montgomery image
bool h_fp_fits(elt_t raw);      // raw < p
// Reed-Solomon row extension for either field (GF2_128<k>: LCH14; Fp128: convolution with the 2^32-order root)
int lf_rs_rows(lfgpu_ctx* c, int field, int k, size_t nrow, size_t n, size_t m, elt_t* d, size_t ld);
int lf_gf_rs_rows_mixed(lfgpu_ctx* c, int k, size_t nrow, size_t n1, size_t n2, size_t lo2, size_t hi2, size_t m, elt_t* d_T, size_t ld);
const GfHostCtx* lf_gf_ctx(lfgpu_ctx* c, int k);
bool lf_gf_ctx_build(GfHostCtx* g, int k);  // the same constants without a context (host-only entry points)
elt_t h_lch14_twiddle(const GfHostCtx* g, unsigned i, u64 u);

// fused single-workgroup sumcheck step (sumcheck.hip): [bind of the previous round-hand] -> [QW scatter + the two
// partial sums of the next one]; results land in c->poll_h = {a0.lo, a0.hi, a2.lo, a2.hi, nh, seq, scalar.lo, scalar.hi, status}; the resident variant reads
// its challenges from c->poll_h + 64 = {r.lo, r.hi, seq}
struct ScSmall {
  int field;
  int do_bind, bind_hand;
  elt_t r;
  int do_eval, eval_hand;
  uint2* hc_in;   // current HQUAD corners / values
  elt_t* vc_in;
  uint2* hc_out;  // destination of the bind (the other half of the ping-pong)
  elt_t* vc_out;
  u32 nh;         // HQUAD size before the bind (host-known)
  elt_t* W[2];    // hand arrays before the bind
  u32 nW[2];
  elt_t* Wdst;    // destination of the dense bind (== W[bind_hand] for in place)
  u64* QW;        // scratch: GF 2 words / Fp 4 limb accumulators per target
};
int lf_sc_small_step(lfgpu_ctx* c, const ScSmall& a, u64 out[8]);
int lf_sc_layer_begin(lfgpu_ctx* c, const ScSmall& a, u32 rh0, u32 rh1, void* d_W_shared, void* wtmp);
int lf_sc_layer_next(lfgpu_ctx* c, const u64* r, u64 out[8]);
bool lf_sc_resident_ok(lfgpu_ctx* c);
// Where each workgroup's range of HQuad::bind_h lands in the output, per round-hand of the shrinking grid, and the size after
// the bind: circuit constants (which entries merge depends on the corner indices only), recorded by the first proof that runs
// the layer through the grid and replayed by the later ones -- their round-hands then need neither the count nor the
// device-wide barrier in front of the offsets.  Lives with the layer's quad.
struct ScGridOffCache {
  u32* d = nullptr;                 // [64 round-hands][LF_SC_GRID_WGS + 1] words
  u32 key[7] = {0, 0, 0, 0, 0, 0, 0};  // {first round-hand, workgroups, HQUAD size, entries per workgroup, both hand sizes, logw} of the record
  int state = 0;                    // 0 none, 1 being recorded (valid once the layer ends well), 2 valid
};
int lf_sc_grid_begin(lfgpu_ctx* c, int field, void* hc_cur, void* vc_cur, void* hc_oth, void* vc_oth, size_t nh, const u32* d_nh, void* W0, size_t nW0,
                     void* W1, size_t nW1, void* Wb00, void* Wb01, void* Wb10, void* Wb11, void* qw, size_t rh0,
                     size_t logw, void* d_state, ScGridOffCache* oc = nullptr, u32* G_out = nullptr, u32* per_wg_out = nullptr,
                     bool state_clean = false);
#define LF_SC_GRID_SYNC_CLEAR_BYTES (64 + 1024)  // head of the grid state that must be zero at launch: counters (both levels), abort flag, challenge slot
// ^ LFGPU_ERR_BUSY (nothing launched, no message): the device's CU budget is short of the grid -- use another driver now
#define LF_SC_GRID_WGS 128                         // most workgroups the shrinking-grid kernel starts with
#define LF_SC_GRID_MAX (256 * 1024)             // largest HQUAD / hand array it takes
#define LF_SC_GRID_STATE_BYTES (64 + 1024 + 32 * LF_SC_GRID_WGS + 4 * LF_SC_GRID_WGS + 64 + 36 * LF_SC_GRID_MAX)
#define LF_SC_SMALL_MAX 8192  // largest HQUAD / hand array the single-workgroup step takes

int lf_hquad_bind_h_cached(lfgpu_ctx* c, int field, size_t n, const void* d_hc, const void* d_vc, const uint64_t r[2], int hand,
                           void* d_hc_out, void* d_vc_out, const u32* d_off_cached, u32** d_off_keep, size_t* n_out);  // sumcheck.hip
struct lfgpu_quad;
int lf_eval_quad_async(lfgpu_quad* q, const void* d_W, void* d_V, int* d_fail);  // quad.hip

// p256.hip (field id 1, 32-byte elements)
int lf_column_leaves32(lfgpu_ctx* c, size_t nrow, size_t ld, size_t col0, size_t ncols, const void* d_T, const void* d_nonces, void* d_out, size_t out0);
int lf_p256_binop(lfgpu_ctx* c, int op, size_t n, const void* d_a, const void* d_b, void* d_out);

static inline unsigned lf_log2(size_t n) {
  unsigned l = 0;
  while (((size_t)1 << l) < n) ++l;
  return l;
}

// ligero.hip: helpers of the sharded (multi-GPU) paths, shared with the ZK driver
int lf_comm_bcast_blob(const lfgpu_comm_ops* cm, std::vector<uint8_t>& blob);  // rank 0's bytes to every rank (host buffers)
void lf_replay_rng(const std::vector<uint8_t>* data, lfgpu_rng_fn* fn, void** user, void* storage /* 32 bytes */);
void lf_record_rng(lfgpu_rng_fn rng, void* rng_user, std::vector<uint8_t>* rec, lfgpu_rng_fn* fn, void** user, void* storage /* 32 bytes */);
