// zk.hip -- host driver of the ZK prover over the device kernels (include/lfgpu_zk.h).
//
// Host control flow restated from the reference (no reference code is linked or copied):
//   Transcript / FSPRF                 lib/random/transcript.h:33-190, lib/random/random.h:57-105
//   CircuitRep::from_bytes (LFC1)      lib/proto/circuit_reader.h:55-233, circuit_writer.h:103-114
//   ZkProver::commit / fill_pad        lib/zk/zk_prover.h:72-96,152-188
//   ZkProver::prove                    lib/zk/zk_prover.h:98-149
//   ProverLayers::prove (padded)       lib/sumcheck/prover_layers.h:106-183,320-344
//   ZkCommon::verifier_constraints     lib/zk/zk_common.h:49-136,406-439
//   LigeroProver::prove                lib/ligero/ligero_prover.h:84-146, inner_product_vector ligero_param.h:382-421
//   ZkProof::write                     lib/zk/zk_proof.h:90-185
// Every data-parallel step is a kernel behind lfgpu.h (eval_quad, sumcheck_layer, raw_eq2, ligero_*); what runs
// here is the sequential bookkeeping the reference also keeps on the host.  This file has no device code.
#include <algorithm>
#include <chrono>
#include <memory>

#include "../../include/lfgpu_zk.h"
#include "fp256.h"
#include "fs_crypto.h"
#include "hostfield.h"
#include "zkint.h"

extern "C" int lfgpu_raw_eq2(lfgpu_ctx*, int, size_t, size_t, const void*, const void*, const uint64_t*, void*);

namespace {
constexpr size_t kMaxBindings = 40;  // Proof::kMaxBindings (lib/sumcheck/circuit.h:84)

}  // namespace

// ------------------------------------------------------------------ built-in transcript
struct lfgpu_transcript {  // Transcript + FSPRF (lib/random/transcript.h:33-190)
  Sha256 sha;
  bool prf = false;
  Aes256 aes;
  u64 nblock = 0;
  uint8_t saved[16];
  size_t rdptr = 16;
  void upd(const uint8_t* p, size_t n) {
    prf = false;  // any write invalidates the PRF (:174-178)
    sha.update(p, n);
  }
  void tag_len(uint8_t tag, u64 n) {
    uint8_t h[9] = {tag};
    for (int i = 0; i < 8; ++i) h[1 + i] = (uint8_t)(n >> (8 * i));
    upd(h, 9);
  }
  void write_bytes(const uint8_t* d, size_t n) {  // tag 0 || u64 length || bytes (:115-120)
    tag_len(0, n);
    if (n) upd(d, n);
  }
  void write_elt(const uint8_t* e, size_t nbytes = 16) {  // tag 1 || image (:136-140)
    const uint8_t t = 1;
    upd(&t, 1);
    upd(e, nbytes);
  }
  void write_elt_array(const uint8_t* e, size_t n, size_t nbytes = 16) {  // tag 2 || u64 count || images (:144-152)
    tag_len(2, n);
    if (n) upd(e, nbytes * n);
  }
  void bytes(uint8_t* out, size_t n) {
    if (!prf) {  // key = SHA-256 of a copy of the running state (:160-172)
      uint8_t key[32];
      sha.digest(key);
      aes.set_key(key);
      nblock = 0;
      rdptr = 16;
      prf = true;
    }
    while (n) {
      if (rdptr == 16) {
        if (n >= 16) {  // whole blocks straight into the output (bulk RandomEngine draws)
          const size_t nb = n / 16;
          aes.ctr_blocks(nblock, nb, out);
          nblock += nb;
          out += 16 * nb;
          n -= 16 * nb;
          continue;
        }
        aes.ctr_blocks(nblock, 1, saved);  // FSPRF::refill (:53-60): AES(LE64 counter || 0^8)
        ++nblock;
        rdptr = 0;
      }
      const size_t take = n < 16 - rdptr ? n : 16 - rdptr;
      memcpy(out, saved + rdptr, take);
      out += take; rdptr += take; n -= take;
    }
  }
};

extern "C" {
lfgpu_transcript* lfgpu_transcript_new(const uint8_t* init, size_t n) {
  if (n && !init) return nullptr;
  lfgpu_transcript* t = new (std::nothrow) lfgpu_transcript();
  if (t) t->write_bytes(init, n);
  return t;
}
void lfgpu_transcript_free(lfgpu_transcript* t) { delete t; }
void lfgpu_transcript_write_bytes(lfgpu_transcript* t, const uint8_t* d, size_t n) { t->write_bytes(d, n); }
void lfgpu_transcript_write_elt(lfgpu_transcript* t, const uint8_t* e) { t->write_elt(e); }
void lfgpu_transcript_write_elt_array(lfgpu_transcript* t, const uint8_t* e, size_t n) { t->write_elt_array(e, n); }
void lfgpu_transcript_bytes(lfgpu_transcript* t, uint8_t* out, size_t n) { t->bytes(out, n); }
void lfgpu_transcript_write_elt_sized(lfgpu_transcript* t, const uint8_t* e, size_t nbytes) { t->write_elt(e, nbytes); }
void lfgpu_transcript_write_elt_array_sized(lfgpu_transcript* t, const uint8_t* e, size_t n, size_t nbytes) { t->write_elt_array(e, n, nbytes); }
void lfgpu_sha256(const uint8_t* data, size_t n, uint8_t out[32]) {
  Sha256 s;
  s.update(data, n);
  s.digest(out);
}
void lfgpu_host_gf2128_mul(const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
  const elt_t r = h_gf_mul(elt_t{a[0], a[1]}, elt_t{b[0], b[1]});
  out[0] = r.lo;
  out[1] = r.hi;
}
int lfgpu_crypto_hw(int force_portable) {
  if (force_portable >= 0) fs_crypto_force_portable(force_portable);
  return fs_crypto_hw();
}
void lfgpu_aes256_ecb_block(const uint8_t key[32], const uint8_t in[16], uint8_t out[16]) {
  Aes256 a;
  a.set_key(key);
  a.encrypt(in, out);
}
static void op_write_bytes(void* u, const uint8_t* d, size_t n) { ((lfgpu_transcript*)u)->write_bytes(d, n); }
static void op_write_elt(void* u, const uint8_t* e) { ((lfgpu_transcript*)u)->write_elt(e); }
static void op_write_arr(void* u, const uint8_t* e, size_t n) { ((lfgpu_transcript*)u)->write_elt_array(e, n); }
static void op_bytes(void* u, uint8_t* o, size_t n) { ((lfgpu_transcript*)u)->bytes(o, n); }
static void op_write_elt_sized(void* u, const uint8_t* e, size_t nb) { ((lfgpu_transcript*)u)->write_elt(e, nb); }
static void op_write_arr_sized(void* u, const uint8_t* e, size_t n, size_t nb) { ((lfgpu_transcript*)u)->write_elt_array(e, n, nb); }
static void* op_clone(void* u) {  // Transcript::clone copies the hash state only; the PRF restarts (:95-99)
  lfgpu_transcript* t = new (std::nothrow) lfgpu_transcript();
  if (t) t->sha = ((lfgpu_transcript*)u)->sha;
  return t;
}
static void op_free(void* u) { delete (lfgpu_transcript*)u; }
void lfgpu_transcript_get_ops(lfgpu_transcript* t, lfgpu_transcript_ops* ops) {
  ops->user = t;
  ops->write_bytes = op_write_bytes;
  ops->write_elt = op_write_elt;
  ops->write_elt_array = op_write_arr;
  ops->gen_bytes = op_bytes;
  ops->clone = op_clone;
  ops->free_clone = op_free;
  ops->write_elt_sized = op_write_elt_sized;
  ops->write_elt_array_sized = op_write_arr_sized;
}
}  // extern "C"

namespace {
// the caller's transcript seen through the hooks, plus the samplers built on RandomEngine::bytes
// to_bytes_field / of_bytes_field / sample of the two fields (lib/gf2k/gf2_128.h:168-190, lib/algebra/fp_generic.h:344-383)
inline void elt_to_bytes(int field, elt_t e, uint8_t out[16]) {
  if (field != LFGPU_FIELD_GF2_128) e = fp_from_mont(e);
  memcpy(out, &e, 16);
}
inline bool elt_of_bytes(int field, const uint8_t in[16], elt_t& e) {
  memcpy(&e, in, 16);
  if (field == LFGPU_FIELD_GF2_128) return true;  // every 128-bit string is an element
  if (!h_fp_fits(e)) return false;
  e = h_fp_to_mont(e);
  return true;
}
template <class Fill>
inline elt_t elt_sample(int field, Fill fill) {  // rejection sampling for Fp128 (exact_bits = 128: no masking)
  for (;;) {
    uint8_t b[16];
    fill(b, 16);
    elt_t e;
    if (elt_of_bytes(field, b, e)) return e;
  }
}

struct Ts {
  const lfgpu_transcript_ops* o;
  void* u;
  int field = LFGPU_FIELD_GF2_128;
  void write_bytes(const uint8_t* d, size_t n) const { o->write_bytes(u, d, n); }
  void write_elt(elt_t e) const {
    uint8_t b[16];
    elt_to_bytes(field, e, b);
    o->write_elt(u, b);
  }
  void write_array(const elt_t* e, size_t n) const {
    if (field == LFGPU_FIELD_GF2_128) {
      o->write_elt_array(u, (const uint8_t*)e, n);
      return;
    }
    std::vector<uint8_t> b(16 * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) elt_to_bytes(field, e[i], &b[16 * i]);
    o->write_elt_array(u, b.data(), n);
  }
  elt_t elt() const {
    return elt_sample(field, [&](uint8_t* b, size_t n) { o->gen_bytes(u, b, n); });
  }
  size_t nat(size_t n) const {  // RandomEngine::nat (lib/random/random.h:57-87): rejection sampling under a bit mask
    size_t l = 0, mask = 0;
    for (size_t nn = n; nn; nn >>= 8) ++l;
    while ((n & mask) != n) mask = (mask << 1) | 1;
    for (;;) {
      uint8_t b[8] = {0};
      o->gen_bytes(u, b, l);
      size_t r = 0;
      for (size_t i = 0; i < l; ++i) r |= (size_t)b[i] << (8 * i);
      r &= mask;
      if (r < n) return r;
    }
  }
  void choose(size_t n, size_t k, size_t* res) const {  // RandomEngine::choose (:89-105): partial Fisher-Yates
    std::vector<size_t> A(n);
    for (size_t i = 0; i < n; ++i) A[i] = i;
    for (size_t i = 0; i < k; ++i) {
      const size_t j = i + nat(n - i);
      std::swap(A[i], A[j]);
      res[i] = A[i];
    }
  }
};

inline size_t layer_size(size_t logw) { return 4 * logw + 3; }  // PadLayout::layer_size (zk_common.h:210-222)
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

// ------------------------------------------------------------------ circuit
extern "C" int lfgpu_circuit_from_lfc1(lfgpu_ctx* c, const uint8_t* b, size_t len, lfgpu_circuit** out) {
  if (!c || !b || !out) return LFGPU_ERR_ARG;
  size_t pos = 0;
  auto need = [&](size_t n) { return len - pos >= n; };
  auto num = [&](size_t* v) {  // 3-byte little-endian (circuit_reader.h:217-233)
    if (!need(3)) return false;
    *v = (size_t)b[pos] | (size_t)b[pos + 1] << 8 | (size_t)b[pos + 2] << 16;
    pos += 3;
    return true;
  };
  if (len < 1 || b[0] != 1) return lf_fail(c, LFGPU_ERR_ARG, "LFC1: bad version byte");
  pos = 1;
  size_t fid, nv, nc, npub, sfb, nin, nl, nk;
  if (!num(&fid) || !num(&nv) || !num(&nc) || !num(&npub) || !num(&sfb) || !num(&nin) || !num(&nl) || !num(&nk))
    return lf_fail(c, LFGPU_ERR_ARG, "LFC1: truncated header");
  if (fid != LFGPU_FIELD_GF2_128 && fid != LFGPU_FIELD_FP128 && fid != LFGPU_FIELD_P256)
    return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "LFC1: field id %zu (the ZK driver handles GF2_128 = 4, Fp128 = 6 and Fp256Base = 1)", fid);
  const int field = (int)fid;
  const size_t esz = field == LFGPU_FIELD_P256 ? 32 : 16;  // Field::kBytes
  if (nc != 1) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "LFC1: nc = %zu copies (logc must be 0)", nc);
  // CircuitReader::read_header's sanity checks (lib/proto/circuit_reader.h:104-110): an oversized subfield_boundary would
  // mark every witness row subfield-only, including the rows that hold the full-field sumcheck pads
  if (npub > nin || sfb > nin || nv == 0 || nl == 0 || nl > 10000 /* CircuitIO::kMaxLayers */) return lf_fail(c, LFGPU_ERR_ARG, "LFC1: inconsistent header");
  if (nk > (len - pos) / esz) return lf_fail(c, LFGPU_ERR_ARG, "LFC1: truncated constant table");
  std::vector<elt_t> kvec((nk ? nk : 1) * (esz / 16));  // Fp256Base: nk 32-byte elements in the same storage
  for (size_t i = 0; i < nk; ++i) {  // of_bytes_field: kBytes little-endian bytes (prime fields: canonical value -> Montgomery)
    const bool fits = field == LFGPU_FIELD_P256 ? h256_of_bytes(b + pos + 32 * i, reinterpret_cast<elt32_t*>(kvec.data())[i])
                                                : elt_of_bytes(field, b + pos + 16 * i, kvec[i]);
    if (!fits) return lf_fail(c, LFGPU_ERR_ARG, "LFC1: constant %zu is not a field element", i);
  }
  pos += esz * nk;
  std::unique_ptr<lfgpu_circuit> C(new lfgpu_circuit());
  C->c = c;
  size_t nterms = 0, nout = nv;
  std::vector<corner4> corners;
  const double t_begin = now_ms();
  double t_upload = 0;
  for (size_t ly = 0; ly < nl; ++ly) {
    size_t logw, nw, nq;
    if (!num(&logw) || !num(&nw) || !num(&nq)) return lf_fail(c, LFGPU_ERR_ARG, "LFC1: truncated layer header");
    // read_layers (circuit_reader.h:161-168): lw in (0, kMaxBindings], 0 < nw, lw <= nw <= 2^lw, nq > 0
    if (logw > kMaxBindings || logw == 0 || nw == 0 || nw < logw || nw > ((size_t)1 << logw) || nq == 0 || !need(12 * nq))
      return lf_fail(c, LFGPU_ERR_ARG, "LFC1: bad layer %zu", ly);
    corners.resize(nq);
    int64_t acc[3] = {0, 0, 0};
    const uint8_t* q = b + pos;  // 12 * nq bytes are there (checked above): four 3-byte little-endian numbers per term
    size_t hmax = 0;
    for (size_t t = 0; t < nq; ++t, q += 12) {
      u32 v[4];
      for (int j = 0; j < 4; ++j) v[j] = (u32)q[3 * j] | (u32)q[3 * j + 1] << 8 | (u32)q[3 * j + 2] << 16;
      for (int j = 0; j < 3; ++j) {  // delta with the sign in the LSB (circuit_writer.h:103-114)
        const int64_t d = (int64_t)(v[j] >> 1);
        acc[j] += (v[j] & 1) ? -d : d;
      }
      if (acc[0] < 0 || (size_t)acc[0] >= nout || acc[1] < 0 || (size_t)acc[1] >= nw || acc[2] < 0 || (size_t)acc[2] >= nw || v[3] >= nk)
        return lf_fail(c, LFGPU_ERR_ARG, "LFC1: layer %zu term %zu out of range", ly, t);
      corners[t] = corner4{(u32)acc[0], (u32)acc[1], (u32)acc[2], v[3]};
      hmax = std::max<size_t>(hmax, (size_t)std::max(acc[1], acc[2]));
    }
    pos += 12 * nq;
    lfgpu_circuit::Layer L{logw, nw, nq, nullptr};
    const double tu0 = now_ms();
    LF_TRY(lf_quad_upload_corners(c, field, nq, corners.data(), hmax, nk, kvec.data(), nout, &L.q));  // indices range-checked above
    t_upload += now_ms() - tu0;
    C->layers.push_back(L);
    nterms += nq;
    nout = nw;
  }
  if (!need(32) || pos + 32 != len) return lf_fail(c, LFGPU_ERR_ARG, "LFC1: bad trailer");
  if (nout != nin) return lf_fail(c, LFGPU_ERR_ARG, "LFC1: input layer width %zu != ninputs %zu", nout, nin);
  if (getenv("LFGPU_VERBOSE"))
    fprintf(stderr, "lfgpu circuit_from_lfc1: %zu terms in %zu layers: %.1f ms (decode %.1f, lfgpu_quad_upload %.1f)\n", nterms, nl, now_ms() - t_begin,
            now_ms() - t_begin - t_upload, t_upload);
  lfgpu_circuit_info& I = C->info;
  I.field = field;
  I.nv = nv; I.nc = nc; I.npub_in = npub; I.subfield_boundary = sfb; I.ninputs = nin; I.nl = nl; I.nterms = nterms;
  C->zeros = std::make_shared<const std::vector<uint8_t>>(nterms, (uint8_t)0);
  I.logv = lf_log2(nv);
  memcpy(I.id, b + pos, 32);
  *out = C.release();
  return LFGPU_OK;
}
// A second handle on an uploaded circuit for another context of the same device: the device arrays of every layer and the
// preamble's zero run are shared (reference-counted; either handle may be freed first), the per-proof caches are the handle's own.
extern "C" int lfgpu_circuit_share(lfgpu_ctx* c, const lfgpu_circuit* src, lfgpu_circuit** out) {
  if (!c || !src || !out) return LFGPU_ERR_ARG;
  if (c->device != src->c->device) return lf_fail(c, LFGPU_ERR_ARG, "circuit_share: the contexts are on different devices");
  LF_HIP(c, hipSetDevice(c->device));
  LF_HIP(c, hipStreamSynchronize(src->c->stream));  // the upload is complete (lfgpu_circuit_from_lfc1 synchronises; cheap)
  std::unique_ptr<lfgpu_circuit> C(new lfgpu_circuit());
  C->c = c;
  C->info = src->info;
  C->zeros = src->zeros;
  for (const auto& l : src->layers) {
    lfgpu_circuit::Layer L{l.logw, l.nw, l.nterms, nullptr};
    LF_TRY(lf_quad_share(c, l.q, &L.q));
    C->layers.push_back(L);
  }
  *out = C.release();
  return LFGPU_OK;
}
extern "C" int lfgpu_circuit_get_info(const lfgpu_circuit* C, lfgpu_circuit_info* info) {
  if (!C || !info) return LFGPU_ERR_ARG;
  *info = C->info;
  return LFGPU_OK;
}
extern "C" int lfgpu_circuit_layer_info(const lfgpu_circuit* C, size_t layer, size_t* logw, size_t* nw, size_t* nterms) {
  if (!C || layer >= C->layers.size()) return LFGPU_ERR_ARG;
  if (logw) *logw = C->layers[layer].logw;
  if (nw) *nw = C->layers[layer].nw;
  if (nterms) *nterms = C->layers[layer].nterms;
  return LFGPU_OK;
}
extern "C" int lfgpu_circuit_free(lfgpu_circuit* C) {
  if (!C) return LFGPU_ERR_ARG;
  delete C;
  return LFGPU_OK;
}

// ------------------------------------------------------------------ ZkProver
struct lfgpu_zk_prover {
  lfgpu_ctx* c = nullptr;
  const lfgpu_circuit* C = nullptr;
  Zk256* z256 = nullptr;  // Fp256Base circuits: the whole prover lives in zk256.hip
  lfgpu_ligero_param param{};
  size_t npub = 0, n_witness = 0, pad_size = 0;
  struct LayerPad {  // Proof-shaped pad (zk_prover.h:152-188): hp[hand][round] = {t0, t2}, wc[2]
    std::vector<elt_t> hp[2];
    elt_t wc[2];
  };
  std::vector<LayerPad> pad, proof;  // proof: the padded (transmitted) values, same shape
  std::vector<elt_t> aux;            // ProofAux::bound_quad per layer
  std::vector<size_t> lqc;
  lfgpu_ligero_prover* lp = nullptr;
  uint8_t root[32] = {0};
  std::vector<elt_t> y_ldt, y_dot, y_q0, y_q2, req;
  std::vector<uint8_t> nonces, path;
  size_t npath = 0;
  bool have_proof = false;
  mutable std::vector<uint8_t> wire;  // ZkProof::write bytes of the held proof (lfgpu_zk_proof_write fills it once)
  mutable bool wire_valid = false;
  // subfield solver for the wire format (GF2_128::solve, lib/gf2k/gf2_128.h:496-508): echelon rows of beta
  struct Row {
    elt_t v;
    u32 comb;
    int pivot;
  };
  std::vector<Row> ech;  // REDUCED echelon form: a row's pivot bit is clear in every other row
  // table form of the same solve: the pivot bits of e, taken as two bytes, index the XOR of the rows (and of their combinations)
  // those bits select -- 2 lookups per element instead of a 16-step elimination (35 000 opened elements in an mdoc hash proof)
  elt_t ech_v[2][256];
  u32 ech_c[2][256];
  // device buffers of the layer inputs (eval_circuit) and the circuit output
  std::vector<void*> d_in;
  void* d_V = nullptr;
  void* h_V = nullptr;  // pinned: outputs (nv elements) then the assert-zero flag, read back without blocking the host
  double ms[6] = {0, 0, 0, 0, 0, 0};
  // lfgpu_zk_prover_set_comm: Ligero tableaux of at least comm_min_bytes are committed with their rows sharded over the
  // communicator's GPUs (lfgpu_ligero_commit_sharded); everything else -- and the whole sumcheck -- runs replicated
  bool have_comm = false;
  lfgpu_comm_ops comm{};
  size_t comm_min_bytes = 0;
  ~lfgpu_zk_prover() {
    if (z256) zk256_free(z256);
    if (lp) lfgpu_ligero_free(lp);
    // the layers' wire values are functions of the witness: scrubbed before the memory goes back to the allocator (as the
    // Ligero tableau is, lfgpu_ligero_free), and so are the host copies of the pads
    if (c && C && !d_in.empty()) {
      for (size_t l = 0; l < d_in.size(); ++l)
        if (d_in[l]) (void)hipMemsetAsync(d_in[l], 0, C->layers[l].nw * 16, c->stream);
      if (d_V) (void)hipMemsetAsync(d_V, 0, C->info.nv * 16, c->stream);
      (void)hipStreamSynchronize(c->stream);
    }
    for (void* p : d_in)
      if (p) (void)hipFree(p);
    if (d_V) (void)hipFree(d_V);
    if (h_V) {
      if (C) memset(h_V, 0, C->info.nv * 16 + 16);
      (void)hipHostFree(h_V);
    }
    for (auto& P : pad) {
      for (auto& v : P.hp) std::fill(v.begin(), v.end(), elt_t{0, 0});
      P.wc[0] = P.wc[1] = elt_t{0, 0};
    }
  }
};

namespace {
int top_bit(elt_t v) { return v.hi ? 64 + (63 - __builtin_clzll(v.hi)) : v.lo ? 63 - __builtin_clzll(v.lo) : -1; }
bool bit_of(elt_t v, int j) { return j >= 64 ? (v.hi >> (j - 64)) & 1 : (v.lo >> j) & 1; }

void build_subfield_solver(lfgpu_zk_prover* zk, const GfHostCtx* g) {
  for (unsigned i = 0; i < g->sub_bits; ++i) {
    elt_t v = g->beta[i];
    u32 comb = 1u << i;
    for (const auto& r : zk->ech)
      if (bit_of(v, r.pivot)) {
        v = gf_add(v, r.v);
        comb ^= r.comb;
      }
    zk->ech.push_back({v, comb, top_bit(v)});  // beta is a basis: v != 0
  }
  // back-substitute: every pivot bit survives in its own row only, so the pivot bits of an element ARE its elimination pattern
  for (size_t i = 0; i < zk->ech.size(); ++i)
    for (size_t j = 0; j < zk->ech.size(); ++j)
      if (j != i && bit_of(zk->ech[j].v, zk->ech[i].pivot)) {
        zk->ech[j].v = gf_add(zk->ech[j].v, zk->ech[i].v);
        zk->ech[j].comb ^= zk->ech[i].comb;
      }
  for (int half = 0; half < 2; ++half)
    for (unsigned m = 0; m < 256; ++m) {
      elt_t v{0, 0};
      u32 cmb = 0;
      for (unsigned b = 0; b < 8; ++b) {
        const size_t r = 8 * half + b;
        if (((m >> b) & 1) && r < zk->ech.size()) {
          v = gf_add(v, zk->ech[r].v);
          cmb ^= zk->ech[r].comb;
        }
      }
      zk->ech_v[half][m] = v;
      zk->ech_c[half][m] = cmb;
    }
}
// (residue, coordinates): residue == 0 iff e lies in the subfield, and then e = sum_i bit_i(u) beta_i
std::pair<elt_t, u32> solve_subfield(const lfgpu_zk_prover* zk, elt_t e) {
  if (zk->ech.size() > 16) {  // (a 32-bit subfield: the plain elimination; the rows are reduced, the order does not matter)
    u32 u = 0;
    for (const auto& r : zk->ech)
      if (bit_of(e, r.pivot)) {
        e = gf_add(e, r.v);
        u ^= r.comb;
      }
    return {e, u};
  }
  unsigned m = 0;
  for (size_t r = 0; r < zk->ech.size(); ++r) m |= (unsigned)bit_of(e, zk->ech[r].pivot) << r;
  const elt_t res = gf_add(e, gf_add(zk->ech_v[0][m & 255], zk->ech_v[1][m >> 8]));
  return {res, zk->ech_c[0][m & 255] ^ zk->ech_c[1][m >> 8]};
}

struct RoundCtx {  // round_h of the padded prover (prover_layers.h:320-329): transmit poly - pad
  const HostField* F;
  const Ts* tst;
  const lfgpu_zk_prover::LayerPad* pad;
  lfgpu_zk_prover::LayerPad* out;
};
void zk_round_cb(void* user, size_t hand, size_t rnd, const uint64_t ev[3][2], uint64_t chal[2]) {
  RoundCtx* r = (RoundCtx*)user;
  const elt_t t0 = r->F->sub(elt_t{ev[0][0], ev[0][1]}, r->pad->hp[hand][2 * rnd]);
  const elt_t t2 = r->F->sub(elt_t{ev[2][0], ev[2][1]}, r->pad->hp[hand][2 * rnd + 1]);
  r->out->hp[hand][2 * rnd] = t0;
  r->out->hp[hand][2 * rnd + 1] = t2;
  r->tst->write_elt(t0);
  r->tst->write_elt(t2);
  const elt_t c = r->tst->elt();
  chal[0] = c.lo;
  chal[1] = c.hi;
}
}  // namespace

namespace {
// ---- ZkCommon::verifier_constraints (lib/zk/zk_common.h:49-136) + input_constraint (:406-439), shared by the prover
// (aux = the bound quads the sumcheck prover recorded) and the verifier (aux = nullptr: Quad::bind_gh_all on the device).
// Replays the verifier's side of the sumcheck on the transcript and returns the sparse rows of A (all but the dense
// private-input block of the last constraint), b, and the EQ vector of the input constraint over all inputs.
struct LinTerm {
  size_t c, w;
  elt_t k;
};
struct ConstraintSet {
  std::vector<LinTerm> a;
  std::vector<elt_t> b;       // one entry per constraint
  std::vector<elt_t> eq_in;   // EQ(g0, i) + alpha EQ(g1, i) for the npub public inputs (folded into b)
  const elt_t* d_eq = nullptr;  // the whole table, i < ninputs, on the device: dense coefficients of the last constraint
  size_t n = 0;               // number of constraints; the dense one is n - 1
};
int build_constraints(lfgpu_ctx* c, const lfgpu_circuit* C, const HostField& F, const Ts& ts, const std::vector<lfgpu_zk_prover::LayerPad>& proof,
                      const std::vector<elt_t>* aux, const elt_t* pub, ConstraintSet& out) {
  const lfgpu_circuit_info& I = C->info;
  const size_t nl = C->layers.size(), npub = I.npub_in;
  std::vector<elt_t> G[2], gh[2];
  for (size_t i = 0; i < kMaxBindings; ++i) (void)ts.elt();  // begin_circuit: Q (unused for logc = 0), then G
  G[0].resize(kMaxBindings);
  for (size_t i = 0; i < kMaxBindings; ++i) G[0][i] = ts.elt();
  G[1] = G[0];
  size_t logv = I.logv, ci = 0, pi = I.ninputs - npub;
  elt_t claims[2] = {elt_t{0, 0}, elt_t{0, 0}};
  std::vector<elt_t> sym;
  struct Deferred {
    size_t ci, acp;  // constraint, position of its claim-pad terms in out.a
    elt_t wc0, wc1;
  };
  std::vector<Deferred> deferred;
  const bool batch_gh = nl <= LF_GH_BATCH_MAX;
  for (size_t ly = 0; ly < nl; ++ly) {
    const auto& L = C->layers[ly];
    const size_t logw = L.logw;
    const elt_t alpha = ts.elt(), beta = ts.elt();
    const size_t n = 3 + layer_size(logw);  // ovp_layer_size
    elt_t known{0, 0};
    sym.assign(n, elt_t{0, 0});
    auto axpy = [&](size_t var, elt_t kv, elt_t k) {  // Expression::axpy
      known = F.add(known, F.mul(k, kv));
      sym[var] = F.add(sym[var], k);
    };
    auto axmy = [&](size_t var, elt_t kv, elt_t k) {  // Expression::axmy
      known = F.sub(known, F.mul(k, kv));
      sym[var] = F.sub(sym[var], k);
    };
    axpy(0, claims[0], F.one);  // ConstraintBuilder::first
    axpy(1, claims[1], alpha);
    gh[0].assign(logw ? logw : 1, elt_t{0, 0});
    gh[1].assign(logw ? logw : 1, elt_t{0, 0});
    const auto& P = proof[ly];
    for (size_t rnd = 0; rnd < logw; ++rnd)
      for (int hand = 0; hand < 2; ++hand) {
        const size_t r = 2 * rnd + hand;
        const elt_t t0e = P.hp[hand][2 * rnd], t2e = P.hp[hand][2 * rnd + 1];
        ts.write_elt(t0e);
        ts.write_elt(t2e);
        const elt_t chal = ts.elt();
        gh[hand][rnd] = chal;
        elt_t lag[3];  // dot_interpolation coefficients: p(chal) = sum_i lag[i] p(P_i)
        for (int i = 0; i < 3; ++i) {
          elt_t num = F.one;
          for (int j = 0; j < 3; ++j)
            if (j != i) num = F.mul(num, F.sub(chal, F.pts[j]));
          lag[i] = F.mul(num, F.invden[i]);
        }
        axmy(3 + 2 * r, t0e, F.one);   // ConstraintBuilder::next: p(1) = claim - p(0)
        known = F.mul(known, lag[1]);  // scale
        for (auto& s : sym)
          if (s.lo | s.hi) s = F.mul(s, lag[1]);
        axpy(3 + 2 * r, t0e, lag[0]);
        axpy(3 + 2 * r + 1, t2e, lag[2]);
      }
    // EQ[Q,C] QUAD[R,L] (Eq::eval with logc = 0 is 1): the prover's aux, or Quad::bind_gh_all on the device.  The
    // verifier's value feeds only ConstraintBuilder::finalize, never the transcript, so the layers' sums are enqueued
    // back to back and finalize runs for all layers after ONE synchronisation below.
    elt_t eqq{0, 0};
    const bool defer = !aux && batch_gh;
    if (aux) {
      eqq = (*aux)[ly];
    } else {
      const uint64_t al[2] = {alpha.lo, alpha.hi}, be[2] = {beta.lo, beta.hi};
      if (defer) {
        LF_TRY(lf_quad_bind_gh_all_enqueue(L.q, logv, G[0].data(), G[1].data(), al, be, logw, L.nw, gh[0].data(), gh[1].data(),
                                           (u64*)((uint8_t*)c->mailbox_d + 512) + 4 * ly));
      } else {
        uint64_t bq[2];
        LF_TRY(lfgpu_quad_bind_gh_all(L.q, logv, G[0].data(), G[1].data(), al, be, logw, L.nw, gh[0].data(), gh[1].data(), bq));
        eqq = elt_t{bq[0], bq[1]};
      }
    }
    const size_t cp = 3 + 4 * logw;  // ConstraintBuilder::finalize
    const size_t a0 = out.a.size(), skip = ly == 0 ? 3 : 0;
    out.b.push_back(defer ? known : F.sub(F.mul(eqq, F.mul(P.wc[0], P.wc[1])), known));
    if (!defer) {
      sym[cp] = F.sub(sym[cp], F.mul(eqq, P.wc[1]));
      sym[cp + 1] = F.sub(sym[cp + 1], F.mul(eqq, P.wc[0]));
      sym[cp + 2] = F.sub(sym[cp + 2], eqq);
    }
    for (size_t i = skip; i < n; ++i) out.a.push_back({ci, pi + i - 3, sym[i]});
    if (defer) deferred.push_back({ci, a0 + cp - skip, P.wc[0], P.wc[1]});
    ++ci;
    ts.write_array(P.wc, 2);
    claims[0] = P.wc[0];
    claims[1] = P.wc[1];
    for (int h = 0; h < 2; ++h) {
      G[h].assign(kMaxBindings, elt_t{0, 0});
      for (size_t r = 0; r < logw; ++r) G[h][r] = gh[h][r];
    }
    logv = logw;
    pi += layer_size(logw);
  }
  if (!deferred.empty()) {  // the layers' bind_gh_all sums: one read-back, then finalize each layer
    std::vector<u64> w(4 * nl);
    LF_HIP(c, hipMemcpyAsync(w.data(), (uint8_t*)c->mailbox_d + 512, nl * 32, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(c, hipStreamSynchronize(c->stream));
    for (const Deferred& d : deferred) {
      uint64_t bq[2];
      lf_quad_bind_gh_all_fold(I.field, &w[4 * d.ci], bq);
      const elt_t eqq{bq[0], bq[1]};
      out.b[d.ci] = F.sub(F.mul(eqq, F.mul(d.wc0, d.wc1)), out.b[d.ci]);  // b held `known` so far
      out.a[d.acp].k = F.sub(out.a[d.acp].k, F.mul(eqq, d.wc1));
      out.a[d.acp + 1].k = F.sub(out.a[d.acp + 1].k, F.mul(eqq, d.wc0));
      out.a[d.acp + 2].k = F.sub(out.a[d.acp + 2].k, eqq);
    }
  }
  const elt_t alpha = ts.elt();
  out.a.push_back({ci, pi - 3, F.sub(elt_t{0, 0}, F.one)});  // input_constraint: -1, -alpha on the input layer's claim pads
  out.a.push_back({ci, pi - 2, F.sub(elt_t{0, 0}, alpha)});
  out.n = ci + 1;
  // EQ table over the inputs on the device: public part folded into b, private part = dense block of A
  out.eq_in.assign(npub, elt_t{0, 0});
  const size_t logn = C->layers[nl - 1].logw;
  if (c->zk_eq_bytes < I.ninputs * 16) {  // context-owned: stays valid until the next run on this context
    LF_HIP(c, hipStreamSynchronize(c->stream));
    if (c->zk_eq) hipFree(c->zk_eq);
    c->zk_eq = nullptr;
    c->zk_eq_bytes = 0;
    if (hipMalloc(&c->zk_eq, I.ninputs * 16) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "zk: EQ table over the inputs");
    c->zk_eq_bytes = I.ninputs * 16;
  }
  void* d_eq = c->zk_eq;
  const uint64_t al[2] = {alpha.lo, alpha.hi};
  LF_TRY(lfgpu_raw_eq2(c, I.field, logn, I.ninputs, gh[0].data(), gh[1].data(), al, d_eq));
  out.d_eq = (const elt_t*)d_eq;
  if (npub) LF_HIP(c, hipMemcpyAsync(out.eq_in.data(), d_eq, npub * 16, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  const auto& P = proof[nl - 1];
  elt_t pub_binding{0, 0};
  for (size_t i = 0; i < npub; ++i) pub_binding = F.add(pub_binding, F.mul(out.eq_in[i], pub[i]));
  out.b.push_back(F.sub(F.add(P.wc[0], F.mul(alpha, P.wc[1])), pub_binding));
  return LFGPU_OK;
}

// LigeroCommon::inner_product_vector (lib/ligero/ligero_param.h:382-421), host share: the sparse terms of A[nwqrow][w]
// -- the linear constraints' terms times alphal and the quadratic copy constraints (A[copy] += aq, A[original] -= aq) --
// as (flat index, value) pairs, sorted with duplicates folded.  The dense private-input block alphal[n-1] * EQ[npub + w]
// is built on the device (lfgpu_ligero_inner_product_rows); the sums commute, so the result is the reference's A.
void inner_product_sparse(const HostField& F, const lfgpu_ligero_param& p, const ConstraintSet& cs, const std::vector<elt_t>& alphal,
                          const std::vector<size_t>& lqc, const std::vector<elt_t>& alphaq, std::vector<uint64_t>& idx, std::vector<elt_t>& val) {
  std::vector<std::pair<uint64_t, elt_t>> t;
  t.reserve(cs.a.size() + 6 * p.nq);
  for (const LinTerm& l : cs.a) t.emplace_back((uint64_t)l.w, F.mul(l.k, alphal[l.c]));
  const size_t base = p.nwrow * p.w;
  const size_t Ax = base, Ay = base + p.nqtriples * p.w, Az = base + 2 * p.nqtriples * p.w;
  for (size_t iw = 0; iw < p.nq; ++iw) {
    const size_t off[3] = {Ax + iw, Ay + iw, Az + iw};
    for (int j = 0; j < 3; ++j) {
      const elt_t aq = alphaq[3 * iw + j];
      t.emplace_back((uint64_t)off[j], aq);
      t.emplace_back((uint64_t)lqc[3 * iw + j], F.sub(elt_t{0, 0}, aq));
    }
  }
  std::stable_sort(t.begin(), t.end(), [](const std::pair<uint64_t, elt_t>& a, const std::pair<uint64_t, elt_t>& b) { return a.first < b.first; });
  idx.clear();
  val.clear();
  for (const auto& e : t) {
    if (!idx.empty() && idx.back() == e.first) val.back() = F.add(val.back(), e.second);
    else {
      idx.push_back(e.first);
      val.push_back(e.second);
    }
  }
}
}  // namespace

extern "C" int lfgpu_zk_prover_new(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc,
                                   lfgpu_zk_prover** out) {
  if (!c || !C || !out || C->c != c) return LFGPU_ERR_ARG;
  std::unique_ptr<lfgpu_zk_prover> zk(new lfgpu_zk_prover());
  zk->c = c;
  zk->C = C;
  if (C->info.field == LFGPU_FIELD_P256) {
    LF_TRY(zk256_new(c, C, rateinv, nreq, block_enc, &zk->z256));
    *out = zk.release();
    return LFGPU_OK;
  }
  zk->npub = C->info.npub_in;
  zk->n_witness = C->info.ninputs - C->info.npub_in;
  for (const auto& l : C->layers) zk->pad_size += layer_size(l.logw);
  // ZkProof: LigeroParam(n_witness + pad_size, nl quadratic constraints, rate, nreq[, block_enc]) (zk_proof.h:63-76)
  LF_TRY(lfgpu_ligero_param_init(&zk->param, C->info.field, 4, zk->n_witness + zk->pad_size, C->info.nl, rateinv, nreq, block_enc));
  if (C->info.field == LFGPU_FIELD_GF2_128) {
    const GfHostCtx* g = lf_gf_ctx(c, 4);
    if (!g) return LFGPU_ERR_ARG;
    build_subfield_solver(zk.get(), g);
  }
  LF_HIP(c, hipSetDevice(c->device));
  zk->d_in.assign(C->layers.size(), nullptr);
  for (size_t l = 0; l < C->layers.size(); ++l)
    if (hipMalloc(&zk->d_in[l], C->layers[l].nw * 16) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "zk: layer %zu inputs", l);
  if (hipMalloc(&zk->d_V, C->info.nv * 16) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "zk: outputs");
  if (hipHostMalloc(&zk->h_V, C->info.nv * 16 + 16, hipHostMallocDefault) != hipSuccess) return lf_fail(c, LFGPU_ERR_NOMEM, "zk: pinned outputs");
  *out = zk.release();
  return LFGPU_OK;
}

extern "C" int lfgpu_zk_prover_set_comm(lfgpu_zk_prover* zk, const lfgpu_comm_ops* comm, size_t min_tableau_bytes) {
  if (!zk) return LFGPU_ERR_ARG;
  if (!comm) {
    zk->have_comm = false;
    if (zk->z256) zk256_set_comm(zk->z256, nullptr, 0);
    return LFGPU_OK;
  }
  if (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world || !comm->all_gather || !comm->all_to_all || !comm->broadcast)
    return lf_fail(zk->c, LFGPU_ERR_ARG, "zk_prover_set_comm: incomplete communicator");
  zk->have_comm = true;
  zk->comm = *comm;
  zk->comm_min_bytes = min_tableau_bytes;
  if (zk->z256) zk256_set_comm(zk->z256, &zk->comm, min_tableau_bytes);
  return LFGPU_OK;
}
extern "C" int lfgpu_zk_prover_param(const lfgpu_zk_prover* zk, lfgpu_ligero_param* p) {
  if (!zk || !p) return LFGPU_ERR_ARG;
  if (zk->z256) return zk256_param(zk->z256, p);
  *p = zk->param;
  return LFGPU_OK;
}

extern "C" int lfgpu_zk_commit(lfgpu_zk_prover* zk, const void* h_W, lfgpu_rng_fn rng, void* rng_user,
                               const lfgpu_transcript_ops* ts, uint8_t root_out[32]) {
  if (!zk || !h_W || !rng || !ts) return LFGPU_ERR_ARG;
  if (zk->z256) {
    // Fp256Base: with a communicator the ranks share ONE RandomEngine -- rank 0's draws of the whole commit (pads, then the
    // Ligero layout) are recorded and broadcast, the other ranks replay them; zk256_commit then shards the tableau's rows when
    // it is above the threshold (lig256_commit; the mdoc signature circuit's 2.5 MB tableau normally stays replicated)
    if (!(zk->have_comm && zk->comm.world > 1)) return zk256_commit(zk->z256, h_W, rng, rng_user, ts, root_out);
    std::vector<uint8_t> stream;
    LF_SCRUB_ON_EXIT(stream);
    alignas(16) unsigned char store[64];
    lfgpu_rng_fn r2 = rng;
    void* u2 = rng_user;
    if (zk->comm.rank == 0) {  // all draws first (no device work, no collective), so that the stream can go out before any collective
      lf_record_rng(rng, rng_user, &stream, &r2, &u2, store);
      const int rc = zk256_commit(zk->z256, h_W, r2, u2, ts, root_out, /*draws_only=*/true);
      if (lf_comm_bcast_blob(&zk->comm, stream)) return lf_fail(zk->c, LFGPU_ERR_HIP, "zk_commit: broadcast hook failed");
      if (rc) return rc;
    } else if (lf_comm_bcast_blob(&zk->comm, stream)) {
      return lf_fail(zk->c, LFGPU_ERR_HIP, "zk_commit: broadcast hook failed");
    }
    lf_replay_rng(&stream, &r2, &u2, store);
    return zk256_commit(zk->z256, h_W, r2, u2, ts, root_out);
  }
  const double t0 = now_ms();
  lfgpu_ctx* c = zk->c;
  const lfgpu_circuit* C = zk->C;
  const size_t nl = C->layers.size();
  const int field = C->info.field;
  const HostField F(c, field);
  // More than one rank (lfgpu_zk_prover_set_comm): every rank runs this function with the same arguments, but there is ONE
  // RandomEngine -- rank 0's.  Its pad draws are recorded and broadcast, the other ranks replay them (the Ligero commit does
  // the same for its own draws), so all ranks hold the same pads, the same commitment and, with their own copies of the
  // transcript, the same proof.
  const bool multi = zk->have_comm && zk->comm.world > 1;
  const bool shard_rows = multi && zk->param.nrow * zk->param.block_enc * 16 >= zk->comm_min_bytes;
  std::vector<uint8_t> pad_stream;
  LF_SCRUB_ON_EXIT(pad_stream);
  alignas(16) unsigned char rng_store[64];
  if (multi) {
    if (zk->comm.rank == 0) {
      lf_record_rng(rng, rng_user, &pad_stream, &rng, &rng_user, rng_store);
    } else {
      if (lf_comm_bcast_blob(&zk->comm, pad_stream)) return lf_fail(c, LFGPU_ERR_HIP, "zk_commit: broadcast hook failed");
      lf_replay_rng(&pad_stream, &rng, &rng_user, rng_store);
    }
  }
  auto draw = [&]() {  // RandomEngine::elt = Field::sample
    return elt_sample(field, [&](uint8_t* b, size_t n) { rng(rng_user, b, n); });
  };
  // witness = private inputs || pad; fill_pad draws, per layer: (t0, t2) for hand 0 then hand 1 of every round,
  // then wc0, wc1 and stores wc0*wc1 (zk_prover.h:152-188, logc = 0)
  std::vector<elt_t> Wv(zk->param.nw);  // witness || pads
  LF_SCRUB_ON_EXIT(Wv);
  memcpy(Wv.data(), (const elt_t*)h_W + zk->npub, zk->n_witness * 16);
  zk->pad.assign(nl, {});
  zk->lqc.assign(3 * nl, 0);
  size_t pi = zk->n_witness;
  for (size_t ly = 0; ly < nl; ++ly) {
    const size_t logw = C->layers[ly].logw;
    auto& P = zk->pad[ly];
    P.hp[0].resize(2 * logw);
    P.hp[1].resize(2 * logw);
    size_t w = pi;
    for (size_t j = 0; j < logw; ++j)
      for (int h = 0; h < 2; ++h) {
        P.hp[h][2 * j] = draw();
        P.hp[h][2 * j + 1] = draw();
        Wv[w++] = P.hp[h][2 * j];
        Wv[w++] = P.hp[h][2 * j + 1];
      }
    P.wc[0] = draw();
    P.wc[1] = draw();
    Wv[w++] = P.wc[0];
    Wv[w++] = P.wc[1];
    Wv[w++] = F.mul(P.wc[0], P.wc[1]);
    const size_t cp = pi + 4 * logw;  // setup_lqc (zk_common.h:149-160): claim_pad(0..2)
    zk->lqc[3 * ly] = cp;
    zk->lqc[3 * ly + 1] = cp + 1;
    zk->lqc[3 * ly + 2] = cp + 2;
    pi += layer_size(logw);
  }
  // (rank 0 first sends what the other ranks are waiting for, whatever went wrong here)
  if (multi && zk->comm.rank == 0 && lf_comm_bcast_blob(&zk->comm, pad_stream)) return lf_fail(c, LFGPU_ERR_HIP, "zk_commit: broadcast hook failed");
  if (pi != zk->param.nw) return lf_fail(c, LFGPU_ERR_ASSERT, "zk_commit: witness layout");
  const size_t sfb = C->info.subfield_boundary >= zk->npub ? C->info.subfield_boundary - zk->npub : 0;
  if (zk->lp) {
    lfgpu_ligero_free(zk->lp);
    zk->lp = nullptr;
  }
  zk->have_proof = false;
  zk->wire_valid = false;
  if (shard_rows) {
    LF_TRY(lfgpu_ligero_commit_sharded(c, field, 4, &zk->param, Wv.data(), sfb, zk->lqc.data(), rng, rng_user, &zk->comm, zk->root, &zk->lp));
  } else if (multi) {  // a small tableau stays whole on every rank (replicas): the one random stream still comes from rank 0
    lfgpu_comm_ops one = zk->comm;
    std::vector<uint8_t> lig_stream;
    LF_SCRUB_ON_EXIT(lig_stream);
    alignas(16) unsigned char st2[64];
    lfgpu_rng_fn r2 = rng;
    void* u2 = rng_user;
    if (zk->comm.rank == 0) lf_record_rng(rng, rng_user, &lig_stream, &r2, &u2, st2);
    else {
      if (lf_comm_bcast_blob(&one, lig_stream)) return lf_fail(c, LFGPU_ERR_HIP, "zk_commit: broadcast hook failed");
      lf_replay_rng(&lig_stream, &r2, &u2, st2);
    }
    const int rc = lfgpu_ligero_commit(c, field, 4, &zk->param, Wv.data(), sfb, zk->lqc.data(), r2, u2, zk->root, &zk->lp);
    // (also after a failed commit: the other ranks are waiting in this broadcast)
    if (zk->comm.rank == 0 && lf_comm_bcast_blob(&one, lig_stream)) return lf_fail(c, LFGPU_ERR_HIP, "zk_commit: broadcast hook failed");
    if (rc) return rc;
  } else {
    LF_TRY(lfgpu_ligero_commit(c, field, 4, &zk->param, Wv.data(), sfb, zk->lqc.data(), rng, rng_user, zk->root, &zk->lp));
  }
  ts->write_bytes(ts->user, zk->root, 32);  // LigeroTranscript::write_commitment
  if (root_out) memcpy(root_out, zk->root, 32);
  zk->ms[0] = now_ms() - t0;
  return LFGPU_OK;
}

extern "C" int lfgpu_zk_prove(lfgpu_zk_prover* zk, const void* h_W, const lfgpu_transcript_ops* tso, int* ok) {
  if (!zk || !h_W || !tso || !ok) return LFGPU_ERR_ARG;
  if (zk->z256) return zk256_prove(zk->z256, h_W, tso, ok);
  lfgpu_ctx* c = zk->c;
  if (!zk->lp) return lf_fail(c, LFGPU_ERR_ARG, "zk_prove: must run commit before prove");
  const double t_start = now_ms();
  const lfgpu_circuit* C = zk->C;
  const lfgpu_circuit_info& I = C->info;
  const size_t nl = C->layers.size();
  const elt_t* W = (const elt_t*)h_W;
  const HostField F(c, I.field);
  const Ts ts{tso, tso->user, I.field};
  *ok = 0;
  zk->have_proof = false;
  zk->wire_valid = false;
  LF_HIP(c, hipSetDevice(c->device));

  // eval_circuit (prover_layers.h:52-104): layer inputs stay resident for the sumcheck
  // The device works through all layers back to back (assert-zero failures and the outputs are read once at the end)
  // while the host hashes the Fiat-Shamir preamble below -- SHA-256 over nterms zero bytes is sequential host work the
  // reference's transcript format fixes, and the evaluation does not depend on it.
  double t0 = now_ms();
  const elt_t* V = (const elt_t*)zk->h_V;
  const int* failed = (const int*)((const uint8_t*)zk->h_V + I.nv * 16);
  {
    LF_HIP(c, hipMemcpyAsync(zk->d_in[nl - 1], W, I.ninputs * 16, hipMemcpyHostToDevice, c->stream));
    int* d_fail = (int*)((uint8_t*)c->mailbox_d + 128);
    LF_HIP(c, hipMemsetAsync(d_fail, 0, 4, c->stream));
    for (size_t l = nl; l-- > 0;) LF_TRY(lf_eval_quad_async(C->layers[l].q, zk->d_in[l], l ? zk->d_in[l - 1] : zk->d_V, d_fail));
    LF_HIP(c, hipMemcpyAsync(zk->h_V, zk->d_V, I.nv * 16, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(c, hipMemcpyAsync((uint8_t*)zk->h_V + I.nv * 16, d_fail, 4, hipMemcpyDeviceToHost, c->stream));
  }
  const double t_enq = now_ms() - t0;

  // initialize_sumcheck_fiat_shamir (zk_common.h:163-180)
  ts.write_bytes(I.id, 32);
  for (size_t i = 0; i < zk->npub; ++i) ts.write_elt(W[i]);
  ts.write_elt(elt_t{0, 0});
  ts.write_bytes(C->zeros->data(), I.nterms);
  void* cl = tso->clone(tso->user);
  if (!cl) {
    hipStreamSynchronize(c->stream);
    return lf_fail(c, LFGPU_ERR_NOMEM, "zk_prove: transcript clone");
  }
  struct CloneGuard {
    const lfgpu_transcript_ops* o;
    void* u;
    ~CloneGuard() { o->free_clone(u); }
  } cg{tso, cl};
  const Ts tst{tso, cl, I.field};

  t0 = now_ms();
  LF_HIP(c, hipStreamSynchronize(c->stream));
  if (*failed) return LFGPU_OK;  // an assert-zero term is non-zero: eval_circuit returns nullptr
  for (size_t i = 0; i < I.nv; ++i)
    if (V[i].lo | V[i].hi) return LFGPU_OK;  // "V->v_[i] != F.zero()"
  zk->ms[2] = t_enq + now_ms() - t0;  // what the evaluation adds to the wall time: enqueue + the wait left after the hashing

  // padded sumcheck (ProverLayers::prove with pad, transcript copy tst)
  t0 = now_ms();
  zk->proof.assign(nl, {});
  zk->aux.assign(nl, elt_t{0, 0});
  std::vector<elt_t> G[2];
  {
    for (size_t i = 0; i < kMaxBindings; ++i) (void)tst.elt();  // begin_circuit: Q then G (transcript_sumcheck.h:49-52)
    G[0].resize(kMaxBindings);
    for (size_t i = 0; i < kMaxBindings; ++i) G[0][i] = tst.elt();
    G[1] = G[0];
  }
  size_t logv = I.logv;
  uint64_t WC[2][2] = {{0, 0}, {0, 0}};
  std::vector<uint64_t> gout;
  for (size_t ly = 0; ly < nl; ++ly) {
    const auto& L = C->layers[ly];
    const elt_t alpha = tst.elt(), beta = tst.elt();
    auto& P = zk->proof[ly];
    P.hp[0].resize(2 * L.logw);
    P.hp[1].resize(2 * L.logw);
    RoundCtx rc{&F, &tst, &zk->pad[ly], &P};
    gout.assign(4 * L.logw + 2, 0);
    uint64_t wc_out[2][2], bq[2];
    const uint64_t al[2] = {alpha.lo, alpha.hi}, be[2] = {beta.lo, beta.hi};
    LF_TRY(lfgpu_sumcheck_layer(L.q, logv, G[0].data(), G[1].data(), al, be, L.logw, L.nw, zk->d_in[ly], WC, zk_round_cb, &rc, wc_out,
                                gout.data(), bq));
    // end_layer (:331-344): transmit wc - pad
    P.wc[0] = F.sub(elt_t{wc_out[0][0], wc_out[0][1]}, zk->pad[ly].wc[0]);
    P.wc[1] = F.sub(elt_t{wc_out[1][0], wc_out[1][1]}, zk->pad[ly].wc[1]);
    tst.write_array(P.wc, 2);
    zk->aux[ly] = elt_t{bq[0], bq[1]};
    memcpy(WC, wc_out, sizeof(WC));
    for (int h = 0; h < 2; ++h) {
      G[h].assign(kMaxBindings, elt_t{0, 0});
      for (size_t r = 0; r < L.logw; ++r) G[h][r] = elt_t{gout[(h * L.logw + r) * 2], gout[(h * L.logw + r) * 2 + 1]};
    }
    logv = L.logw;
  }
  zk->ms[3] = now_ms() - t0;

  // verifier_constraints with aux (zk_common.h:49-136): replay the verifier symbolically on the ORIGINAL transcript
  t0 = now_ms();
  ConstraintSet cs;
  LF_TRY(build_constraints(c, C, F, ts, zk->proof, &zk->aux, W, cs));
  const size_t nconstraints = cs.n;
  const lfgpu_ligero_param& p = zk->param;
  zk->ms[4] = now_ms() - t0;

  // LigeroProver::prove (ligero_prover.h:84-146)
  t0 = now_ms();
  {
    uint8_t hash_of_A[32] = {0xde, 0xad, 0xbe, 0xef};  // zk_prover.h:143
    ts.write_bytes(hash_of_A, 32);
    std::vector<elt_t> u_ldt(p.nwqrow);
    for (auto& e : u_ldt) e = ts.elt();
    zk->y_ldt.assign(p.block, elt_t{0, 0});
    static const bool verbose = getenv("LFGPU_VERBOSE") != nullptr;
    double tq[6] = {now_ms(), 0, 0, 0, 0, 0};
    LF_TRY(lfgpu_ligero_low_degree_proof(zk->lp, u_ldt.data(), zk->y_ldt.data()));
    tq[1] = now_ms();
    std::vector<elt_t> alphal(nconstraints), alphaq(3 * p.nq);
    for (auto& e : alphal) e = ts.elt();
    for (auto& e : alphaq) e = ts.elt();
    std::vector<uint64_t> a_idx;
    std::vector<elt_t> a_val;
    inner_product_sparse(F, p, cs, alphal, zk->lqc, alphaq, a_idx, a_val);
    zk->y_dot.assign(p.dblock, elt_t{0, 0});
    tq[2] = now_ms();
    {
      const elt_t ad = alphal[cs.n - 1];
      const uint64_t sc[2] = {ad.lo, ad.hi};
      LF_TRY(lfgpu_ligero_dot_proof_sparse(zk->lp, cs.d_eq + zk->npub, zk->n_witness, sc, a_idx.data(), a_val.data(), a_idx.size(), zk->y_dot.data()));
    }
    tq[3] = now_ms();
    std::vector<elt_t> u_quad(p.nqtriples ? p.nqtriples : 1);
    for (size_t i = 0; i < p.nqtriples; ++i) u_quad[i] = ts.elt();
    zk->y_q0.assign(p.r, elt_t{0, 0});
    zk->y_q2.assign(p.dblock - p.block, elt_t{0, 0});
    LF_TRY(lfgpu_ligero_quadratic_proof(zk->lp, u_quad.data(), zk->y_q0.data(), zk->y_q2.data()));
    tq[4] = now_ms();
    ts.write_array(zk->y_ldt.data(), zk->y_ldt.size());
    ts.write_array(zk->y_dot.data(), zk->y_dot.size());
    ts.write_array(zk->y_q0.data(), zk->y_q0.size());
    ts.write_array(zk->y_q2.data(), zk->y_q2.size());
    std::vector<size_t> idx(p.nreq);
    ts.choose(p.block_ext, p.nreq, idx.data());
    zk->req.assign(p.nrow * p.nreq, elt_t{0, 0});
    zk->nonces.assign(p.nreq * 32, 0);
    const size_t cap = p.nreq * p.mc_pathlen + 1;
    zk->path.assign(cap * 32, 0);
    LF_TRY(lfgpu_ligero_open(zk->lp, idx.data(), zk->req.data(), zk->nonces.data(), zk->path.data(), cap, &zk->npath));
    tq[5] = now_ms();
    if (verbose)
      fprintf(stderr, "lfgpu zk ligero_prove: ldt %.2f ms | sparse terms of A %.2f | dot %.2f | quad %.2f | challenges+open %.2f\n", tq[1] - tq[0],
              tq[2] - tq[1], tq[3] - tq[2], tq[4] - tq[3], tq[5] - tq[4]);
  }
  zk->ms[5] = now_ms() - t0;
  zk->ms[1] = now_ms() - t_start;
  zk->have_proof = true;
  *ok = 1;
  return LFGPU_OK;
}

extern "C" int lfgpu_zk_proof_write(const lfgpu_zk_prover* zk, uint8_t* buf, size_t cap, size_t* nbytes) {
  if (!zk || !nbytes) return LFGPU_ERR_ARG;
  if (zk->z256) return zk256_proof_write(zk->z256, buf, cap, nbytes);
  if (!zk->have_proof) return lf_fail(zk->c, LFGPU_ERR_ARG, "zk_proof_write: no proof");
  std::vector<uint8_t>& o = zk->wire;  // serialised once per proof: the size query and the copy share it
  if (zk->wire_valid) {
    *nbytes = o.size();
    if (buf) {
      if (cap < o.size()) return lf_fail(zk->c, LFGPU_ERR_ARG, "zk_proof_write: buffer too small (%zu < %zu)", cap, o.size());
      memcpy(buf, o.data(), o.size());
    }
    return LFGPU_OK;
  }
  o.clear();
  const int field = zk->C->info.field;
  auto pute = [&](elt_t e) {
    uint8_t b[16];
    elt_to_bytes(field, e, b);
    o.insert(o.end(), b, b + 16);
  };
  auto putsz = [&](size_t g) {  // write_size: 4 bytes LE (zk_proof.h:211-216)
    for (int i = 0; i < 4; ++i) o.push_back((uint8_t)(g >> (8 * i)));
  };
  o.insert(o.end(), zk->root, zk->root + 32);  // write_com
  for (size_t ly = 0; ly < zk->proof.size(); ++ly) {  // write_sc_proof: p(0) and p(2) of both hands per round, then wc
    const auto& P = zk->proof[ly];
    const size_t logw = zk->C->layers[ly].logw;
    for (size_t wi = 0; wi < logw; ++wi)
      for (int k = 0; k < 2; ++k) {
        pute(P.hp[0][2 * wi + k]);
        pute(P.hp[1][2 * wi + k]);
      }
    pute(P.wc[0]);
    pute(P.wc[1]);
  }
  for (elt_t e : zk->y_ldt) pute(e);  // write_com_proof
  for (elt_t e : zk->y_dot) pute(e);
  for (elt_t e : zk->y_q0) pute(e);
  for (elt_t e : zk->y_q2) pute(e);
  o.insert(o.end(), zk->nonces.begin(), zk->nonces.end());
  // opened columns: alternating runs of full-field / subfield elements, run-length prefixed (:156-178)
  constexpr size_t kMaxRunLen = (size_t)1 << 25;
  const size_t nreq_elts = zk->req.size();
  // GF2_128: solve every opened element against the subfield basis ONCE (residue == 0 iff it lies in the subfield; the
  // coordinates are its 2-byte image)
  std::vector<u32> sub_coord;
  std::vector<uint8_t> sub_flag;
  if (field == LFGPU_FIELD_GF2_128) {
    sub_coord.resize(nreq_elts);
    sub_flag.resize(nreq_elts);
    for (size_t i = 0; i < nreq_elts; ++i) {
      const auto r = solve_subfield(zk, zk->req[i]);
      sub_flag[i] = (r.first.lo | r.first.hi) == 0;
      sub_coord[i] = r.second;
    }
  }
  auto is_sub = [&](size_t i) { return field != LFGPU_FIELD_GF2_128 ? true : sub_flag[i] != 0; };
  o.reserve(o.size() + nreq_elts * 16 + 32 * zk->npath + 64);
  size_t ci = 0;
  bool subfield_run = false;
  while (ci < nreq_elts) {
    size_t runlen = 0;
    while (ci + runlen < nreq_elts && runlen < kMaxRunLen) {
      if (is_sub(ci + runlen) != subfield_run) break;
      ++runlen;
    }
    putsz(runlen);
    for (size_t i = ci; i < ci + runlen; ++i) {
      if (subfield_run && field == LFGPU_FIELD_GF2_128) {
        const u32 u = sub_coord[i];  // to_bytes_subfield: 2 bytes LE
        o.push_back((uint8_t)u);
        o.push_back((uint8_t)(u >> 8));
      } else {  // full-field run, or Fp128 where to_bytes_subfield == to_bytes_field
        pute(zk->req[i]);
      }
    }
    ci += runlen;
    subfield_run = !subfield_run;
  }
  putsz(zk->npath);
  o.insert(o.end(), zk->path.begin(), zk->path.begin() + 32 * zk->npath);
  zk->wire_valid = true;
  *nbytes = o.size();
  if (buf) {
    if (cap < o.size()) return lf_fail(zk->c, LFGPU_ERR_ARG, "zk_proof_write: buffer too small (%zu < %zu)", cap, o.size());
    memcpy(buf, o.data(), o.size());
  }
  return LFGPU_OK;
}

extern "C" int lfgpu_zk_timings(const lfgpu_zk_prover* zk, double ms[6]) {
  if (!zk || !ms) return LFGPU_ERR_ARG;
  if (zk->z256) return zk256_timings(zk->z256, ms);
  memcpy(ms, zk->ms, sizeof(zk->ms));
  return LFGPU_OK;
}

extern "C" int lfgpu_zk_prover_free(lfgpu_zk_prover* zk) {
  if (!zk) return LFGPU_ERR_ARG;
  delete zk;
  return LFGPU_OK;
}

// ------------------------------------------------------------------ ZkVerifier
// ZkVerifier::recv_commitment + verify (lib/zk/zk_verifier.h:68-94) over the wire bytes of ZkProof::write:
//   ZkProof::read                           lib/zk/zk_proof.h:107-112,218-345
//   ZkCommon::verifier_constraints, aux == nullptr (bind_quad -> Quad::bind_gh_all)   lib/zk/zk_common.h:49-136,441-450
//   LigeroVerifier::verify                  lib/ligero/ligero_verifier.h:42-270
//   MerkleCommitmentVerifier::verify        lib/merkle/merkle_commitment.h:85-99, merkle_tree.h:160-209
// Device work: bind_gh_all of every layer (the bulk: one pass over all corners of the circuit), the Reed-Solomon
// extension of the nwqrow rows of A and of the three y vectors, the gather at the opened columns.  Host: transcript
// replay, symbolic constraints, the nreq column hashes and the Merkle recomputation.
namespace {
struct ParsedProof {
  uint8_t root[32];
  std::vector<lfgpu_zk_prover::LayerPad> sc;
  std::vector<elt_t> y_ldt, y_dot, y_q0, y_q2, req;
  std::vector<uint8_t> nonces, path;
  size_t npath = 0;
};

struct Reader {
  const uint8_t* p;
  size_t left;
  bool have(size_t n) const { return left >= n; }
  const uint8_t* next(size_t n) {
    const uint8_t* r = p;
    p += n;
    left -= n;
    return r;
  }
  int field = LFGPU_FIELD_GF2_128;
  bool bad = false;  // an of_bytes_field failed (Fp128: value >= p)
  elt_t elt() {
    elt_t e;
    if (!elt_of_bytes(field, next(16), e)) bad = true;
    return e;
  }
  size_t size4() {
    const uint8_t* b = next(4);
    return (size_t)b[0] | (size_t)b[1] << 8 | (size_t)b[2] << 16 | (size_t)b[3] << 24;
  }
};

// ZkProof::read; false on underflow or inconsistent sizes (the reference returns false as well)
bool parse_proof(const lfgpu_circuit* C, const lfgpu_ligero_param& p, const GfHostCtx* g, const uint8_t* buf, size_t len, ParsedProof& pr) {
  const int field = C->info.field;
  const size_t sub_bytes = field == LFGPU_FIELD_GF2_128 ? 2 : 16;
  Reader rd{buf, len, field};
  if (!rd.have(32)) return false;
  memcpy(pr.root, rd.next(32), 32);
  pr.sc.assign(C->layers.size(), {});
  for (size_t ly = 0; ly < C->layers.size(); ++ly) {
    const size_t logw = C->layers[ly].logw;
    if (!rd.have((logw * 4 + 2) * 16)) return false;
    auto& P = pr.sc[ly];
    P.hp[0].resize(2 * logw);
    P.hp[1].resize(2 * logw);
    for (size_t wi = 0; wi < logw; ++wi)
      for (int k = 0; k < 2; ++k) {
        P.hp[0][2 * wi + k] = rd.elt();
        P.hp[1][2 * wi + k] = rd.elt();
      }
    P.wc[0] = rd.elt();
    P.wc[1] = rd.elt();
  }
  auto vec = [&](std::vector<elt_t>& v, size_t n) {
    if (!rd.have(n * 16)) return false;
    v.resize(n);
    for (auto& e : v) e = rd.elt();
    return true;
  };
  if (!vec(pr.y_ldt, p.block) || !vec(pr.y_dot, p.dblock) || !vec(pr.y_q0, p.r) || !vec(pr.y_q2, p.dblock - p.block)) return false;
  if (!rd.have(p.nreq * 32)) return false;
  pr.nonces.assign(rd.p, rd.p + p.nreq * 32);
  rd.next(p.nreq * 32);
  const size_t total = p.nreq * p.nrow;
  constexpr size_t kMaxRunLen = (size_t)1 << 25, kMaxNumDigests = (size_t)1 << 25;
  pr.req.assign(total, elt_t{0, 0});
  size_t ci = 0;
  bool subfield_run = false;
  while (ci < total) {
    if (!rd.have(4)) return false;
    const size_t runlen = rd.size4();
    if (runlen >= kMaxRunLen || ci + runlen > total) return false;
    if (subfield_run) {
      if (!rd.have(runlen * sub_bytes)) return false;
      for (size_t i = ci; i < ci + runlen; ++i) {  // of_bytes_subfield: of_scalar(u) = sum_i bit_i(u) beta_i
        if (field != LFGPU_FIELD_GF2_128) {  // Fp128: of_bytes_subfield == of_bytes_field
          pr.req[i] = rd.elt();
          continue;
        }
        const uint8_t* b = rd.next(2);  // (kSubFieldBytes = 2: the wire format is GF2_128<4>'s)
        pr.req[i] = gf_add(g->sub_tab[0][b[0]], g->sub_tab[1][b[1]]);  // of_scalar through the byte tables (GfHostCtx)
      }
    } else {
      if (!rd.have(runlen * 16)) return false;
      for (size_t i = ci; i < ci + runlen; ++i) pr.req[i] = rd.elt();
    }
    ci += runlen;
    subfield_run = !subfield_run;
  }
  if (!rd.have(4)) return false;
  const size_t sz = rd.size4();
  if (sz < p.nreq || sz >= kMaxNumDigests || sz > p.nreq * p.mc_pathlen || !rd.have(sz * 32)) return false;
  pr.npath = sz;
  pr.path.assign(rd.p, rd.p + sz * 32);
  rd.next(sz * 32);
  return !rd.bad;
}

}  // namespace

static void hash2(const uint8_t* a, const uint8_t* b, uint8_t out[32]) {  // Digest::hash2: SHA-256(left || right)
  Sha256 s;
  s.update(a, 32);
  s.update(b, 32);
  s.digest(out);
}

// MerkleTreeVerifier::verify_compressed_proof (merkle_tree.h:160-209)
bool lf_merkle_verify(size_t n, const uint8_t root[32], const uint8_t* path, size_t npath, const uint8_t* leaves, const size_t* pos, size_t np) {
  std::vector<uint8_t> layers(2 * n * 32, 0);
  std::vector<bool> defined(2 * n, false), tree(2 * n, false);
  for (size_t ip = 0; ip < np; ++ip) {
    if (pos[ip] >= n) return false;
    tree[pos[ip] + n] = true;
  }
  for (size_t i = n; i-- > 1;) tree[i] = tree[2 * i] || tree[2 * i + 1];
  size_t sz = 0;
  for (size_t i = n; i-- > 1;) {
    if (tree[i]) {
      size_t child = 2 * i;
      if (tree[child]) child = 2 * i + 1;
      if (!tree[child]) {
        if (sz >= npath) return false;
        memcpy(&layers[child * 32], path + 32 * sz++, 32);
        defined[child] = true;
      }
    }
  }
  if (sz != npath) return false;  // the whole proof must be consumed
  for (size_t ip = 0; ip < np; ++ip) {
    memcpy(&layers[(pos[ip] + n) * 32], leaves + 32 * ip, 32);
    defined[pos[ip] + n] = true;
  }
  for (size_t i = n; i-- > 1;)
    if (defined[2 * i] && defined[2 * i + 1]) {
      hash2(&layers[2 * i * 32], &layers[(2 * i + 1) * 32], &layers[i * 32]);
      defined[i] = true;
    }
  return defined[1] && memcmp(root, &layers[32], 32) == 0;
}

static int zk_verify_impl(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, const uint8_t* proof, size_t proof_len,
                          const void* h_pub, const lfgpu_transcript_ops* tso, bool committed, int* ok, const char** why_out);
extern "C" int lfgpu_zk_verify(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, const uint8_t* proof,
                               size_t proof_len, const void* h_pub, const lfgpu_transcript_ops* tso, int* ok, const char** why_out) {
  return zk_verify_impl(c, C, rateinv, nreq, block_enc, proof, proof_len, h_pub, tso, false, ok, why_out);
}
extern "C" int lfgpu_zk_verify_committed(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, const uint8_t* proof,
                                         size_t proof_len, const void* h_pub, const lfgpu_transcript_ops* tso, int* ok, const char** why_out) {
  return zk_verify_impl(c, C, rateinv, nreq, block_enc, proof, proof_len, h_pub, tso, true, ok, why_out);
}
static int zk_verify_impl(lfgpu_ctx* c, const lfgpu_circuit* C, size_t rateinv, size_t nreq, size_t block_enc, const uint8_t* proof, size_t proof_len,
                          const void* h_pub, const lfgpu_transcript_ops* tso, bool committed, int* ok, const char** why_out) {
  static const char* kWhy[] = {"ok", "proof does not parse", "merkle_check failed", "low_degree_check failed", "dot_check failed",
                               "wrong dot product", "quadratic_check failed"};
  if (!c || !C || C->c != c || !proof || !tso || !ok || (C->info.npub_in && !h_pub)) return LFGPU_ERR_ARG;
  *ok = 0;
  if (C->info.field == LFGPU_FIELD_P256) return zk256_verify(c, C, rateinv, nreq, block_enc, proof, proof_len, h_pub, tso, committed, ok, why_out);
  auto fail = [&](int w) {
    if (why_out) *why_out = kWhy[w];
    return LFGPU_OK;
  };
  const lfgpu_circuit_info& I = C->info;
  const size_t nl = C->layers.size(), npub = I.npub_in, n_witness = I.ninputs - npub;
  size_t pad_size = 0;
  for (const auto& l : C->layers) pad_size += layer_size(l.logw);
  lfgpu_ligero_param p{};
  const int field = I.field;
  LF_TRY(lfgpu_ligero_param_init(&p, field, 4, n_witness + pad_size, nl, rateinv, nreq, block_enc));
  const GfHostCtx* g = lf_gf_ctx(c, 4);
  if (!g) return LFGPU_ERR_ARG;
  static const bool verbose = getenv("LFGPU_VERBOSE") != nullptr;
  double tv[6] = {now_ms(), 0, 0, 0, 0, 0};
  ParsedProof pr;
  if (!parse_proof(C, p, g, proof, proof_len, pr)) return fail(1);
  tv[1] = now_ms();
  LF_HIP(c, hipSetDevice(c->device));
  const HostField F(c, field);
  const Ts ts{tso, tso->user, field};
  const elt_t* pub = (const elt_t*)h_pub;

  // recv_commitment (unless the caller has done it: ZkVerifier::recv_commitment and verify are separate calls, and the mdoc
  // verifier draws its MAC key between them, mdoc_zk.cc:676-681), initialize_sumcheck_fiat_shamir
  if (!committed) ts.write_bytes(pr.root, 32);
  ts.write_bytes(I.id, 32);
  for (size_t i = 0; i < npub; ++i) ts.write_elt(pub[i]);
  ts.write_elt(elt_t{0, 0});
  ts.write_bytes(C->zeros->data(), I.nterms);

  // verifier_constraints with aux == nullptr: the bound quad of every layer comes from bind_gh_all
  ConstraintSet cs;
  LF_TRY(build_constraints(c, C, F, ts, pr.sc, nullptr, pub, cs));
  const size_t nconstraints = cs.n;
  std::vector<size_t> lqc(3 * nl);
  {
    size_t pi = n_witness;
    for (size_t ly = 0; ly < nl; ++ly) {  // setup_lqc (zk_common.h:149-160)
      const size_t cp = pi + 4 * C->layers[ly].logw;
      lqc[3 * ly] = cp;
      lqc[3 * ly + 1] = cp + 1;
      lqc[3 * ly + 2] = cp + 2;
      pi += layer_size(C->layers[ly].logw);
    }
  }
  tv[2] = now_ms();
  // LigeroVerifier::verify: replay the challenges
  uint8_t hash_of_A[32] = {0xde, 0xad, 0xbe, 0xef};
  ts.write_bytes(hash_of_A, 32);
  std::vector<elt_t> u_ldt(p.nwqrow), alphal(nconstraints), alphaq(3 * p.nq), u_quad(p.nqtriples ? p.nqtriples : 1);
  for (auto& e : u_ldt) e = ts.elt();
  for (auto& e : alphal) e = ts.elt();
  for (auto& e : alphaq) e = ts.elt();
  for (size_t i = 0; i < p.nqtriples; ++i) u_quad[i] = ts.elt();
  ts.write_array(pr.y_ldt.data(), pr.y_ldt.size());
  ts.write_array(pr.y_dot.data(), pr.y_dot.size());
  ts.write_array(pr.y_q0.data(), pr.y_q0.size());
  ts.write_array(pr.y_q2.data(), pr.y_q2.size());
  std::vector<size_t> idx(p.nreq);
  ts.choose(p.block_ext, p.nreq, idx.data());
  auto req_at = [&](size_t i, size_t j) -> elt_t { return pr.req[i * p.nreq + j]; };

  {  // merkle_check: leaf r = SHA-256(nonce_r || column r of the opening)
    std::vector<uint8_t> leaves(p.nreq * 32);
    for (size_t r = 0; r < p.nreq; ++r) {
      Sha256 s;
      s.update(&pr.nonces[32 * r], 32);
      for (size_t i = 0; i < p.nrow; ++i) {
        uint8_t eb[16];
        elt_to_bytes(field, req_at(i, r), eb);
        s.update(eb, 16);
      }
      s.digest(&leaves[32 * r]);
    }
    if (!lf_merkle_verify(p.block_ext, pr.root, pr.path.data(), pr.npath, leaves.data(), idx.data(), p.nreq)) return fail(2);
  }

  tv[3] = now_ms();
  // device: rows [0, nwqrow) = [0^r | A_i] extended block -> block_enc, rows nwqrow.. = y_ldt, y_dot, y_quad
  std::vector<uint64_t> a_idx;
  std::vector<elt_t> a_val;
  inner_product_sparse(F, p, cs, alphal, lqc, alphaq, a_idx, a_val);
  const size_t nrows_dev = p.nwqrow + 3, ld = p.block_enc;
  void* dT = nullptr;
  // scratch4: the RS extension below runs its FFT passes through `scratch` / `scratch2` (fft.hip, lch_bs.hip, rs.hip)
  LF_TRY(lf_scratch4(c, (nrows_dev * ld + (size_t)nrows_dev * p.nreq) * 16 + 256, &dT));
  elt_t* d_T = (elt_t*)dT;
  elt_t* d_req = d_T + nrows_dev * ld;
  {  // only the first dblock columns of a row are inputs: clear them on the device, then strided copies
    LF_HIP(c, hipMemset2DAsync(d_T, ld * 16, 0, p.dblock * 16, nrows_dev, c->stream));
    {  // inner_product_vector + layout_Aext on the device
      const elt_t ad = alphal[cs.n - 1];
      const uint64_t sc[2] = {ad.lo, ad.hi};
      LF_TRY(lfgpu_ligero_inner_product_rows(c, field, p.w, p.r, ld, p.nwqrow, cs.d_eq + npub, n_witness, sc, a_idx.data(), a_val.data(), a_idx.size(), d_T));
    }
    LF_HIP(c, hipMemcpyAsync(d_T + (p.nwqrow + 0) * ld, pr.y_ldt.data(), p.block * 16, hipMemcpyHostToDevice, c->stream));
    LF_HIP(c, hipMemcpyAsync(d_T + (p.nwqrow + 1) * ld, pr.y_dot.data(), p.dblock * 16, hipMemcpyHostToDevice, c->stream));
    elt_t* yq = d_T + (p.nwqrow + 2) * ld;  // y_quad = y_quad_0 | 0^w | y_quad_2
    LF_HIP(c, hipMemcpyAsync(yq, pr.y_q0.data(), p.r * 16, hipMemcpyHostToDevice, c->stream));
    LF_HIP(c, hipMemcpyAsync(yq + p.block, pr.y_q2.data(), (p.dblock - p.block) * 16, hipMemcpyHostToDevice, c->stream));
    LF_HIP(c, hipStreamSynchronize(c->stream));
  }
  LF_TRY(lf_rs_rows(c, field, 4, p.nwqrow + 1, p.block, p.block_enc, d_T, ld));                       // A rows and y_ldt
  LF_TRY(lf_rs_rows(c, field, 4, 2, p.dblock, p.block_enc, d_T + (p.nwqrow + 1) * ld, ld));            // y_dot, y_quad
  LF_TRY(lfgpu_gather_columns(c, nrows_dev, ld, p.dblock, d_T, idx.data(), p.nreq, d_req));
  std::vector<elt_t> ext(nrows_dev * p.nreq);
  LF_HIP(c, hipMemcpyAsync(ext.data(), d_req, ext.size() * 16, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  auto ext_at = [&](size_t row, size_t j) -> elt_t { return ext[row * p.nreq + j]; };
  tv[4] = now_ms();

  for (size_t j = 0; j < p.nreq; ++j) {  // low_degree_check
    elt_t yc = req_at(p.ildt, j);
    for (size_t i = 0; i < p.nwqrow; ++i) yc = F.add(yc, F.mul(u_ldt[i], req_at(i + p.iw, j)));
    if (!elt_eq(yc, ext_at(p.nwqrow, j))) return fail(3);
  }
  for (size_t j = 0; j < p.nreq; ++j) {  // dot_check
    elt_t yc = req_at(p.idot, j);
    for (size_t i = 0; i < p.nwqrow; ++i) yc = F.add(yc, F.mul(ext_at(i, j), req_at(i + p.iw, j)));
    if (!elt_eq(yc, ext_at(p.nwqrow + 1, j))) return fail(4);
  }
  {  // the putative value of the inner product
    elt_t want{0, 0}, got{0, 0};
    for (size_t k = 0; k < nconstraints; ++k) want = F.add(want, F.mul(cs.b[k], alphal[k]));
    for (size_t j = 0; j < p.w; ++j) got = F.add(got, pr.y_dot[p.r + j]);
    if (!elt_eq(want, got)) return fail(5);
  }
  {  // quadratic_check
    const size_t iqx = p.iq, iqy = iqx + p.nqtriples, iqz = iqy + p.nqtriples;
    for (size_t j = 0; j < p.nreq; ++j) {
      elt_t yc = req_at(p.iquad, j);
      for (size_t i = 0; i < p.nqtriples; ++i) {
        const elt_t tmp = F.sub(req_at(iqz + i, j), F.mul(req_at(iqx + i, j), req_at(iqy + i, j)));  // z - x*y
        yc = F.add(yc, F.mul(u_quad[i], tmp));
      }
      if (!elt_eq(yc, ext_at(p.nwqrow + 2, j))) return fail(6);
    }
  }
  if (verbose)
    fprintf(stderr, "lfgpu zk_verify: parse %.2f ms | FS init + constraints (bind_gh_all) %.2f | challenges + merkle %.2f | A + RS extension %.2f | checks %.2f\n",
            tv[1] - tv[0], tv[2] - tv[1], tv[3] - tv[2], tv[4] - tv[3], now_ms() - tv[4]);
  *ok = 1;
  return fail(0);
}
