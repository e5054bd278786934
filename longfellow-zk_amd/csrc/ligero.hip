// ligero.hip -- LigeroProver::commit / prove on a device-resident tableau.
//
// Reference: LigeroParam::layout (lib/ligero/ligero_param.h:185-295),
// LigeroProver::commit / layout_* / low_degree_proof / dot_proof / quadratic_proof /
// compute_req (lib/ligero/ligero_prover.h:58-79,171-351), MerkleCommitment
// (lib/merkle/merkle_commitment.h:50-73).
//
// Host side: all RandomEngine draws happen on the host in the reference's order (they do
// not depend on computed data), the un-encoded rows are assembled in one host buffer and
// uploaded once; the GPU then RS-encodes every row (K3/K4) and commits the columns
// (K5/K6).  The tableau stays in HBM for the prove-side row combinations (K12).
#include <algorithm>
#include <chrono>
#include <new>

#include "ctx.h"

struct lfgpu_ligero_prover {
  lfgpu_ctx* c;
  int field, k;
  lfgpu_ligero_param p;
  elt_t* d_T;             // [nrow][block_enc]
  uint8_t* d_layers;      // [2*block_ext][32]
  size_t T_bytes = 0, L_bytes = 0;
  std::vector<uint8_t> nonces;  // block_ext * 32 (host copy for open)
  // rows [row_lo, row_hi) of the tableau live in d_T (a slab of a multi-GPU commit; [0, nrow) on one GPU).  The prove
  // functions then return this slab's PARTIAL sums (rows it does not hold contribute zero); the caller folds the ranks.
  size_t row_lo = 0, row_hi = 0;
  bool owns = true;  // d_T / d_layers are freed (stashed) by lfgpu_ligero_free
  // a prover made by lfgpu_ligero_commit_sharded: the prove entry points fold the ranks' partial results through these hooks
  bool sharded = false;
  lfgpu_comm_ops comm{};
  std::vector<std::pair<size_t, size_t>> spans;  // row slab of every rank
  bool has(size_t i) const { return i >= row_lo && i < row_hi; }
  elt_t* row(size_t i) const { return d_T + (i - row_lo) * p.block_enc; }
};

static size_t ceildiv(size_t a, size_t b) { return (a + b - 1) / b; }
static size_t merkle_tree_len(size_t n) {  // merkle_tree.h:62-70
  size_t r = 1;
  size_t pos = n - 1;
  for (pos += n; pos > 1; pos >>= 1) ++r;
  return r;
}

// LigeroParam::layout (lib/ligero/ligero_param.h:185-295): fills *p for block_enc = e and returns the proof-size
// estimate, or SIZE_MAX where the reference rejects the layout.
static size_t ligero_layout(lfgpu_ligero_param* p, int field, int k, size_t e) {
  const size_t max_lg_size = 28, max_size = (size_t)1 << max_lg_size;
  const size_t field_bytes = field == LFGPU_FIELD_P256 ? 32 : 16;  // Field::kBytes / kSubFieldBytes
  const size_t subfield_bytes = field == LFGPU_FIELD_GF2_128 ? (((size_t)1 << k) / 8) : field_bytes;
  const size_t nw = p->nw, nq = p->nq, rateinv = p->rateinv, nreq = p->nreq;
  p->r = nreq;
  p->block_enc = e;
  size_t subfield_bits = 8 * subfield_bytes;
  if (subfield_bits <= max_lg_size && e >= ((size_t)1 << subfield_bits)) return SIZE_MAX;
  if (e > max_size || rateinv > max_size || (e + 1) < (2 + rateinv)) return SIZE_MAX;
  p->block = (e + 1) / (2 + rateinv);
  if (p->block < p->r) return SIZE_MAX;
  p->w = p->block - p->r;
  if (p->w < p->r) return SIZE_MAX;
  p->dblock = 2 * p->block - 1;
  if (e < p->dblock) return SIZE_MAX;
  p->block_ext = e - p->dblock;
  p->nwrow = ceildiv(nw, p->w);
  p->nqtriples = ceildiv(nq, p->w);
  p->nwqrow = p->nwrow + 3 * p->nqtriples;
  p->nrow = p->nwqrow + 3;
  if (p->nrow >= max_size / e) return SIZE_MAX;
  if (p->block_ext == 0) return SIZE_MAX;  // merkle_commitment_len(0) is undefined; a commitment needs leaves
  p->mc_pathlen = merkle_tree_len(p->block_ext);
  uint64_t sz = 32;                                                     // commitment
  sz += (uint64_t)p->mc_pathlen / 2 * (uint64_t)nreq * 32;             // Merkle openings (approximation)
  sz += (uint64_t)p->block * field_bytes;                              // y_ldt
  sz += (uint64_t)p->dblock * field_bytes;                             // y_dot
  sz += (uint64_t)(p->dblock - p->w) * field_bytes;                    // y_quad
  sz += (uint64_t)nreq * 32;                                           // nonces
  sz += (uint64_t)p->nrow * (uint64_t)nreq * (uint64_t)subfield_bytes; // req
  return (size_t)sz;
}

// block_enc != 0: LigeroParam(nw, nq, rateinv, nreq, block_enc) (ligero_param.h:172-178);
// block_enc == 0: the deprecated ctor that searches block_enc over powers of two for the smallest proof
// (ligero_param.h:152-169; what ZkProof(c, rate, req) and the benchmarks use).
extern "C" int lfgpu_ligero_param_init(lfgpu_ligero_param* p, int field, int k, size_t nw, size_t nq, size_t rateinv,
                                       size_t nreq, size_t block_enc) {
  if (!p) return LFGPU_ERR_ARG;
  memset(p, 0, sizeof(*p));
  p->nw = nw;
  p->nq = nq;
  p->rateinv = rateinv;
  p->nreq = nreq;
  if (block_enc == 0) {
    size_t best = SIZE_MAX, best_e = 1;
    for (size_t e = 1; e <= ((size_t)1 << 28); e *= 2) {
      size_t sz = ligero_layout(p, field, k, e);
      if (sz < best) {
        best = sz;
        best_e = e;
      }
    }
    block_enc = best_e;
  }
  if (ligero_layout(p, field, k, block_enc) == SIZE_MAX) return LFGPU_ERR_ARG;
  if (!(p->block_enc > p->block)) return LFGPU_ERR_ARG;  // sanity(): block_enc > block
  p->ildt = 0;
  p->idot = 1;
  p->iquad = 2;
  p->iw = 3;
  p->iq = p->iw + p->nwrow;
  return LFGPU_OK;
}

// ---- field sampling on the host (RandomEngine::elt / subfield_elt, lib/random/random.h:38-48)
struct Sampler {
  int field, k;
  const GfHostCtx* g;
  lfgpu_rng_fn rng;
  void* user;
  bool exact = false;  // lfgpu_set_rng_exact_calls: one call per element, as the reference draws
  // n consecutive full-field elements.  GF2_128::sample consumes exactly 16 bytes per element
  // (gf2_128.h:182-190), so n draws of 16 bytes equal one draw of 16n bytes for every byte-stream
  // RandomEngine of the reference (LCG test engines, Transcript/FSPRF, SecureRandomEngine); Fp128 uses
  // rejection sampling and is drawn element by element.
  void elts(elt_t* out, size_t n) {
    if (field == LFGPU_FIELD_GF2_128 && !exact) {
      rng(user, reinterpret_cast<uint8_t*>(out), 16 * n);  // little-endian host: bytes are the Elt image
    } else {
      for (size_t i = 0; i < n; ++i) out[i] = elt();
    }
  }
  elt_t elt() {
    if (field == LFGPU_FIELD_GF2_128) {  // GF2_128::sample gf2_128.h:182-190: 16 bytes LE
      uint8_t b[16];
      rng(user, b, 16);
      elt_t e;
      memcpy(&e.lo, b, 8);
      memcpy(&e.hi, b + 8, 8);
      return e;
    }
    // FpGeneric::sample fp_generic.h:360-371: 16 bytes, reject >= p, to Montgomery
    for (;;) {
      uint8_t b[16];
      rng(user, b, 16);
      elt_t e;
      memcpy(&e.lo, b, 8);
      memcpy(&e.hi, b + 8, 8);
      if (e.hi < FP_P_HI || (e.hi == FP_P_HI && e.lo < FP_P_LO)) {
        // to_montgomery = mul by R^2 (fp_generic.h:278-282); C++11 initialises a function-local static once, thread-safely
        static const elt_t rsq = [] {
          elt_t r{1, 0};
          for (int i = 0; i < 256; ++i) r = fp_add(r, r);
          return r;
        }();
        return fp_mul(e, rsq);
      }
    }
  }
  // n consecutive subfield elements: kSubFieldBytes each, so for a byte-stream engine one draw of n * kSubFieldBytes bytes;
  // of_scalar through the byte tables (the mdoc hash circuit draws 34 000 of them per commit: 2.0 -> 0.2 ms of host time)
  void subfield_elts(elt_t* out, size_t n) {
    if (field != LFGPU_FIELD_GF2_128 || exact) {
      for (size_t i = 0; i < n; ++i) out[i] = subfield_elt();
      return;
    }
    const size_t nb = g->sub_bits / 8;
    uint8_t buf[1024];
    for (size_t done = 0; done < n;) {
      const size_t m = std::min(n - done, sizeof(buf) / nb);
      rng(user, buf, m * nb);
      for (size_t i = 0; i < m; ++i) {
        elt_t t = g->sub_tab[0][buf[i * nb]];
        for (size_t b = 1; b < nb; ++b) t = gf_add(t, g->sub_tab[b][buf[i * nb + b]]);
        out[done + i] = t;
      }
      done += m;
    }
  }
  elt_t subfield_elt() {
    if (field != LFGPU_FIELD_GF2_128) return elt();  // FpGeneric::sample_subfield = sample
    // GF2_128::sample_subfield gf2_128.h:192-214: kSubFieldBytes LE -> of_scalar
    uint8_t b[8] = {0};
    size_t nb = g->sub_bits / 8;
    rng(user, b, nb);
    u64 u = 0;
    for (size_t i = nb; i-- > 0;) u = (u << 8) | b[i];
    elt_t t{0, 0};
    for (unsigned bit = 0; bit < g->sub_bits; ++bit, u >>= 1)
      if (u & 1) t = gf_add(t, g->beta[bit]);
    return t;
  }
};

static inline elt_t h_add(int field, elt_t a, elt_t b) { return field == LFGPU_FIELD_GF2_128 ? gf_add(a, b) : fp_add(a, b); }
static inline elt_t h_sub(int field, elt_t a, elt_t b) { return field == LFGPU_FIELD_GF2_128 ? gf_add(a, b) : fp_sub(a, b); }
static inline elt_t h_mul(int field, elt_t a, elt_t b) { return field == LFGPU_FIELD_GF2_128 ? gf_mul(a, b) : fp_mul(a, b); }

int lf_rs_rows(lfgpu_ctx* c, int field, int k, size_t nrow, size_t n, size_t m, elt_t* d, size_t ld) {
  if (nrow == 0) return LFGPU_OK;
  if (field == LFGPU_FIELD_GF2_128) return lfgpu_gf2128_rs_encode_rows(c, k, nrow, n, m, d, ld);
  // Fp128: the 2^32-order root of lib/algebra/fp_p128.h:48-56
  static const char* kOmega = "164956748514267535023998284330560247862";
  unsigned __int128 v = 0;
  for (const char* s = kOmega; *s; ++s) v = v * 10 + (unsigned)(*s - '0');
  elt_t raw{(u64)v, (u64)(v >> 64)};
  elt_t rsq{1, 0};
  for (int i = 0; i < 256; ++i) rsq = fp_add(rsq, rsq);
  elt_t w = fp_mul(raw, rsq);
  uint64_t om[2] = {w.lo, w.hi};
  return lfgpu_fp128_rs_encode_rows(c, nrow, n, m, om, (uint64_t)1 << 32, d, ld);
}

// The host half of LigeroProver::commit (ligero_prover.h:171-270 + merkle_commitment.h:52-54): every RandomEngine draw in
// the reference's order; the un-encoded image (first dblock columns) of rows [row_lo, row_hi) goes to H
// ([(row_hi - row_lo)][dblock]); rows outside the slab are drawn into a spare row and dropped, so that ranks which replay
// the same byte stream hold consistent slabs of ONE tableau.  nonces (block_ext * 32 bytes) may be null: the draw still
// happens.  Returns LFGPU_OK / LFGPU_ERR_ARG / LFGPU_ERR_ASSERT with a message in err (256 bytes).
static int ligero_layout_host(int field, int k, const GfHostCtx* g, const lfgpu_ligero_param& p, const elt_t* W, size_t subfield_boundary,
                              const size_t* h_lqc, lfgpu_rng_fn rng, void* user, size_t row_lo, size_t row_hi, elt_t* H, uint8_t* nonces,
                              char* err, bool exact = false) {
  Sampler S{field, k, g, rng, user, exact};
  const elt_t zero{0, 0};
  const size_t hw = p.dblock;
  std::vector<elt_t> spare(hw);
  auto row = [&](size_t i) -> elt_t* {
    if (i >= row_lo && i < row_hi) return H + (i - row_lo) * hw;
    return spare.data();
  };
  auto own = [&](size_t i) { return i >= row_lo && i < row_hi; };
  for (size_t i = row_lo; i < row_hi; ++i) std::fill(row(i), row(i) + hw, zero);
  // layout_blinding_rows (ligero_prover.h:171-205)
  S.elts(row(p.ildt), p.block);
  {
    elt_t* d = row(p.idot);
    S.elts(d, p.dblock);
    elt_t sum = zero;
    for (size_t j = 0; j < p.w; ++j) sum = h_add(field, sum, d[p.r + j]);
    d[p.r] = h_sub(field, d[p.r], sum);
  }
  {
    elt_t* q = row(p.iquad);
    S.elts(q, p.dblock);
    for (size_t j = 0; j < p.w; ++j) q[p.r + j] = zero;
  }
  // layout_witness_rows (:207-231)
  for (size_t i = 0; i < p.nwrow; ++i) {
    elt_t* t = row(i + p.iw);
    const bool subfield_only = ((i + 1) * p.w <= subfield_boundary);
    if (subfield_only) {
      S.subfield_elts(t, p.r);
    } else {
      S.elts(t, p.r);
    }
    if (!own(i + p.iw)) continue;
    const size_t max_col = std::min(p.w, p.nw - i * p.w);
    for (size_t j = 0; j < max_col; ++j) t[p.r + j] = W[i * p.w + j];
  }
  // layout_quadratic_rows (:233-270)
  const size_t iqx = p.iq, iqy = iqx + p.nqtriples, iqz = iqy + p.nqtriples;
  for (size_t i = 0; i < p.nqtriples; ++i) {
    S.elts(row(iqx + i), p.r);
    S.elts(row(iqy + i), p.r);
    S.elts(row(iqz + i), p.r);
    for (size_t j = 0; j < p.w && j + i * p.w < p.nq; ++j) {
      const size_t* l = &h_lqc[3 * (j + i * p.w)];
      if (l[0] >= p.nw || l[1] >= p.nw || l[2] >= p.nw) {
        snprintf(err, 256, "ligero_commit: lqc index >= nw");
        return LFGPU_ERR_ARG;
      }
      const elt_t prod = h_mul(field, W[l[0]], W[l[1]]);
      if (!(prod.lo == W[l[2]].lo && prod.hi == W[l[2]].hi)) {
        snprintf(err, 256, "ligero_commit: invalid quadratic constraints (ligero_prover.h:259-260)");
        return LFGPU_ERR_ASSERT;
      }
      if (own(iqx + i)) row(iqx + i)[j + p.r] = W[l[0]];
      if (own(iqy + i)) row(iqy + i)[j + p.r] = W[l[1]];
      if (own(iqz + i)) row(iqz + i)[j + p.r] = W[l[2]];
    }
  }
  // MerkleCommitment::commit draws one 32-byte nonce per leaf, after the layout (merkle_commitment.h:52-54): one draw of
  // 32 * block_ext bytes is the same stream for every byte-stream engine
  if (nonces && exact) {
    for (size_t j = 0; j < p.block_ext; ++j) rng(user, nonces + 32 * j, 32);
  } else if (nonces) {
    rng(user, nonces, 32 * p.block_ext);
  } else {
    std::vector<uint8_t> drop(32 * p.block_ext);
    if (exact) for (size_t j = 0; j < p.block_ext; ++j) rng(user, drop.data() + 32 * j, 32);
    else rng(user, drop.data(), drop.size());
  }
  return LFGPU_OK;
}

static bool ligero_args_ok(int field, int k, const lfgpu_ligero_param* pp, const void* h_W, const size_t* h_lqc, lfgpu_rng_fn rng) {
  if (!pp || !rng || (pp->nw && !h_W) || (pp->nq && !h_lqc)) return false;
  if (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128) return false;
  if (field == LFGPU_FIELD_GF2_128 && k != 4 && k != 5) return false;
  return true;
}

// Multi-GPU commit, host half (no device needed): see include/lfgpu.h
extern "C" int lfgpu_ligero_layout_rows(int field, int k, const lfgpu_ligero_param* pp, const void* h_W, size_t subfield_boundary,
                                        const size_t* h_lqc, lfgpu_rng_fn rng, void* user, size_t row_lo, size_t row_hi, void* h_rows,
                                        uint8_t* h_nonces) {
  if (!ligero_args_ok(field, k, pp, h_W, h_lqc, rng) || row_lo > row_hi || row_hi > pp->nrow || (row_hi > row_lo && !h_rows)) return LFGPU_ERR_ARG;
  GfHostCtx g;
  if (field == LFGPU_FIELD_GF2_128 && !lf_gf_ctx_build(&g, k)) return LFGPU_ERR_ARG;
  char err[256];
  return ligero_layout_host(field, k, &g, *pp, (const elt_t*)h_W, subfield_boundary, h_lqc, rng, user, row_lo, row_hi, (elt_t*)h_rows, h_nonces, err);
}

// rows [row_lo, row_hi) of the tableau: upload the un-encoded image and RS-extend every row to block_enc
// (rows IDOT / IQUAD carry dblock values, every other row block: ligero_prover.h:175,184,203,210,237)
static int ligero_encode_slab(lfgpu_ctx* c, int field, int k, const lfgpu_ligero_param& p, size_t row_lo, size_t row_hi, const elt_t* H, elt_t* d_slab) {
  const size_t nr = row_hi - row_lo, ld = p.block_enc, hw = p.dblock;
  if (nr == 0) return LFGPU_OK;
  LF_HIP(c, hipMemcpy2DAsync(d_slab, ld * 16, H, hw * 16, hw * 16, nr, hipMemcpyHostToDevice, c->stream));
  // the rows with dblock values form one contiguous global range [idot, iquad + 1); clip it to the slab
  const size_t g_lo = std::min(p.idot, p.iquad), g_hi = std::max(p.idot, p.iquad) + 1;
  const bool contiguous = g_hi - g_lo == 2;
  const size_t lo2 = std::min(std::max(g_lo, row_lo), row_hi) - row_lo, hi2 = std::min(std::max(g_hi, row_lo), row_hi) - row_lo;
  int rc = field == LFGPU_FIELD_GF2_128 && contiguous
               ? lf_gf_rs_rows_mixed(c, k, nr, p.block, p.dblock, lo2, hi2, p.block_enc, d_slab, ld)
               : LFGPU_ERR_UNSUPPORTED;
  if (rc == LFGPU_ERR_UNSUPPORTED) {  // rows larger than the LDS-resident kernel / Fp128: group by group
    if (!contiguous) return lf_fail(c, LFGPU_ERR_ARG, "ligero: IDOT and IQUAD rows must be adjacent");
    LF_TRY(lf_rs_rows(c, field, k, lo2, p.block, p.block_enc, d_slab, ld));
    LF_TRY(lf_rs_rows(c, field, k, hi2 - lo2, p.dblock, p.block_enc, d_slab + lo2 * ld, ld));
    LF_TRY(lf_rs_rows(c, field, k, nr - hi2, p.block, p.block_enc, d_slab + hi2 * ld, ld));
    rc = LFGPU_OK;
  }
  return rc;
}

extern "C" int lfgpu_ligero_encode_rows(lfgpu_ctx* c, int field, int k, const lfgpu_ligero_param* pp, size_t row_lo, size_t row_hi,
                                        const void* h_rows, void* d_slab) {
  if (!c || !pp || row_lo > row_hi || row_hi > pp->nrow || (row_hi > row_lo && (!h_rows || !d_slab)))
    return lf_fail(c, LFGPU_ERR_ARG, "ligero_encode_rows: bad argument");
  if (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128) return lf_fail(c, LFGPU_ERR_ARG, "ligero_encode_rows: field");
  if (field == LFGPU_FIELD_GF2_128 && !lf_gf_ctx(c, k)) return lf_fail(c, LFGPU_ERR_ARG, "ligero_encode_rows: subfield_log_bits must be 4 or 5");
  LF_HIP(c, hipSetDevice(c->device));
  LF_TRY(ligero_encode_slab(c, field, k, *pp, row_lo, row_hi, (const elt_t*)h_rows, (elt_t*)d_slab));
  LF_HIP(c, hipStreamSynchronize(c->stream));  // h_rows is the caller's: the upload must be over before we return
  return LFGPU_OK;
}

extern "C" int lfgpu_ligero_commit(lfgpu_ctx* c, int field, int k, const lfgpu_ligero_param* pp, const void* h_W,
                                   size_t subfield_boundary, const size_t* h_lqc, lfgpu_rng_fn rng, void* user,
                                   uint8_t root_out[32], lfgpu_ligero_prover** out) {
  if (!c || !root_out || !out || !ligero_args_ok(field, k, pp, h_W, h_lqc, rng))
    return lf_fail(c, LFGPU_ERR_ARG, "ligero_commit: bad argument (null pointer, field or subfield_log_bits)");
  const GfHostCtx* g = nullptr;
  if (field == LFGPU_FIELD_GF2_128) g = lf_gf_ctx(c, k);
  const lfgpu_ligero_param& p = *pp;
  const size_t ld = p.block_enc;
  LF_HIP(c, hipSetDevice(c->device));

  lfgpu_ligero_prover* pr = new lfgpu_ligero_prover();
  pr->c = c;
  pr->field = field;
  pr->k = k;
  pr->p = p;
  pr->d_T = nullptr;
  pr->d_layers = nullptr;
  pr->nonces.resize(p.block_ext * 32);
  auto fail = [&](int rc) {
    lfgpu_ligero_free(pr);
    return rc;
  };
  // host image of the un-encoded rows (only the first dblock columns are ever non-trivial) + the nonces: the single-GPU
  // commit is the slab [0, nrow) of the sharded one
  static const bool verbose = getenv("LFGPU_VERBOSE") != nullptr;
  auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tv0 = verbose ? clk() : 0;
  std::vector<elt_t> H(p.nrow * p.dblock);
  LF_SCRUB_ON_EXIT(H);
  {
    char err[256] = {0};
    const int rc = ligero_layout_host(field, k, g, p, (const elt_t*)h_W, subfield_boundary, h_lqc, rng, user, 0, p.nrow, H.data(),
                                      pr->nonces.data(), err, c->rng_exact != 0);
    if (rc) return fail(lf_fail(c, rc, "%s", err));
  }
  const double tv1 = verbose ? clk() : 0;
  const size_t tb = p.nrow * ld * 16, lb = 2 * p.block_ext * 32;
  // buffers of an earlier prover with the same shape where there is one (the context's pool)
  if (lf_pool_get(c, tb, (void**)&pr->d_T) != LFGPU_OK) return fail(lf_fail(c, LFGPU_ERR_NOMEM, "ligero_commit: tableau alloc"));
  if (lf_pool_get(c, lb, (void**)&pr->d_layers) != LFGPU_OK) return fail(lf_fail(c, LFGPU_ERR_NOMEM, "ligero_commit: layers alloc"));
  pr->T_bytes = tb;
  pr->L_bytes = lb;
  pr->row_lo = 0;
  pr->row_hi = p.nrow;
  void* d_non = nullptr;
  int rc = lf_scratch3(c, p.block_ext * 32, &d_non);
  if (rc) return fail(rc);
  if (hipMemcpyAsync(d_non, pr->nonces.data(), p.block_ext * 32, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    return fail(lf_fail(c, LFGPU_ERR_HIP, "ligero_commit: upload failed"));
  if ((rc = ligero_encode_slab(c, field, k, p, 0, p.nrow, H.data(), pr->d_T))) return fail(rc);
  double tv2 = 0;
  if (verbose) {
    (void)hipStreamSynchronize(c->stream);
    tv2 = clk();
  }
  if ((rc = lfgpu_column_commit(c, field, p.nrow, ld, p.dblock, p.block_ext, pr->d_T, d_non, pr->d_layers, root_out)))
    return fail(rc);
  if (verbose)
    fprintf(stderr, "lfgpu ligero_commit: %zu rows x %zu (block %zu, dblock %zu): host layout + draws %.2f ms | upload + RS encode %.2f | column hash + tree %.2f\n",
            p.nrow, ld, p.block, p.dblock, tv1 - tv0, tv2 - tv1, clk() - tv2);
  *out = pr;
  return LFGPU_OK;
}

// ------------------------------------------------------------------ rows sharded over the GPUs of a node
static void shard_split(size_t n, int rank, int world, size_t* start, size_t* count) {  // contiguous, sizes differ by at most 1
  const size_t base = n / (size_t)world, rem = n % (size_t)world, r = (size_t)rank;
  *start = r * base + std::min(r, rem);
  *count = base + (r < rem ? 1 : 0);
}
extern "C" int lfgpu_ligero_row_shard(const lfgpu_ligero_param* p, int rank, int world, size_t* row_lo, size_t* row_hi) {
  if (!p || !row_lo || !row_hi || world < 1 || rank < 0 || rank >= world) return LFGPU_ERR_ARG;
  size_t lo, cnt;
  shard_split(p->nrow, rank, world, &lo, &cnt);
  *row_lo = std::min(lo, p->iq);
  *row_hi = rank == world - 1 ? p->nrow : std::min(lo + cnt, p->iq);
  return LFGPU_OK;
}
namespace {
struct RecordRng {  // rank 0: the caller's engine, every byte kept
  lfgpu_rng_fn rng;
  void* user;
  std::vector<uint8_t>* rec;
};
void record_fn(void* u, uint8_t* buf, size_t n) {
  RecordRng* r = (RecordRng*)u;
  r->rng(r->user, buf, n);
  r->rec->insert(r->rec->end(), buf, buf + n);
}
struct ReplayRng {  // every rank: the broadcast stream (running dry yields zeros and sets `dry`)
  const std::vector<uint8_t>* data;
  size_t pos = 0;
  bool dry = false;
};
void replay_fn(void* u, uint8_t* buf, size_t n) {
  ReplayRng* r = (ReplayRng*)u;
  if (r->pos + n > r->data->size()) {
    r->dry = true;
    memset(buf, 0, n);
    return;
  }
  memcpy(buf, r->data->data() + r->pos, n);
  r->pos += n;
}
bool comm_ok(const lfgpu_comm_ops* cm) { return cm && cm->world >= 1 && cm->rank >= 0 && cm->rank < cm->world && cm->all_gather && cm->all_to_all && cm->broadcast; }
}  // namespace
// one rank's byte string to every rank (length first); host buffers
int lf_comm_bcast_blob(const lfgpu_comm_ops* cm, std::vector<uint8_t>& blob) {
  if (cm->world == 1) return LFGPU_OK;
  uint64_t n = cm->rank == 0 ? blob.size() : 0;
  if (cm->broadcast(cm->user, &n, 8, 0, 0, nullptr)) return LFGPU_ERR_HIP;
  if (cm->rank != 0) blob.assign(n, 0);
  if (n && cm->broadcast(cm->user, blob.data(), n, 0, 0, nullptr)) return LFGPU_ERR_HIP;
  return LFGPU_OK;
}
void lf_replay_rng(const std::vector<uint8_t>* data, lfgpu_rng_fn* fn, void** user, void* storage /* >= sizeof(ReplayRng) */) {
  ReplayRng* r = new (storage) ReplayRng{data, 0, false};
  *fn = replay_fn;
  *user = r;
}
void lf_record_rng(lfgpu_rng_fn rng, void* rng_user, std::vector<uint8_t>* rec, lfgpu_rng_fn* fn, void** user, void* storage /* >= sizeof(RecordRng) */) {
  RecordRng* r = new (storage) RecordRng{rng, rng_user, rec};
  *fn = record_fn;
  *user = r;
}
// field sum over the ranks of a host vector every rank holds a partial of (RCCL has no XOR / mod-p reduction: all_gather + fold)
static int comm_fold(const lfgpu_ligero_prover* pr, elt_t* y, size_t n) {
  const lfgpu_comm_ops& cm = pr->comm;
  if (cm.world == 1) return LFGPU_OK;
  std::vector<elt_t> all((size_t)cm.world * n);
  if (cm.all_gather(cm.user, y, all.data(), n * 16, 0, nullptr)) return lf_fail(pr->c, LFGPU_ERR_HIP, "ligero (sharded): all_gather hook failed");
  for (size_t j = 0; j < n; ++j) y[j] = all[j];
  for (int q = 1; q < cm.world; ++q) {
    const elt_t* v = all.data() + (size_t)q * n;
    if (pr->field == LFGPU_FIELD_GF2_128)
      for (size_t j = 0; j < n; ++j) y[j] = gf_add(y[j], v[j]);
    else
      for (size_t j = 0; j < n; ++j) y[j] = fp_add(y[j], v[j]);
  }
  return LFGPU_OK;
}

static int layout_sharded_host(int field, int k, const GfHostCtx* g, const lfgpu_ligero_param& p, const void* h_W, size_t subfield_boundary,
                               const size_t* h_lqc, lfgpu_rng_fn rng, void* user, const lfgpu_comm_ops* cm, bool exact, size_t row_lo, size_t row_hi,
                               elt_t* H, uint8_t* nonces, char* err) {
  // the RandomEngine is ONE sequential stream (ligero_prover.h:171-270, merkle_commitment.h:52-54): rank 0 draws all of it
  // (recorded while it lays out its own slab), the other ranks replay it and keep their rows
  std::vector<uint8_t> stream;
  LF_SCRUB_ON_EXIT(stream);
  if (cm->rank == 0) {  // rank 0 lays out its own slab while it draws: one pass, and the others wait for nothing else
    RecordRng rec{rng, user, &stream};
    const int rc = ligero_layout_host(field, k, g, p, (const elt_t*)h_W, subfield_boundary, h_lqc, record_fn, &rec, row_lo, row_hi, H, nonces, err, exact);
    const bool sent = lf_comm_bcast_blob(cm, stream) == LFGPU_OK;  // even after a failed layout: the others are waiting in the broadcast
    if (rc) return rc;
    if (!sent) {
      snprintf(err, 256, "ligero (sharded): broadcast hook failed");
      return LFGPU_ERR_HIP;
    }
    return LFGPU_OK;
  }
  if (lf_comm_bcast_blob(cm, stream)) {
    snprintf(err, 256, "ligero (sharded): broadcast hook failed");
    return LFGPU_ERR_HIP;
  }
  ReplayRng rep{&stream, 0, false};
  const int rc = ligero_layout_host(field, k, g, p, (const elt_t*)h_W, subfield_boundary, h_lqc, replay_fn, &rep, row_lo, row_hi, H, nonces, err, exact);
  if (rc) return rc;
  if (rep.dry || rep.pos != stream.size()) {
    snprintf(err, 256, "ligero (sharded): the replayed random stream does not match the layout (%zu of %zu bytes)", rep.pos, stream.size());
    return LFGPU_ERR_ASSERT;
  }
  return LFGPU_OK;
}
extern "C" int lfgpu_ligero_layout_rows_sharded(int field, int k, const lfgpu_ligero_param* pp, const void* h_W, size_t subfield_boundary,
                                                const size_t* h_lqc, lfgpu_rng_fn rng, void* user, const lfgpu_comm_ops* cm, void* h_rows,
                                                uint8_t* h_nonces) {
  if (!pp || !comm_ok(cm) || (cm->rank == 0 && !rng) || (pp->nw && !h_W) || (pp->nq && !h_lqc)) return LFGPU_ERR_ARG;
  if (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128) return LFGPU_ERR_ARG;
  GfHostCtx g;
  if (field == LFGPU_FIELD_GF2_128 && !lf_gf_ctx_build(&g, k)) return LFGPU_ERR_ARG;
  size_t lo, hi;
  LF_TRY(lfgpu_ligero_row_shard(pp, cm->rank, cm->world, &lo, &hi));
  if (hi > lo && !h_rows) return LFGPU_ERR_ARG;
  char err[256] = {0};
  return layout_sharded_host(field, k, &g, *pp, h_W, subfield_boundary, h_lqc, rng, user, cm, false, lo, hi, (elt_t*)h_rows, h_nonces, err);
}

extern "C" int lfgpu_ligero_commit_sharded(lfgpu_ctx* c, int field, int k, const lfgpu_ligero_param* pp, const void* h_W, size_t subfield_boundary,
                                           const size_t* h_lqc, lfgpu_rng_fn rng, void* user, const lfgpu_comm_ops* cm, uint8_t root_out[32],
                                           lfgpu_ligero_prover** out) {
  if (!c || !pp || !root_out || !out || !comm_ok(cm) || (cm->rank == 0 && !rng) || (pp->nw && !h_W) || (pp->nq && !h_lqc))
    return lf_fail(c, LFGPU_ERR_ARG, "ligero_commit_sharded: null argument / incomplete communicator");
  if (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128) return lf_fail(c, LFGPU_ERR_ARG, "ligero_commit_sharded: field");
  const lfgpu_ligero_param& p = *pp;
  if (p.ildt != 0 || p.idot != 1 || p.iquad != 2 || p.iw != 3) return lf_fail(c, LFGPU_ERR_ARG, "ligero_commit_sharded: row order");
  LF_HIP(c, hipSetDevice(c->device));
  const GfHostCtx* g = field == LFGPU_FIELD_GF2_128 ? lf_gf_ctx(c, k) : nullptr;
  if (field == LFGPU_FIELD_GF2_128 && !g) return lf_fail(c, LFGPU_ERR_ARG, "ligero_commit_sharded: k");
  const int world = cm->world, rank = cm->rank;
  lfgpu_ligero_prover* pr = new lfgpu_ligero_prover();
  pr->c = c;
  pr->field = field;
  pr->k = k;
  pr->p = p;
  pr->d_T = nullptr;
  pr->d_layers = nullptr;
  pr->nonces.resize(p.block_ext * 32);
  pr->sharded = true;
  pr->comm = *cm;
  pr->spans.resize(world);
  for (int q = 0; q < world; ++q) (void)lfgpu_ligero_row_shard(&p, q, world, &pr->spans[q].first, &pr->spans[q].second);
  pr->row_lo = pr->spans[rank].first;
  pr->row_hi = pr->spans[rank].second;
  void* d_send = nullptr;
  void* d_cols = nullptr;
  size_t send_bytes_tot = 0, cols_bytes = 0;
  // the exchange buffers hold encoded rows of the tableau (witness, pads, blinding rows): scrubbed like the tableau itself
  // (lfgpu_ligero_free) before they go back to the pool, in stream order behind their last use
  auto retire = [&](void*& d, size_t bytes) {
    if (!d) return;
    (void)hipMemsetAsync(d, 0, bytes, c->stream);
    lf_pool_put(c, d, bytes);
    d = nullptr;
  };
  auto fail = [&](int rc) {
    retire(d_send, send_bytes_tot);
    retire(d_cols, cols_bytes);
    lfgpu_ligero_free(pr);
    return rc;
  };
  const size_t nr = pr->row_hi - pr->row_lo, ld = p.block_enc, ncols = p.block_ext;
  static const bool verbose = getenv("LFGPU_VERBOSE") != nullptr;
  auto clk = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tv0 = clk();
  double tv1 = tv0, tv2 = tv0, tv3 = tv0, tv4 = tv0;
  // 1. host layout of this rank's slab from the one random stream
  std::vector<elt_t> H(std::max<size_t>(nr, 1) * p.dblock);
  LF_SCRUB_ON_EXIT(H);
  {
    char err[256] = {0};
    const int rc = layout_sharded_host(field, k, g, p, h_W, subfield_boundary, h_lqc, rng, user, cm, c->rng_exact != 0, pr->row_lo, pr->row_hi, H.data(),
                                       pr->nonces.data(), err);
    if (rc) return fail(lf_fail(c, rc, "%s", err));
  }
  tv1 = clk();
  // 2. RS-extend the slab: rows are independent, no collective
  const size_t tb = std::max<size_t>(nr, 1) * ld * 16, lb = 2 * ncols * 32;
  if (lf_pool_get(c, tb, (void**)&pr->d_T) != LFGPU_OK) return fail(lf_fail(c, LFGPU_ERR_NOMEM, "ligero_commit_sharded: slab alloc"));
  pr->T_bytes = tb;
  if (lf_pool_get(c, lb, (void**)&pr->d_layers) != LFGPU_OK) return fail(lf_fail(c, LFGPU_ERR_NOMEM, "ligero_commit_sharded: layers alloc"));
  pr->L_bytes = lb;
  int rc = nr ? ligero_encode_slab(c, field, k, p, pr->row_lo, pr->row_hi, H.data(), pr->d_T) : LFGPU_OK;
  if (rc) return fail(rc);
  // 3. a leaf hashes ALL rows of one column: one all_to_all re-partitions the encoded columns [dblock, block_enc) so that
  //    this rank owns columns [mc0, mc0 + mcn) of every row.  Send block for rank q: my rows x its columns, contiguous.
  std::vector<size_t> soff(world), sbytes(world), roff(world), rbytes(world);
  size_t mc0 = 0, mcn = 0;
  shard_split(ncols, rank, world, &mc0, &mcn);
  for (int q = 0; q < world; ++q) {
    size_t c0, cn;
    shard_split(ncols, q, world, &c0, &cn);
    soff[q] = send_bytes_tot;
    sbytes[q] = nr * cn * 16;
    send_bytes_tot += sbytes[q];
    roff[q] = pr->spans[q].first * mcn * 16;  // slabs are contiguous and ascending: the received blocks ARE the column matrix
    rbytes[q] = (pr->spans[q].second - pr->spans[q].first) * mcn * 16;
  }
  cols_bytes = std::max<size_t>(p.nrow * mcn * 16, 16);
  send_bytes_tot = std::max<size_t>(send_bytes_tot, 16);
  if (lf_pool_get(c, send_bytes_tot, &d_send) != LFGPU_OK || lf_pool_get(c, cols_bytes, &d_cols) != LFGPU_OK)
    return fail(lf_fail(c, LFGPU_ERR_NOMEM, "ligero_commit_sharded: exchange buffers"));
  for (int q = 0; q < world && nr; ++q) {
    size_t c0, cn;
    shard_split(ncols, q, world, &c0, &cn);
    if (cn && hipMemcpy2DAsync((uint8_t*)d_send + soff[q], cn * 16, pr->d_T + p.dblock + c0, ld * 16, cn * 16, nr, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
      return fail(lf_fail(c, LFGPU_ERR_HIP, "ligero_commit_sharded: pack"));
  }
  if (verbose) {
    (void)hipStreamSynchronize(c->stream);
    tv2 = clk();
  }
  if (cm->all_to_all(cm->user, d_send, soff.data(), sbytes.data(), d_cols, roff.data(), rbytes.data(), 1, c->stream))
    return fail(lf_fail(c, LFGPU_ERR_HIP, "ligero_commit_sharded: all_to_all hook failed"));
  tv3 = clk();
  // 4. local leaves of my columns, all_gather of the digests (ragged: padded to the largest share), the tree on every rank
  size_t maxn = 0, dummy = 0;
  shard_split(ncols, 0, world, &dummy, &maxn);
  void* d_non = nullptr;
  if ((rc = lf_scratch3(c, ncols * 32 + (size_t)(world + 1) * maxn * 32 + 64, &d_non))) return fail(rc);
  uint8_t* d_mine = (uint8_t*)d_non + ncols * 32;
  uint8_t* d_all = d_mine + maxn * 32;
  if (hipMemcpyAsync(d_non, pr->nonces.data(), ncols * 32, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
      hipMemsetAsync(d_mine, 0, maxn * 32, c->stream) != hipSuccess)
    return fail(lf_fail(c, LFGPU_ERR_HIP, "ligero_commit_sharded: nonce upload"));
  if (mcn && (rc = lfgpu_column_leaves(c, field, p.nrow, mcn, 0, mcn, d_cols, (const uint8_t*)d_non + mc0 * 32, d_mine))) return fail(rc);
  if (cm->all_gather(cm->user, d_mine, d_all, maxn * 32, 1, c->stream)) return fail(lf_fail(c, LFGPU_ERR_HIP, "ligero_commit_sharded: all_gather hook failed"));
  if (hipMemsetAsync(pr->d_layers, 0, ncols * 32, c->stream) != hipSuccess) return fail(lf_fail(c, LFGPU_ERR_HIP, "ligero_commit_sharded: layers"));
  for (int q = 0; q < world; ++q) {
    size_t c0, cn;
    shard_split(ncols, q, world, &c0, &cn);
    if (cn && hipMemcpyAsync(pr->d_layers + (ncols + c0) * 32, d_all + (size_t)q * maxn * 32, cn * 32, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
      return fail(lf_fail(c, LFGPU_ERR_HIP, "ligero_commit_sharded: leaves"));
  }
  if ((rc = lfgpu_merkle_build_tree(c, ncols, pr->d_layers, root_out))) return fail(rc);
  tv4 = clk();
  if (verbose)
    fprintf(stderr, "lfgpu ligero_commit_sharded rank %d/%d: rows [%zu, %zu): layout + stream %.2f ms | upload + RS encode + pack %.2f | all_to_all %.2f | leaves + all_gather + tree %.2f\n",
            rank, world, pr->row_lo, pr->row_hi, tv1 - tv0, tv2 - tv1, tv3 - tv2, tv4 - tv3);
  retire(d_send, send_bytes_tot);
  retire(d_cols, cols_bytes);
  *out = pr;
  return LFGPU_OK;
}

// host-only check of a caller's hooks
extern "C" int lfgpu_comm_selftest(const lfgpu_comm_ops* cm) {
  if (!comm_ok(cm)) return LFGPU_ERR_ARG;
  const int W = cm->world, r = cm->rank;
  {  // broadcast from every root
    for (int root = 0; root < W; ++root) {
      uint8_t b[37];
      for (int i = 0; i < 37; ++i) b[i] = (uint8_t)(r == root ? 11 * root + i : 0xEE);
      if (cm->broadcast(cm->user, b, 37, root, 0, nullptr)) return LFGPU_ERR_HIP;
      for (int i = 0; i < 37; ++i)
        if (b[i] != (uint8_t)(11 * root + i)) return LFGPU_ERR_ASSERT;
    }
  }
  {  // all_gather
    std::vector<uint8_t> mine(53), all((size_t)53 * W);
    for (int i = 0; i < 53; ++i) mine[i] = (uint8_t)(7 * r + 3 * i);
    if (cm->all_gather(cm->user, mine.data(), all.data(), 53, 0, nullptr)) return LFGPU_ERR_HIP;
    for (int q = 0; q < W; ++q)
      for (int i = 0; i < 53; ++i)
        if (all[(size_t)53 * q + i] != (uint8_t)(7 * q + 3 * i)) return LFGPU_ERR_ASSERT;
  }
  {  // ragged all_to_all: rank p sends 5 + p + 2 q bytes to rank q (zero-length blocks included: p = q = 0 sends 5)
    std::vector<size_t> so(W), sb(W), ro(W), rb(W);
    size_t st = 0, rt = 0;
    for (int q = 0; q < W; ++q) {
      so[q] = st;
      sb[q] = (size_t)(5 + r + 2 * q) * ((r + q) % 3 != 2);
      st += sb[q];
      ro[q] = rt;
      rb[q] = (size_t)(5 + q + 2 * r) * ((q + r) % 3 != 2);
      rt += rb[q];
    }
    std::vector<uint8_t> send(st + 1), recv(rt + 1, 0xEE);
    for (int q = 0; q < W; ++q)
      for (size_t i = 0; i < sb[q]; ++i) send[so[q] + i] = (uint8_t)(r * 31 + q * 17 + i);
    if (cm->all_to_all(cm->user, send.data(), so.data(), sb.data(), recv.data(), ro.data(), rb.data(), 0, nullptr)) return LFGPU_ERR_HIP;
    for (int q = 0; q < W; ++q)
      for (size_t i = 0; i < rb[q]; ++i)
        if (recv[ro[q] + i] != (uint8_t)(q * 31 + r * 17 + i)) return LFGPU_ERR_ASSERT;
  }
  return LFGPU_OK;
}

extern "C" int lfgpu_ligero_free(lfgpu_ligero_prover* pr) {
  if (!pr) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = pr->c;
  // keep the buffers for the next commit of the same shape (the context's pool); the stream is in order, so work still
  // queued on them finishes before anything a later commit enqueues
  auto stash = [&](void* p, size_t bytes) {
    if (!p) return;
    // the tableau holds the witness, the pads and the blinding rows: scrub it before it outlives its prover (enqueued on the
    // stream, in order behind the prover's last kernels)
    if (bytes) (void)hipMemsetAsync(p, 0, bytes, c ? c->stream : nullptr);
    if (c && bytes) lf_pool_put(c, p, bytes);
    else (void)hipFree(p);
  };
  if (pr->owns) {
    stash(pr->d_T, pr->T_bytes);
    stash(pr->d_layers, pr->L_bytes);
  }
  delete pr;
  return LFGPU_OK;
}
extern "C" int lfgpu_ligero_tableau(lfgpu_ligero_prover* pr, void** d_T) {
  if (!pr || !d_T) return LFGPU_ERR_ARG;
  *d_T = pr->d_T;
  return LFGPU_OK;
}

// y[j] = T0[j] + sum_i A[i][j] * T[i][j]        (dot_proof accumulation, Blas::vaxpy blas.h:71-78)
// Workgroup = 64 columns x 16 row slices (a lane per column alone would be <= 4 workgroups of 150 sequential
// products each: latency-bound); slices are folded through LDS.
template <int F>
__global__ __launch_bounds__(1024) void rows_vaxpy_kernel(u32 nrows, size_t n, const elt_t* __restrict__ T0, const elt_t* __restrict__ A,
                                                          size_t lda, const elt_t* __restrict__ T, size_t ld, elt_t* __restrict__ y) {
  __shared__ elt_t part[16][64];
  const u32 col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const size_t j = (size_t)blockIdx.x * 64 + col;
  elt_t acc = elt_zero();
  if (j < n)
    for (u32 i = slice; i < nrows; i += 16)
      acc = Fld<F>::add(acc, Fld<F>::mul(ld16(&T[(size_t)i * ld + j]), ld16(&A[(size_t)i * lda + j])));
  part[slice][col] = acc;
  __syncthreads();
  if (slice == 0 && j < n) {
    acc = ld16(&T0[j]);
    for (u32 k = 0; k < 16; ++k) acc = Fld<F>::add(acc, part[k][col]);
    st16(&y[j], acc);
  }
}
// y[j] = Tq[j] + sum_i u[i] * (z_i[j] - x_i[j]*y_i[j])   (quadratic_proof :311-333)
template <int F>
__global__ void quad_combo_kernel(u32 nt, size_t n, const elt_t* __restrict__ Tq, const elt_t* __restrict__ u,
                                  const elt_t* __restrict__ X, const elt_t* __restrict__ Y, const elt_t* __restrict__ Z,
                                  size_t ld, elt_t* __restrict__ y) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  elt_t acc = ld16(&Tq[j]);
  for (u32 i = 0; i < nt; ++i) {
    elt_t t = Fld<F>::sub(ld16(&Z[(size_t)i * ld + j]), Fld<F>::mul(ld16(&X[(size_t)i * ld + j]), ld16(&Y[(size_t)i * ld + j])));
    acc = Fld<F>::add(acc, Fld<F>::mul(ld16(&u[i]), t));
  }
  st16(&y[j], acc);
}
// Aext[i] = [0^r | A[i*w .. (i+1)*w) | 0...]   (layout_Aext ligero_param.h:423-430)
__global__ void layout_aext_kernel(u32 r, u32 w, size_t lda, const elt_t* __restrict__ A, elt_t* __restrict__ Aext) {
  u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  u32 i = blockIdx.y;
  if (j >= r + w) return;
  st16(&Aext[(size_t)i * lda + j], j < r ? elt_zero() : ld16(&A[(size_t)i * w + (j - r)]));
}

#define LIG_DISPATCH(field, KERNEL, grid, block, ...)                                  \
  do {                                                                                 \
    if ((field) == LFGPU_FIELD_GF2_128)                                                \
      hipLaunchKernelGGL(KERNEL<FIELD_GF2_128>, grid, block, 0, c->stream, __VA_ARGS__); \
    else                                                                               \
      hipLaunchKernelGGL(KERNEL<FIELD_FP128>, grid, block, 0, c->stream, __VA_ARGS__);   \
  } while (0)

// witness / quadratic rows [a_lo, a_hi) (indices relative to iw) that this prover's slab holds
static void owned_wq_rows(const lfgpu_ligero_prover* pr, size_t* a_lo, size_t* a_hi) {
  const lfgpu_ligero_param& p = pr->p;
  const size_t lo = std::min(std::max(pr->row_lo, p.iw), p.iw + p.nwqrow), hi = std::min(std::max(pr->row_hi, p.iw), p.iw + p.nwqrow);
  *a_lo = lo - p.iw;
  *a_hi = std::max(hi, lo) - p.iw;
}

extern "C" int lfgpu_ligero_low_degree_proof(lfgpu_ligero_prover* pr, const void* h_u, void* h_y) {
  if (!pr || !h_u || !h_y) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = pr->c;
  const lfgpu_ligero_param& p = pr->p;
  LF_HIP(c, hipSetDevice(c->device));
  void* dy = nullptr;
  LF_TRY(lf_scratch3(c, p.block * 16, &dy));
  if (pr->has(p.ildt))
    LF_HIP(c, hipMemcpyAsync(dy, pr->row(p.ildt), p.block * 16, hipMemcpyDeviceToDevice, c->stream));
  else
    LF_HIP(c, hipMemsetAsync(dy, 0, p.block * 16, c->stream));
  size_t a_lo, a_hi;
  owned_wq_rows(pr, &a_lo, &a_hi);
  if (a_hi > a_lo)
    LF_TRY(lfgpu_rows_axpy(c, pr->field, a_hi - a_lo, p.block, dy, (const uint64_t*)h_u + 2 * a_lo, pr->row(p.iw + a_lo), p.block_enc));
  LF_TRY(lfgpu_memcpy_d2h(c, h_y, dy, p.block * 16));
  return pr->sharded ? comm_fold(pr, (elt_t*)h_y, p.block) : LFGPU_OK;
}

// ---- the inner-product matrix built on the device (inner_product_vector + layout_Aext, ligero_param.h:382-430)
// rows[i][r + j] = scale * dense[i*w + j] over the first ndense flat positions (the private-input block of the last
// constraint), then rows[pos(idx)] += val for the caller's sparse terms (duplicates already folded by the caller)
template <int F>
__global__ void a_rows_dense_kernel(u32 r, u32 w, size_t ld, elt_t scale, const elt_t* __restrict__ dense, size_t n,
                                    elt_t* __restrict__ rows) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t i = t / w, j = t % w;
  st16(&rows[i * ld + r + j], Fld<F>::mul(scale, ld16(&dense[t])));
}
template <int F>
__global__ void a_rows_sparse_kernel(u32 r, u32 w, size_t ld, const u64* __restrict__ idx, const elt_t* __restrict__ val, size_t n,
                                     elt_t* __restrict__ rows) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const size_t i = idx[t] / w, j = idx[t] % w;
  elt_t* dst = &rows[i * ld + r + j];
  st16(dst, Fld<F>::add(ld16(dst), ld16(&val[t])));
}

extern "C" int lfgpu_ligero_inner_product_rows(lfgpu_ctx* c, int field, size_t w, size_t r, size_t ld, size_t nrows, const void* d_dense,
                                               size_t ndense, const uint64_t scale[2], const uint64_t* h_idx, const void* h_val,
                                               size_t nsparse, void* d_rows) {
  if (!c || !d_rows || w == 0 || r + w > ld || (ndense && (!d_dense || !scale)) || (nsparse && (!h_idx || !h_val)) || ndense > nrows * w)
    return lf_fail(c, LFGPU_ERR_ARG, "ligero_inner_product_rows: bad argument");
  if (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "ligero_inner_product_rows: field");
  for (size_t t = 0; t < nsparse; ++t) {
    if (h_idx[t] >= nrows * w) return lf_fail(c, LFGPU_ERR_ARG, "ligero_inner_product_rows: sparse index out of range");
    if (t && h_idx[t] <= h_idx[t - 1]) return lf_fail(c, LFGPU_ERR_ARG, "ligero_inner_product_rows: sparse indices must be strictly increasing");
  }
  LF_HIP(c, hipSetDevice(c->device));
  LF_HIP(c, hipMemset2DAsync(d_rows, ld * 16, 0, (r + w) * 16, nrows, c->stream));
  if (ndense) {
    const elt_t sc{scale[0], scale[1]};
    LIG_DISPATCH(field, a_rows_dense_kernel, dim3((u32)((ndense + 255) / 256)), dim3(256), (u32)r, (u32)w, ld, sc, (const elt_t*)d_dense, ndense,
                 (elt_t*)d_rows);
  }
  if (nsparse) {
    void* d_sp = nullptr;
    LF_TRY(lf_scratch2(c, nsparse * 24 + 64, &d_sp));
    elt_t* d_val = (elt_t*)d_sp;
    u64* d_idx = (u64*)(d_val + nsparse);
    if (nsparse * 24 <= LF_STAGE_SLOT) {  // one staged copy: values then indices
      std::vector<uint8_t> pack(nsparse * 24);
      memcpy(pack.data(), h_val, nsparse * 16);
      memcpy(pack.data() + nsparse * 16, h_idx, nsparse * 8);
      LF_TRY(lf_stage_upload(c, d_sp, pack.data(), pack.size()));
    } else {
      LF_HIP(c, hipMemcpyAsync(d_val, h_val, nsparse * 16, hipMemcpyHostToDevice, c->stream));
      LF_HIP(c, hipMemcpyAsync(d_idx, h_idx, nsparse * 8, hipMemcpyHostToDevice, c->stream));
      LF_HIP(c, hipStreamSynchronize(c->stream));
    }
    LIG_DISPATCH(field, a_rows_sparse_kernel, dim3((u32)((nsparse + 255) / 256)), dim3(256), (u32)r, (u32)w, ld, (const u64*)d_idx,
                 (const elt_t*)d_val, nsparse, (elt_t*)d_rows);
  }
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

// dot_proof once the rows [0^r | A_i | 0...] stand in dAext (lda = dblock; all nwqrow rows, of which the slab's are
// used): extend, combine, read back
static int dot_proof_finish(lfgpu_ligero_prover* pr, elt_t* dAext, elt_t* dy, void* h_y) {
  lfgpu_ctx* c = pr->c;
  const lfgpu_ligero_param& p = pr->p;
  const size_t lda = p.dblock;
  size_t a_lo, a_hi;
  owned_wq_rows(pr, &a_lo, &a_hi);
  const elt_t* T0 = nullptr;
  if (pr->has(p.idot)) {
    T0 = pr->row(p.idot);
  } else {  // a zero row behind dy (the callers reserve 2 * dblock)
    LF_HIP(c, hipMemsetAsync(dy + p.dblock, 0, p.dblock * 16, c->stream));
    T0 = dy + p.dblock;
  }
  if (a_hi > a_lo) LF_TRY(lf_rs_rows(c, pr->field, pr->k, a_hi - a_lo, p.block, p.dblock, dAext + a_lo * lda, lda));
  LIG_DISPATCH(pr->field, rows_vaxpy_kernel, dim3((u32)((p.dblock + 63) / 64)), dim3(1024), (u32)(a_hi - a_lo), p.dblock, T0,
               (const elt_t*)(dAext + a_lo * lda), lda, (const elt_t*)(a_hi > a_lo ? pr->row(p.iw + a_lo) : pr->d_T), p.block_enc, dy);
  LF_HIP(c, hipGetLastError());
  LF_TRY(lfgpu_memcpy_d2h(c, h_y, dy, p.dblock * 16));
  return pr->sharded ? comm_fold(pr, (elt_t*)h_y, p.dblock) : LFGPU_OK;
}

extern "C" int lfgpu_ligero_dot_proof(lfgpu_ligero_prover* pr, const void* h_A, void* h_y) {
  if (!pr || !h_A || !h_y) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = pr->c;
  const lfgpu_ligero_param& p = pr->p;
  LF_HIP(c, hipSetDevice(c->device));
  const size_t lda = p.dblock;
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, (p.nwqrow * p.w + p.nwqrow * lda + 2 * p.dblock) * 16 + 64, &sc));
  elt_t* dA = (elt_t*)sc;
  elt_t* dAext = dA + p.nwqrow * p.w;
  elt_t* dy = dAext + p.nwqrow * lda;
  LF_HIP(c, hipMemcpyAsync(dA, h_A, p.nwqrow * p.w * 16, hipMemcpyHostToDevice, c->stream));
  LF_HIP(c, hipMemsetAsync(dAext, 0, p.nwqrow * lda * 16, c->stream));
  hipLaunchKernelGGL(layout_aext_kernel, dim3((u32)((p.block + 255) / 256), (u32)p.nwqrow), dim3(256), 0, c->stream,
                     (u32)p.r, (u32)p.w, lda, (const elt_t*)dA, dAext);
  return dot_proof_finish(pr, dAext, dy, h_y);
}

extern "C" int lfgpu_ligero_dot_proof_sparse(lfgpu_ligero_prover* pr, const void* d_dense, size_t ndense, const uint64_t scale[2],
                                             const uint64_t* h_idx, const void* h_val, size_t nsparse, void* h_y) {
  if (!pr || !h_y) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = pr->c;
  const lfgpu_ligero_param& p = pr->p;
  LF_HIP(c, hipSetDevice(c->device));
  const size_t lda = p.dblock;
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, (p.nwqrow * lda + 2 * p.dblock) * 16 + 64, &sc));
  if (d_dense && ndense) {  // the dense block must not live in the scratch this call is about to overwrite
    const uint8_t* lo = (const uint8_t*)sc;
    const uint8_t* d = (const uint8_t*)d_dense;
    if (d + ndense * 16 > lo && d < lo + (p.nwqrow * lda + 2 * p.dblock) * 16) return lf_fail(c, LFGPU_ERR_ARG, "ligero_dot_proof_sparse: dense block aliases scratch");
  }
  elt_t* dAext = (elt_t*)sc;
  elt_t* dy = dAext + p.nwqrow * lda;
  LF_HIP(c, hipMemsetAsync(dAext, 0, p.nwqrow * lda * 16, c->stream));
  LF_TRY(lfgpu_ligero_inner_product_rows(c, pr->field, p.w, p.r, lda, p.nwqrow, d_dense, ndense, scale, h_idx, h_val, nsparse, dAext));
  return dot_proof_finish(pr, dAext, dy, h_y);
}

extern "C" int lfgpu_ligero_quadratic_proof(lfgpu_ligero_prover* pr, const void* h_u_quad, void* h_y0, void* h_y2) {
  if (!pr || !h_y0 || !h_y2 || (pr->p.nqtriples && !h_u_quad)) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = pr->c;
  const lfgpu_ligero_param& p = pr->p;
  LF_HIP(c, hipSetDevice(c->device));
  // a triple (x_i, y_i, z_i) is multiplied element-wise, so a slab must hold all quadratic rows or none of them
  const size_t q_lo = p.iq, q_hi = p.iq + 3 * p.nqtriples;
  const bool all_q = p.nqtriples == 0 || (pr->row_lo <= q_lo && q_hi <= pr->row_hi);
  const bool no_q = pr->row_hi <= q_lo || pr->row_lo >= q_hi;
  if (!all_q && !no_q) return lf_fail(c, LFGPU_ERR_ARG, "quadratic_proof: the slab splits the quadratic rows (shard so that one rank holds [iq, nrow))");
  const size_t nt = all_q ? p.nqtriples : 0;
  void* sc = nullptr;
  LF_TRY(lf_scratch3(c, (p.nqtriples + 2 * p.dblock) * 16 + 64, &sc));
  elt_t* du = (elt_t*)sc;
  elt_t* dy = du + p.nqtriples + 1;
  elt_t* dz = dy + p.dblock;
  if (nt) LF_HIP(c, hipMemcpyAsync(du, h_u_quad, p.nqtriples * 16, hipMemcpyHostToDevice, c->stream));
  const size_t ld = p.block_enc;
  const elt_t* Tq = nullptr;
  if (pr->has(p.iquad)) {
    Tq = pr->row(p.iquad);
  } else {
    LF_HIP(c, hipMemsetAsync(dz, 0, p.dblock * 16, c->stream));
    Tq = dz;
  }
  const elt_t* X = nt ? pr->row(p.iq) : pr->d_T;
  const elt_t* Y = X + p.nqtriples * ld;
  const elt_t* Z = Y + p.nqtriples * ld;
  LIG_DISPATCH(pr->field, quad_combo_kernel, dim3((u32)((p.dblock + 255) / 256)), dim3(256), (u32)nt, p.dblock, Tq, (const elt_t*)du, X, Y, Z, ld, dy);
  LF_HIP(c, hipGetLastError());
  std::vector<elt_t> y(p.dblock);
  LF_TRY(lfgpu_memcpy_d2h(c, y.data(), dy, p.dblock * 16));
  if (pr->sharded) LF_TRY(comm_fold(pr, y.data(), p.dblock));
  if (pr->sharded || (pr->row_lo == 0 && pr->row_hi == p.nrow))  // sanity check of the reference (:335-337); a bare slab's partial sums are checked by its caller after the fold
    for (size_t j = 0; j < p.w; ++j)
      if (y[p.r + j].lo | y[p.r + j].hi) return lf_fail(c, LFGPU_ERR_ASSERT, "quadratic_proof: W part is nonzero");
  memcpy(h_y0, y.data(), p.r * 16);
  memcpy(h_y2, y.data() + p.block, (p.dblock - p.block) * 16);
  return LFGPU_OK;
}

// a prover over rows [row_lo, row_hi) that the caller has laid out and encoded (lfgpu_ligero_encode_rows) in d_slab;
// d_layers / h_nonces: the Merkle heap and nonces of the whole commitment (for lfgpu_ligero_open), may be null
extern "C" int lfgpu_ligero_prover_from_slab(lfgpu_ctx* c, int field, int k, const lfgpu_ligero_param* pp, size_t row_lo, size_t row_hi,
                                             void* d_slab, void* d_layers, const uint8_t* h_nonces, lfgpu_ligero_prover** out) {
  if (!c || !pp || !out || row_lo > row_hi || row_hi > pp->nrow || (row_hi > row_lo && !d_slab))
    return lf_fail(c, LFGPU_ERR_ARG, "ligero_prover_from_slab: bad argument");
  if (field != LFGPU_FIELD_GF2_128 && field != LFGPU_FIELD_FP128) return lf_fail(c, LFGPU_ERR_ARG, "ligero_prover_from_slab: field");
  lfgpu_ligero_prover* pr = new lfgpu_ligero_prover();
  pr->c = c;
  pr->field = field;
  pr->k = k;
  pr->p = *pp;
  pr->d_T = (elt_t*)d_slab;
  pr->d_layers = (uint8_t*)d_layers;
  pr->row_lo = row_lo;
  pr->row_hi = row_hi;
  pr->owns = false;
  if (h_nonces) pr->nonces.assign(h_nonces, h_nonces + 32 * pp->block_ext);
  *out = pr;
  return LFGPU_OK;
}

extern "C" int lfgpu_ligero_open(lfgpu_ligero_prover* pr, const size_t* idx, void* h_req, uint8_t* h_nonces,
                                 uint8_t* h_path, size_t path_cap, size_t* npath) {
  if (!pr || !idx || !h_req || !h_nonces || !npath) return LFGPU_ERR_ARG;
  lfgpu_ctx* c = pr->c;
  const lfgpu_ligero_param& p = pr->p;
  for (size_t i = 0; i < p.nreq; ++i)
    if (idx[i] >= p.block_ext) return lf_fail(c, LFGPU_ERR_ARG, "ligero_open: index out of range");
  if (pr->nonces.size() != 32 * p.block_ext || !pr->d_layers) return lf_fail(c, LFGPU_ERR_ARG, "ligero_open: this prover holds no commitment");
  // a slab returns its own rows of req ((row_hi - row_lo) x nreq, in row order); a sharded prover gathers every rank's rows
  const size_t nr = pr->row_hi - pr->row_lo;
  std::vector<uint8_t> mine;
  size_t maxr = nr;
  if (pr->sharded) {
    for (const auto& sp : pr->spans) maxr = std::max(maxr, sp.second - sp.first);
    mine.assign(maxr * p.nreq * 16, 0);
  }
  void* h_dst = pr->sharded ? (void*)mine.data() : h_req;
  if (nr) {
    void* dreq = nullptr;
    LF_TRY(lf_scratch3(c, nr * p.nreq * 16, &dreq));
    LF_TRY(lfgpu_gather_columns(c, nr, p.block_enc, p.dblock, pr->d_T, idx, p.nreq, dreq));
    LF_TRY(lfgpu_memcpy_d2h(c, h_dst, dreq, nr * p.nreq * 16));
  }
  if (pr->sharded) {
    const lfgpu_comm_ops& cm = pr->comm;
    std::vector<uint8_t> all((size_t)cm.world * mine.size());
    if (cm.all_gather(cm.user, mine.data(), all.data(), mine.size(), 0, nullptr)) return lf_fail(c, LFGPU_ERR_HIP, "ligero_open (sharded): all_gather hook failed");
    for (int q = 0; q < cm.world; ++q)
      memcpy((uint8_t*)h_req + pr->spans[q].first * p.nreq * 16, all.data() + (size_t)q * mine.size(), (pr->spans[q].second - pr->spans[q].first) * p.nreq * 16);
  }
  for (size_t i = 0; i < p.nreq; ++i) memcpy(h_nonces + 32 * i, &pr->nonces[32 * idx[i]], 32);
  return lfgpu_merkle_open(c, p.block_ext, pr->d_layers, idx, p.nreq, h_path, path_cap, npath);
}
