// rs.hip -- K3 (GF(2^128) LCH14 Reed-Solomon row encode) and K4 (Fp128 RS row encode).
//
// K3 reference: LCH14ReedSolomon::interpolate (lib/gf2k/lch14_reed_solomon.h:49-103)
// with LCH14::BidirectionalFFT / bidir_recur (lib/gf2k/lch14.h:146-217).
//
// The truncated ("bidirectional") transform is an irregular recursion, but for a
// given (l, n) it is a FIXED sequence of butterfly sweeps, identical for every
// row.  The host unrolls the recursion once into an op list (cached per (k,n,m));
// one workgroup per row keeps the 2^l-coefficient vector in LDS and replays it:
//   phase 1: bidirectional FFT  -> missing evaluations of coset 0 + coefficients
//   phase 2: one forward FFT per further coset, streamed out to the row.
// HBM traffic per row is the algorithmic minimum: n reads + (m - n) writes.
//
// K4 reference: ReedSolomon::interpolate (lib/algebra/reed_solomon.h:93-110) via
// FFTConvolution (lib/algebra/convolution.h:56-106), built on lfgpu_fp128_fft.
#include <string>

#include "ctx.h"

#define RS_THREADS 512

enum { OP_FWD = 0, OP_BWD = 1, OP_DIAG = 2, OP_FFT_STAGE = 3, OP_IFFT_STAGE = 4 };
struct RsOp {
  u32 kind;
  u32 i;       // stage: s = 1 << i
  u32 base;    // offset of the sub-array in the coefficient vector
  u32 lo, hi;  // range ops: uv in [lo, hi); stage ops: lo = log2(size of the sub-FFT)
  u32 tw;      // offset into the twiddle pool (range ops: 1 entry; stage ops: 2^(L-1-i) entries)
};

struct RsPlan {
  const RsOp* ops;
  const elt_t* tw;
  u32 nops_bidir;    // ops [0, nops_bidir): phase 1
  u32 l, n, m;
  u32 ncoset;        // cosets 1 .. ncoset-1 follow
  u32 coset_tw_off;  // twiddle pool offset of coset c's table: coset_tw_off + (c-1)*(2^l - 1)
};

__device__ __forceinline__ void run_op(elt_t* B, const RsOp op, const elt_t* __restrict__ twp) {
  const u32 tid = threadIdx.x;
  const u32 s = 1u << op.i;
  if (op.kind <= OP_DIAG) {
    const elt_t tw = ld16(&twp[op.tw]);
    for (u32 uv = op.lo + tid; uv < op.hi; uv += RS_THREADS) {
      elt_t b0 = ld16(&B[op.base + uv]), b1 = ld16(&B[op.base + uv + s]);
      if (op.kind == OP_FWD) {  // lch14.h:219-223
        b0 = gf_add(b0, gf_mul(tw, b1));
        b1 = gf_add(b1, b0);
      } else if (op.kind == OP_BWD) {  // :225-229
        b1 = gf_add(b1, b0);
        b0 = gf_add(b0, gf_mul(tw, b1));
      } else {  // diag :232-237: forward at [uv+s], backward at [uv]
        elt_t t = b1;
        b1 = gf_add(b1, b0);
        b0 = gf_add(b0, gf_mul(tw, t));
      }
      st16(&B[op.base + uv], b0);
      st16(&B[op.base + uv + s], b1);
    }
  } else {
    const u32 half = 1u << (op.lo - 1);
    for (u32 b = tid; b < half; b += RS_THREADS) {
      u32 v = b & (s - 1), u = b >> op.i;
      u32 i0 = op.base + (u << (op.i + 1)) + v, i1 = i0 + s;
      elt_t tw = ld16(&twp[op.tw + u]);
      elt_t b0 = ld16(&B[i0]), b1 = ld16(&B[i1]);
      if (op.kind == OP_FFT_STAGE) {
        b0 = gf_add(b0, gf_mul(tw, b1));
        b1 = gf_add(b1, b0);
      } else {
        b1 = gf_add(b1, b0);
        b0 = gf_add(b0, gf_mul(tw, b1));
      }
      st16(&B[i0], b0);
      st16(&B[i1], b1);
    }
  }
}

// rows [lo2, hi2) use plan p2 (another n): the Ligero tableau mixes block- and dblock-long rows (ligero_prover.h:171-270)
// and one launch for all of them beats three latency-bound ones
__global__ __launch_bounds__(RS_THREADS) void gf_rs_rows_kernel(RsPlan p1, RsPlan p2, u32 lo2, u32 hi2, elt_t* __restrict__ T, size_t ld) {
  const RsPlan& p = (blockIdx.x >= lo2 && blockIdx.x < hi2) ? p2 : p1;
  extern __shared__ elt_t lds[];
  const u32 fftn = 1u << p.l, tid = threadIdx.x;
  elt_t* Cc = lds;          // coefficients
  elt_t* Wk = lds + fftn;   // work buffer for the coset FFTs
  elt_t* y = T + (size_t)blockIdx.x * ld;
  for (u32 i = tid; i < fftn; i += RS_THREADS) st16(&Cc[i], i < p.n ? ld16(&y[i]) : elt_zero());
  __syncthreads();
  for (u32 o = 0; o < p.nops_bidir; ++o) {
    run_op(Cc, p.ops[o], p.tw);
    __syncthreads();
  }
  // missing evaluations of the first coset, then revert to pure coefficients
  const u32 top = p.m < fftn ? p.m : fftn;
  for (u32 i = p.n + tid; i < fftn; i += RS_THREADS) {
    if (i < top) st16(&y[i], ld16(&Cc[i]));
    st16(&Cc[i], elt_zero());
  }
  __syncthreads();
  for (u32 cs = 1; cs < p.ncoset; ++cs) {
    const u32 b = cs << p.l;
    for (u32 i = tid; i < fftn; i += RS_THREADS) st16(&Wk[i], ld16(&Cc[i]));
    __syncthreads();
    u32 off = p.coset_tw_off + (cs - 1) * (fftn - 1);
    for (u32 step = 0; step < p.l; ++step) {
      u32 i = p.l - 1 - step;
      RsOp op{OP_FFT_STAGE, i, 0, p.l, 0, off};
      // table layout: stage l-1 first ... stage 0 last
      run_op(Wk, op, p.tw);
      off += 1u << (p.l - 1 - i);
      __syncthreads();
    }
    for (u32 i = tid; i < fftn && b + i < p.m; i += RS_THREADS) st16(&y[b + i], ld16(&Wk[i]));
    __syncthreads();
  }
}

// ---- rows larger than LDS (2^l > 4096): coefficients in a device work buffer Cc[row][2^l].  The recursion of
// bidir_recur is unrolled on the host into (a) whole FFT / IFFT blocks, which run through the batched LCH14 transform
// (bit-sliced for >= 32 rows: a handful of passes each instead of one global sweep per butterfly stage), and (b) the
// partial butterfly ranges between them (one elementwise launch each); the further cosets are transformed in place in the
// rows where they fit.
__global__ __launch_bounds__(256) void gf_rs_big_load_kernel(u32 n, u32 fftn, const elt_t* __restrict__ T, size_t ld, elt_t* __restrict__ Cc) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= fftn) return;
  const size_t r = blockIdx.y;
  st16(&Cc[r * fftn + i], i < n ? ld16(&T[r * ld + i]) : elt_zero());
}
// butterflies uv in [lo, hi) of one level: kind OP_FWD / OP_BWD / OP_DIAG (lch14.h:219-237)
__global__ __launch_bounds__(256) void gf_rs_big_range_kernel(u32 kind, u32 s, u32 base, u32 lo, u32 hi, elt_t tw, u32 fftn, elt_t* __restrict__ Cc) {
  elt_t* B = Cc + (size_t)blockIdx.y * fftn;
  const u32 uv = lo + blockIdx.x * 256 + threadIdx.x;
  if (uv >= hi) return;
  const u32 i0 = base + uv, i1 = i0 + s;
  elt_t b0 = ld16(&B[i0]), b1 = ld16(&B[i1]);
  if (kind == OP_FWD) {
    b0 = gf_add(b0, gf_mul(tw, b1));
    b1 = gf_add(b1, b0);
  } else if (kind == OP_BWD) {
    b1 = gf_add(b1, b0);
    b0 = gf_add(b0, gf_mul(tw, b1));
  } else {
    const elt_t x = b1;
    b1 = gf_add(b1, b0);
    b0 = gf_add(b0, gf_mul(tw, x));
  }
  st16(&B[i0], b0);
  st16(&B[i1], b1);
}
// dst[r][0..w) = src[r][0..w) (a strided copy of 16-byte elements; all index arithmetic in 64 bits)
__global__ __launch_bounds__(256) void gf_rs_big_copy_kernel(u32 w, const elt_t* __restrict__ src, size_t sld, elt_t* __restrict__ dst, size_t dld) {
  const u32 i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w) return;
  const size_t r = blockIdx.y;
  st16(&dst[r * dld + i], ld16(&src[r * sld + i]));
}
// evaluations n..top of the first coset out to the rows; coefficients n..fftn back to zero
__global__ __launch_bounds__(256) void gf_rs_big_store_kernel(u32 n, u32 top, u32 fftn, elt_t* __restrict__ T, size_t ld, elt_t* __restrict__ Cc) {
  const u32 i = n + blockIdx.x * 256 + threadIdx.x;
  if (i >= fftn) return;
  const size_t r = blockIdx.y;
  if (i < top) st16(&T[r * ld + i], ld16(&Cc[r * fftn + i]));
  st16(&Cc[r * fftn + i], elt_zero());
}

// ---- host: unroll bidir_recur (lch14.h:185-217) into ops
struct PlanBuilder {
  const GfHostCtx* g;
  std::vector<RsOp> ops;
  std::vector<elt_t> tw;
  u32 single(unsigned i, u64 coset) {
    tw.push_back(h_lch14_twiddle(g, i, coset));
    return (u32)tw.size() - 1;
  }
  // stage table for stage i of a size-2^L FFT with `coset`: tw[u] = twiddle(i, coset ^ (u << (i+1)))
  u32 stage_table(unsigned i, unsigned L, u64 coset) {
    u32 off = (u32)tw.size();
    for (u32 u = 0; u < (1u << (L - 1 - i)); ++u) tw.push_back(h_lch14_twiddle(g, i, coset ^ ((u64)u << (i + 1))));
    return off;
  }
  void fft(unsigned L, u64 coset, u32 base) {  // lch14.h:106-124
    for (unsigned i = L; i-- > 0;) ops.push_back(RsOp{OP_FFT_STAGE, i, base, L, 0, stage_table(i, L, coset)});
  }
  void ifft(unsigned L, u64 coset, u32 base) {  // lch14.h:126-144
    for (unsigned i = 0; i < L; ++i) ops.push_back(RsOp{OP_IFFT_STAGE, i, base, L, 0, stage_table(i, L, coset)});
  }
  void bidir(unsigned i, u64 coset, u32 k, u32 base) {
    if (i-- > 0) {
      u32 s = 1u << i;
      u32 t = single(i, coset);
      if (k < s) {
        if (k < s) ops.push_back(RsOp{OP_FWD, i, base, k, s, t});
        bidir(i, coset, k, base);
        if (k > 0) ops.push_back(RsOp{OP_DIAG, i, base, 0, k, t});
        if (i > 0) fft(i, coset + s, base + s);
      } else {
        if (i > 0) ifft(i, coset, base);
        if (k - s < s) ops.push_back(RsOp{OP_DIAG, i, base, k - s, s, t});
        bidir(i, coset + s, k - s, base + s);
        if (k - s > 0) ops.push_back(RsOp{OP_BWD, i, base, 0, k - s, t});
      }
    }
  }
};


static int gf_rs_plan(lfgpu_ctx* c, const GfHostCtx* g, int k, size_t n, size_t m, RsPlan* out) {
  const unsigned l = lf_log2(n);
  RsPlan plan;
  char kb[96];
  snprintf(kb, sizeof(kb), "rsplan:%d:%zu:%zu", k, n, m);
  std::string key(kb);
  auto it = c->blobs.find(key);
  if (it == c->blobs.end()) {
    PlanBuilder pb{g, {}, {}};
    pb.bidir(l, 0, (u32)n, 0);
    plan.nops_bidir = (u32)pb.ops.size();
    plan.l = l;
    plan.n = (u32)n;
    plan.m = (u32)m;
    u32 ncoset = 1;
    while (((size_t)ncoset << l) < m) ++ncoset;
    plan.ncoset = ncoset;
    plan.coset_tw_off = (u32)pb.tw.size();
    for (u32 cs = 1; cs < ncoset; ++cs)
      for (unsigned i = l; i-- > 0;) pb.stage_table(i, l, (u64)cs << l);
    if (pb.ops.empty()) pb.ops.push_back(RsOp{OP_FWD, 0, 0, 0, 0, 0});
    if (pb.tw.empty()) pb.tw.push_back(elt_t{0, 0});
    void *dops = nullptr, *dtw = nullptr;
    LF_TRY(lf_table(c, key + ":ops", pb.ops.data(), pb.ops.size() * sizeof(RsOp), &dops));
    LF_TRY(lf_table(c, key + ":tw", pb.tw.data(), pb.tw.size() * 16, &dtw));
    plan.ops = (const RsOp*)dops;
    plan.tw = (const elt_t*)dtw;
    c->blobs[key] = std::string((const char*)&plan, sizeof(plan));
  } else {
    memcpy(&plan, it->second.data(), sizeof(plan));
  }
  *out = plan;
  return LFGPU_OK;
}

extern "C" int lfgpu_gf2128_lch14_fft(lfgpu_ctx*, int, int, size_t, unsigned, uint64_t, void*, size_t);

enum { BIG_FFT = 10, BIG_IFFT = 11 };
struct BigOp {
  u32 kind;    // OP_FWD / OP_BWD / OP_DIAG: butterflies uv in [lo, hi) at level i;  BIG_FFT / BIG_IFFT: a whole block of 2^i points
  u32 i, base, lo, hi;
  u64 coset;
  elt_t tw;
};
// bidir_recur (lch14.h:185-217) unrolled with its FFT / IFFT calls kept whole
static void big_bidir(const GfHostCtx* g, std::vector<BigOp>& ops, unsigned i, u64 coset, u32 k, u32 base) {
  if (i-- > 0) {
    const u32 s = 1u << i;
    const elt_t t = h_lch14_twiddle(g, i, coset);
    if (k < s) {
      ops.push_back(BigOp{OP_FWD, i, base, k, s, coset, t});
      big_bidir(g, ops, i, coset, k, base);
      if (k > 0) ops.push_back(BigOp{OP_DIAG, i, base, 0, k, coset, t});
      if (i > 0) ops.push_back(BigOp{BIG_FFT, i, base + s, 0, 0, coset + s, t});
    } else {
      if (i > 0) ops.push_back(BigOp{BIG_IFFT, i, base, 0, 0, coset, t});
      if (k - s < s) ops.push_back(BigOp{OP_DIAG, i, base, k - s, s, coset, t});
      big_bidir(g, ops, i, coset + s, k - s, base + s);
      if (k - s > 0) ops.push_back(BigOp{OP_BWD, i, base, 0, k - s, coset, t});
    }
  }
}

// lch_bs.hip: the bit-sliced tower representation as a work format (unit buffers of combos x stride columns)
size_t lf_bs_units_bytes(lfgpu_ctx* c, int k, size_t rows, u32 stride);
int lf_bs_tower_op(lfgpu_ctx* c, int k, int op, size_t rows, u32 a0, u32 a1, u32 a2, u32 a3, u64 coset, elt_t tw, void* U, u32 stride, const void* src,
                   void* dst, size_t ld);
enum { BS_OP_CIN = 0, BS_OP_COUT = 1, BS_OP_FFT = 2, BS_OP_RANGE = 3, BS_OP_COPY = 4 };

// >= 32 rows: the whole encoder in the tower representation -- ONE conversion in, every FFT / IFFT block of the truncated
// transform as butterfly passes on a sub-block of the unit buffer, the partial ranges as bit-sliced elementwise launches (a
// product by the level's one twiddle is ~20 XORs per element there, ~510 VALU operations in gf_mul), one conversion out per
// coset of evaluations.  Round 2 before this: every block and coset converted in and out on its own (S-lig 70.4 ms).
static int gf_rs_rows_big_tower(lfgpu_ctx* c, int k, size_t nrow, size_t n, size_t m, elt_t* T, size_t ld, unsigned l, const std::vector<BigOp>& ops) {
  const u32 fftn = 1u << l;
  const size_t ub = lf_bs_units_bytes(c, k, nrow, fftn);
  void *U = nullptr, *U2 = nullptr;
  LF_TRY(lf_scratch2(c, ub, &U));
  const elt_t z{0, 0};
  LF_TRY(lf_bs_tower_op(c, k, BS_OP_CIN, nrow, fftn, (u32)n, 0, 0, 0, z, U, fftn, T, nullptr, ld));
  for (const BigOp& op : ops) {
    if (op.kind == BIG_FFT || op.kind == BIG_IFFT) {
      LF_TRY(lf_bs_tower_op(c, k, BS_OP_FFT, nrow, op.i, op.kind == BIG_IFFT ? 1u : 0u, op.base, 0, op.coset, z, U, fftn, nullptr, nullptr, 0));
    } else if (op.hi > op.lo) {
      const u32 kind = op.kind == OP_FWD ? 0u : op.kind == OP_BWD ? 1u : 2u;
      LF_TRY(lf_bs_tower_op(c, k, BS_OP_RANGE, nrow, kind, 1u << op.i, op.base + op.lo, op.hi - op.lo, 0, op.tw, U, fftn, nullptr, nullptr, 0));
    }
  }
  const u32 top = m < fftn ? (u32)m : fftn;
  if (n < fftn) LF_TRY(lf_bs_tower_op(c, k, BS_OP_COUT, nrow, fftn, (u32)n, top, 0, 0, z, U, fftn, nullptr, T, ld));  // evaluations n..top of the first coset
  if (m > fftn) LF_TRY(lf_scratch(c, ub, &U2));
  for (size_t base = fftn; base < m; base += fftn) {  // further cosets: FFT of the coefficients [c_0 .. c_{n-1}, 0 ...] with coset offset `base`
    // out of place: the first butterfly pass reads the coefficients from U (columns >= n as zero) and writes U2
    LF_TRY(lf_bs_tower_op(c, k, BS_OP_FFT, nrow, l, 0, 0, (u32)n, (u64)base, z, U2, fftn, U, nullptr, 0));
    const u32 w = (u32)std::min<size_t>(fftn, m - base);
    LF_TRY(lf_bs_tower_op(c, k, BS_OP_COUT, nrow, fftn, 0, w, 0, 0, z, U2, fftn, nullptr, T + base, ld));
  }
  return LFGPU_OK;
}

static int gf_rs_rows_big(lfgpu_ctx* c, const GfHostCtx* g, int k, size_t nrow, size_t n, size_t m, elt_t* T, size_t ld, unsigned l) {
  if ((size_t)1 << g->sub_bits < ((size_t)1 << l)) return lf_fail(c, LFGPU_ERR_ARG, "gf2128_rs_encode_rows: 2^l exceeds the subfield of GF2_128<%d>", k);
  const u32 fftn = 1u << l;
  char kb[96];
  snprintf(kb, sizeof(kb), "rsbig2:%d:%zu", k, n);
  const std::string key(kb);
  auto it = c->blobs.find(key);
  if (it == c->blobs.end()) {
    std::vector<BigOp> ops;
    big_bidir(g, ops, l, 0, (u32)n, 0);
    c->blobs[key] = std::string((const char*)ops.data(), ops.size() * sizeof(BigOp));
    it = c->blobs.find(key);
  }
  const size_t nops = it->second.size() / sizeof(BigOp);
  std::vector<BigOp> ops(nops);  // copy out: the FFT calls below may insert into c->blobs
  memcpy(ops.data(), it->second.data(), nops * sizeof(BigOp));
  {
    static const int tower = getenv("LFGPU_RS_TOWER") ? atoi(getenv("LFGPU_RS_TOWER")) : 1;  // 0: per-block conversions (A/B)
    if (tower && nrow >= 32 && l >= 7) return gf_rs_rows_big_tower(c, k, nrow, n, m, T, ld, l, ops);
  }
  void* sc = nullptr;
  LF_TRY(lf_scratch2(c, nrow * fftn * 16, &sc));  // scratch2: the batched FFT takes `scratch`
  elt_t* Cc = (elt_t*)sc;
  hipLaunchKernelGGL(gf_rs_big_load_kernel, dim3((fftn + 255) / 256, (u32)nrow), dim3(256), 0, c->stream, (u32)n, fftn, (const elt_t*)T, ld, Cc);
  for (const BigOp& op : ops) {
    if (op.kind == BIG_FFT || op.kind == BIG_IFFT) {
      LF_TRY(lfgpu_gf2128_lch14_fft(c, k, op.kind == BIG_IFFT ? 1 : 0, nrow, op.i, op.coset, Cc + op.base, fftn));
    } else if (op.hi > op.lo) {
      hipLaunchKernelGGL(gf_rs_big_range_kernel, dim3((op.hi - op.lo + 255) / 256, (u32)nrow), dim3(256), 0, c->stream, op.kind, 1u << op.i, op.base,
                         op.lo, op.hi, op.tw, fftn, Cc);
    }
  }
  const u32 top = m < fftn ? (u32)m : fftn;
  if (n < fftn)
    hipLaunchKernelGGL(gf_rs_big_store_kernel, dim3((fftn - (u32)n + 255) / 256, (u32)nrow), dim3(256), 0, c->stream, (u32)n, top, fftn, T, ld, Cc);
  LF_HIP(c, hipGetLastError());
  for (size_t base = fftn; base < m; base += fftn) {  // further cosets: FFT of the coefficients with coset offset `base`
    if (base + fftn <= m) {  // fits: copy the coefficients into the row and transform in place (lch14_reed_solomon.h:84-90)
      hipLaunchKernelGGL(gf_rs_big_copy_kernel, dim3((fftn + 255) / 256, (u32)nrow), dim3(256), 0, c->stream, fftn, (const elt_t*)Cc, (size_t)fftn, T + base, ld);
      LF_TRY(lfgpu_gf2128_lch14_fft(c, k, 0, nrow, l, (uint64_t)base, T + base, ld));
    } else {  // partial fit, last coset: transform the work buffer and copy what fits (:91-98)
      LF_TRY(lfgpu_gf2128_lch14_fft(c, k, 0, nrow, l, (uint64_t)base, Cc, fftn));
      hipLaunchKernelGGL(gf_rs_big_copy_kernel, dim3((u32)((m - base + 255) / 256), (u32)nrow), dim3(256), 0, c->stream, (u32)(m - base), (const elt_t*)Cc,
                         (size_t)fftn, T + base, ld);
      LF_HIP(c, hipGetLastError());
    }
  }
  return LFGPU_OK;
}

extern "C" int lfgpu_gf2128_rs_encode_rows(lfgpu_ctx* c, int k, size_t nrow, size_t n, size_t m, void* d_T,
                                           size_t ld) {
  if (!c || (!d_T && nrow)) return lf_fail(c, LFGPU_ERR_ARG, "gf2128_rs_encode_rows: null argument");
  const GfHostCtx* g = lf_gf_ctx(c, k);
  if (!g) return lf_fail(c, LFGPU_ERR_ARG, "gf2128_rs_encode_rows: subfield_log_bits must be 4 or 5");
  if (n == 0 || m < n || ld < m) return lf_fail(c, LFGPU_ERR_ARG, "gf2128_rs_encode_rows: need 0 < n <= m <= ld");
  if (nrow == 0 || m == n) return LFGPU_OK;
  const unsigned l = lf_log2(n);
  // evaluation points of_scalar(j), j < m, must exist in the subfield (ligero_param.h:197-202)
  if (k < 6 && g->sub_bits < 64 && m > ((size_t)1 << g->sub_bits))
    return lf_fail(c, LFGPU_ERR_ARG, "gf2128_rs_encode_rows: m exceeds the subfield domain");
  if (l > 20) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "gf2128_rs_encode_rows: n > 2^20");
  LF_HIP(c, hipSetDevice(c->device));
  if (l > 12) return gf_rs_rows_big(c, g, k, nrow, n, m, (elt_t*)d_T, ld, l);

  RsPlan plan;
  LF_TRY(gf_rs_plan(c, g, k, n, m, &plan));
  size_t lds = (size_t)32 << l;
  if (!(c->attr_done & 2u)) {
    LF_HIP(c, hipFuncSetAttribute((const void*)gf_rs_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 << 12));
    c->attr_done |= 2u;
  }
  hipLaunchKernelGGL(gf_rs_rows_kernel, dim3((u32)nrow), dim3(RS_THREADS), lds, c->stream, plan, plan, 0u, 0u, (elt_t*)d_T, ld);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

// all rows of a Ligero tableau in one launch: rows [lo2, hi2) are n2 long, the others n1 (both <= 4096 -> LDS rows);
// returns LFGPU_ERR_UNSUPPORTED when the shapes need the general path (the caller then encodes group by group)
int lf_gf_rs_rows_mixed(lfgpu_ctx* c, int k, size_t nrow, size_t n1, size_t n2, size_t lo2, size_t hi2, size_t m, elt_t* d_T, size_t ld) {
  const GfHostCtx* g = lf_gf_ctx(c, k);
  if (!g || n1 == 0 || n2 == 0 || m <= n1 || m <= n2 || ld < m || lf_log2(n1) > 12 || lf_log2(n2) > 12 ||
      (g->sub_bits < 64 && m > ((size_t)1 << g->sub_bits)))
    return LFGPU_ERR_UNSUPPORTED;
  LF_HIP(c, hipSetDevice(c->device));
  RsPlan p1, p2;
  LF_TRY(gf_rs_plan(c, g, k, n1, m, &p1));
  LF_TRY(gf_rs_plan(c, g, k, n2, m, &p2));
  if (!(c->attr_done & 2u)) {
    LF_HIP(c, hipFuncSetAttribute((const void*)gf_rs_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 << 12));
    c->attr_done |= 2u;
  }
  const size_t lds = (size_t)32 << (p1.l > p2.l ? p1.l : p2.l);
  hipLaunchKernelGGL(gf_rs_rows_kernel, dim3((u32)nrow), dim3(RS_THREADS), lds, c->stream, p1, p2, (u32)lo2, (u32)hi2, d_T, ld);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

extern "C" int lfgpu_gf2128_rs_encode_tableau(lfgpu_ctx* c, int k, size_t nrow, size_t n1, size_t n2, size_t lo2, size_t hi2, size_t m,
                                              void* d_T, size_t ld) {
  if (!c || (!d_T && nrow) || lo2 > hi2 || hi2 > nrow) return lf_fail(c, LFGPU_ERR_ARG, "gf2128_rs_encode_tableau: bad argument");
  if (nrow == 0) return LFGPU_OK;
  const int rc = lf_gf_rs_rows_mixed(c, k, nrow, n1, n2, lo2, hi2, m, (elt_t*)d_T, ld);
  if (rc == LFGPU_ERR_UNSUPPORTED) return lf_fail(c, rc, "gf2128_rs_encode_tableau: shapes need lfgpu_gf2128_rs_encode_rows per group");
  return rc;
}

extern "C" int lfgpu_gf2128_rs_encode_rows_host(lfgpu_ctx* c, int k, size_t nrow, size_t n, size_t m, void* h_T,
                                                size_t ld) {
  if (!c || !h_T) return LFGPU_ERR_ARG;
  void* d = nullptr;
  size_t bytes = nrow * ld * 16;
  LF_TRY(lf_scratch2(c, bytes, &d));
  LF_HIP(c, hipMemcpyAsync(d, h_T, bytes, hipMemcpyHostToDevice, c->stream));
  LF_TRY(lfgpu_gf2128_rs_encode_rows(c, k, nrow, n, m, d, ld));
  LF_HIP(c, hipMemcpyAsync(h_T, d, bytes, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}

// ------------------------------------------------------------------ K4: Fp128
// x_i = binom_i * y_i (i < n), zero-padded to P;  z = fftb(fftf(x) . yhat);  y_k = lead_{k-d} * z_k (k >= n)
__global__ void fp_rs_pre_kernel(u32 n, u32 P, const elt_t* __restrict__ binom, const elt_t* __restrict__ T, size_t ld,
                                 elt_t* __restrict__ X) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  elt_t v = elt_zero();
  if (i < n) v = fp_mul(ld16(&binom[i]), ld16(&T[(size_t)blockIdx.y * ld + i]));
  st16(&X[(size_t)blockIdx.y * P + i], v);
}
__global__ void fp_rs_pointwise_kernel(u32 P, const elt_t* __restrict__ yhat, elt_t* __restrict__ X) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  elt_t* x = &X[(size_t)blockIdx.y * P + i];
  st16(x, fp_mul(ld16(x), ld16(&yhat[i])));
}
__global__ void fp_rs_post_kernel(u32 n, u32 m, u32 P, const elt_t* __restrict__ lead, const elt_t* __restrict__ X,
                                  elt_t* __restrict__ T, size_t ld) {
  u32 i = n + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  st16(&T[(size_t)blockIdx.y * ld + i], fp_mul(ld16(&lead[i - (n - 1)]), ld16(&X[(size_t)blockIdx.y * P + i])));
}

extern "C" int lfgpu_fp128_rs_encode_rows(lfgpu_ctx* c, size_t nrow, size_t n, size_t m, const uint64_t omega[2],
                                          uint64_t omega_order, void* d_T, size_t ld) {
  if (!c || !omega || (!d_T && nrow)) return lf_fail(c, LFGPU_ERR_ARG, "fp128_rs_encode_rows: null argument");
  if (n == 0 || m < n || ld < m) return lf_fail(c, LFGPU_ERR_ARG, "fp128_rs_encode_rows: need 0 < n <= m <= ld");
  if (nrow == 0 || m == n) return LFGPU_OK;
  size_t P = 1;
  while (P < m) P <<= 1;
  if (P > omega_order) return lf_fail(c, LFGPU_ERR_ARG, "fp128_rs_encode_rows: omega_order < padding");
  if (P > ((size_t)1 << 20)) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "fp128_rs_encode_rows: m > 2^20 not covered yet");
  LF_HIP(c, hipSetDevice(c->device));
  const size_t d = n - 1;
  char kb[128];
  snprintf(kb, sizeof(kb), "fprs:%zu:%zu:%llx:%llx:%llx", n, m, (u64)omega[0], (u64)omega[1], (u64)omega_order);
  std::string key(kb);
  void *dbinom = nullptr, *dlead = nullptr, *dyhat = nullptr;
  if (!lf_table_lookup(c, key + ":yhat", &dyhat)) {
    // constants of the ReedSolomon ctor (reed_solomon.h:51-88)
    std::vector<elt_t> inv(m), lead(m - n + 1), binom(n), yh(P, elt_t{0, 0});
    {  // batch inverse of 1..m-1 (AlgebraUtil::batch_inverse_arithmetic); inv[0] = 0
      std::vector<elt_t> pre(m);
      elt_t acc = h_fp_of_scalar(1);
      for (size_t i = 1; i < m; ++i) {
        pre[i] = acc;
        acc = fp_mul(acc, h_fp_of_scalar(i));
      }
      elt_t ia = m > 1 ? h_fp_inv(acc) : acc;
      inv[0] = elt_t{0, 0};
      for (size_t i = m; i-- > 1;) {
        inv[i] = fp_mul(ia, pre[i]);
        ia = fp_mul(ia, h_fp_of_scalar(i));
      }
    }
    elt_t one = h_fp_of_scalar(1), zero{0, 0};
    lead[0] = one;
    binom[0] = one;
    for (size_t i = 1; i + d < m; ++i) lead[i] = fp_mul(lead[i - 1], fp_mul(h_fp_of_scalar(d + i), inv[i]));
    for (size_t kk = d; kk < m; ++kk) {
      lead[kk - d] = fp_mul(lead[kk - d], h_fp_of_scalar(kk - d));
      if (d % 2 == 1) lead[kk - d] = fp_sub(zero, lead[kk - d]);
    }
    for (size_t i = 1; i < n; ++i) binom[i] = fp_mul(binom[i - 1], fp_mul(h_fp_of_scalar(n - i), inv[i]));
    for (size_t i = 1; i < n; i += 2) binom[i] = fp_sub(zero, binom[i]);
    // yhat = fftf(pad(inverses)) / P  (convolution.h:64-75), computed with the device FFT
    for (size_t i = 0; i < m; ++i) yh[i] = inv[i];
    void* tmp = nullptr;
    LF_TRY(lf_scratch2(c, P * 16, &tmp));
    LF_HIP(c, hipMemcpy(tmp, yh.data(), P * 16, hipMemcpyHostToDevice));
    LF_TRY(lfgpu_fp128_fft(c, 1, 1, P, omega, omega_order, tmp, P));
    LF_HIP(c, hipStreamSynchronize(c->stream));
    LF_HIP(c, hipMemcpy(yh.data(), tmp, P * 16, hipMemcpyDeviceToHost));
    elt_t sc = h_fp_inv(h_fp_of_scalar(P));
    for (size_t i = 0; i < P; ++i) yh[i] = fp_mul(yh[i], sc);
    LF_TRY(lf_table(c, key + ":binom", binom.data(), binom.size() * 16, &dbinom));
    LF_TRY(lf_table(c, key + ":lead", lead.data(), lead.size() * 16, &dlead));
    LF_TRY(lf_table(c, key + ":yhat", yh.data(), yh.size() * 16, &dyhat));
  } else {
    lf_table_lookup(c, key + ":binom", &dbinom);
    lf_table_lookup(c, key + ":lead", &dlead);
  }
  void* X = nullptr;
  LF_TRY(lf_scratch2(c, nrow * P * 16, &X));
  dim3 gp((u32)((P + 255) / 256), (u32)nrow);
  hipLaunchKernelGGL(fp_rs_pre_kernel, gp, dim3(256), 0, c->stream, (u32)n, (u32)P, (const elt_t*)dbinom,
                     (const elt_t*)d_T, ld, (elt_t*)X);
  LF_TRY(lfgpu_fp128_fft(c, 1, nrow, P, omega, omega_order, X, P));
  hipLaunchKernelGGL(fp_rs_pointwise_kernel, gp, dim3(256), 0, c->stream, (u32)P, (const elt_t*)dyhat, (elt_t*)X);
  LF_TRY(lfgpu_fp128_fft(c, 0, nrow, P, omega, omega_order, X, P));
  dim3 go((u32)((m - n + 255) / 256), (u32)nrow);
  hipLaunchKernelGGL(fp_rs_post_kernel, go, dim3(256), 0, c->stream, (u32)n, (u32)m, (u32)P, (const elt_t*)dlead,
                     (const elt_t*)X, (elt_t*)d_T, ld);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

extern "C" int lfgpu_fp128_rs_encode_rows_host(lfgpu_ctx* c, size_t nrow, size_t n, size_t m, const uint64_t omega[2], uint64_t omega_order,
                                               void* h_T, size_t ld) {
  if (!c || !h_T) return LFGPU_ERR_ARG;
  void* d = nullptr;
  const size_t bytes = nrow * ld * 16;
  LF_TRY(lf_scratch4(c, bytes, &d));  // scratch4: the encode below works through scratch (FFT) and scratch2 (convolution)
  LF_HIP(c, hipMemcpyAsync(d, h_T, bytes, hipMemcpyHostToDevice, c->stream));
  LF_TRY(lfgpu_fp128_rs_encode_rows(c, nrow, n, m, omega, omega_order, d, ld));
  LF_HIP(c, hipMemcpyAsync(h_T, d, bytes, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}
