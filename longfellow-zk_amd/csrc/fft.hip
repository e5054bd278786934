// fft.hip -- K1 (Fp128 radix-2 FFT) and K2 (GF(2^128) LCH14 additive FFT).
//
// Both transforms are run as one or two "tile passes".  A pass gives each
// workgroup a tile of T points x C batch columns (T*C <= 8192 elements = 128 KiB
// of the CU's 160 KiB LDS), runs log2(T) butterfly stages entirely in LDS and
// writes the tile back once, so HBM sees one read + one write per pass:
//   n <= 8192 : one pass, batch = rows
//   n  > 8192 : n = n1 * n2 (n2 = 1024).  Pass A: n1-point transforms down the
//               strided dimension (C consecutive columns per tile => coalesced
//               C*16-byte segments), pass B: n2-point transforms on contiguous rows.
// Reference: lib/algebra/fft.h:70-201 (Fp128), lib/gf2k/lch14.h:92-144 (LCH14).
#include <string>

#include "ctx.h"

int lf_lch14_fft_bitsliced(lfgpu_ctx* c, int k, int inverse, size_t rows, unsigned l, u64 coset, void* d_B, size_t ld);

#define TILE_ELTS 8192u  // largest tile (128 KiB)
// Tile geometry: 2^13 elements / 1024 threads (one workgroup per CU) or 2^12 elements / 512 threads
// (two workgroups per CU, so one workgroup's HBM phase overlaps the other's butterfly phase).
// c->tile_log = 12 by default: measured 46.9 ms vs 54.8 ms per 2^20 x 1024 Fp128 batch (profiles/r01)

struct TilePlan {
  const elt_t* src;
  elt_t* dst;
  long long src_row, dst_row;    // grid.y (row) strides, elements
  long long src_tile, dst_tile;  // grid.x (tile) strides
  long long sk, sc, dk, dc;      // strides of the point index k and batch index c
  u32 logT, logC, nbatch;        // nbatch: total batch columns over all tiles (bounds)
  u32 kfast_src, kfast_dst;      // 1: consecutive lanes walk k (stride-1 side), 0: walk c
  u32 wlds;                      // Fp128: stage twiddles staged in LDS behind the tile
};

__device__ __forceinline__ u32 bitrev(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// LDS slot of point k, column c.  Rows are C*16 bytes; bit-reversed / strided row indices would put
// the 8 lanes of one ds_write_b128 pass on the same banks, so the top four bits of k are XORed into
// the low slot bits (a bijection inside every aligned 16-slot = 256-byte bank row).
__device__ __forceinline__ u32 lds_slot(u32 k, u32 c, u32 logT, u32 logC) {
  u32 slot = (k << logC) + c;
  if (logT >= 8 && logC >= 3) slot ^= (k >> (logT - 4)) & 15u;
  return slot;
}

// ------------------------------------------------------------------ K1: Fp128 (and the F64_2 of the reference's FFT tests)
// What K1 needs of a field with 16-byte elements: add, sub, the product with a twiddle (a power of the root) and, on the
// host, the general product, 1 and the inverse.  Fp128 is the field of the prover; F64_2 = Fp2<Fp<1>> over
// p = 2^64 - 2^32 + 1 is the second field of lib/algebra/fft_test.cc:205-229 (BM_FFT_F64_2) -- same plan, same kernel.
struct Fp128Ops {
  static constexpr const char* kKey = "fp";
  static LF_HD elt_t add(elt_t a, elt_t b) { return fp_add(a, b); }
  static LF_HD elt_t sub(elt_t a, elt_t b) { return fp_sub(a, b); }
  static LF_HD elt_t mul_tw(elt_t a, elt_t w) { return fp_mul(a, w); }
  static elt_t hmul(elt_t a, elt_t b) { return fp_mul(a, b); }
  static elt_t one() { return h_fp_of_scalar(1); }
  static elt_t inv(elt_t a) { return h_fp_inv(a); }
};
static u64 h_f64_pow(u64 x, u64 e) {
  u64 r = 0xFFFFFFFFull;  // 2^64 mod p: the Montgomery image of 1
  for (; e; e >>= 1) {
    if (e & 1) r = f64_mul(r, x);
    x = f64_mul(x, x);
  }
  return r;
}
// REAL: the root lies in the base field (as in the reference's test and benchmark, fft_test.cc:211-217), so every
// twiddle has a zero imaginary part and its product costs two base-field products instead of three
template <bool REAL>
struct F64x2Ops {
  static constexpr const char* kKey = "f64x2";
  static LF_HD elt_t add(elt_t a, elt_t b) { return f64x2_add(a, b); }
  static LF_HD elt_t sub(elt_t a, elt_t b) { return f64x2_sub(a, b); }
  static LF_HD elt_t mul_tw(elt_t a, elt_t w) { return REAL ? f64x2_mul_real(a, w.lo) : f64x2_mul(a, w); }
  static elt_t hmul(elt_t a, elt_t b) { return f64x2_mul(a, b); }
  static elt_t one() { return elt_t{0xFFFFFFFFull, 0ull}; }
  static elt_t inv(elt_t a) {  // Fp2::invert (fp2.h:112-123): conj(a) / (re^2 + im^2)
    const u64 d = f64_add(f64_mul(a.lo, a.lo), f64_mul(a.hi, a.hi));
    return f64x2_mul_real(elt_t{a.lo, f64_sub(0, a.hi)}, h_f64_pow(d, F64_P - 2));
  }
};

// W[i << wshift] = w_T^i (i < T/2).  Optional inter-pass twiddle w_n^{j*(col)}
// = tw_lo[e & 1023] * tw_hi[e >> 10].
template <class O, int R, int FFT_THREADS>
__device__ __forceinline__ void fp_radix_round(elt_t* s, const elt_t* Wl, u32 wsh, u32 logT, u32 logC, u32 st, u32 tid) {
  constexpr u32 N = 1u << R;
  const u32 T = 1u << logT, C = 1u << logC, m = 1u << st;
  for (u32 e = tid; e < (T >> R) * C; e += FFT_THREADS) {
    const u32 c = e & (C - 1), b = e >> logC;
    const u32 j = b & (m - 1);
    const u32 i0 = ((b >> st) << (st + R)) + j;
    elt_t x[N];
    u32 q[N];
#pragma unroll
    for (u32 a = 0; a < N; ++a) {
      q[a] = lds_slot(i0 + a * m, c, logT, logC);
      x[a] = ld16(&s[q[a]]);
    }
#pragma unroll
    for (u32 t = 0; t < (u32)R; ++t) {
      const u32 half = 1u << t;
#pragma unroll
      for (u32 a = 0; a < N; ++a) {
        if (a & half) continue;
        const u32 jj = j + (a & (half - 1)) * m;
        if (t > 0 || j) {
          if ((a & (half - 1)) || j) x[a + half] = O::mul_tw(x[a + half], ld16(&Wl[(size_t)(jj << (logT - 1 - st - t)) << wsh]));
        }
        const elt_t u = x[a], v = x[a + half];
        x[a] = O::add(u, v);
        x[a + half] = O::sub(u, v);
      }
    }
#pragma unroll
    for (u32 a = 0; a < N; ++a) st16(&s[q[a]], x[a]);
  }
}

// tw_hi == nullptr with tw_lo != nullptr: tw_lo is the FULL inter-pass table [j][column] (one product per element
// instead of two; rows of the grid then vary fastest so that a tile's slice of the table stays in L2 while the
// batch rows stream past it).
// WLDS: the stage twiddles sit in LDS behind the tile (p.wlds; always, for 2^12-element tiles) -- a template parameter so that
// their reads are LDS instructions with 32-bit addresses instead of flat loads through a generic pointer
template <class O, int FFT_THREADS, bool WLDS>
__global__ __launch_bounds__(FFT_THREADS) void fp_fft_tile(TilePlan p, const elt_t* __restrict__ W, u32 wshift,
                                                           const elt_t* __restrict__ tw_lo,
                                                           const elt_t* __restrict__ tw_hi, u32 row_fast) {
  extern __shared__ elt_t s[];
  const u32 T = 1u << p.logT, C = 1u << p.logC, tid = threadIdx.x;
  const u32 bx = row_fast ? blockIdx.y : blockIdx.x, by = row_fast ? blockIdx.x : blockIdx.y;  // (tile, batch row)
  const u32 cbase = bx << p.logC;
  const elt_t* src = p.src + (long long)by * p.src_row + (long long)bx * p.src_tile;
  elt_t* dst = p.dst + (long long)by * p.dst_row + (long long)bx * p.dst_tile;

  for (u32 e = tid; e < T * C; e += FFT_THREADS) {
    u32 k, c;
    if (p.kfast_src) { k = e & (T - 1); c = e >> p.logT; } else { c = e & (C - 1); k = e >> p.logC; }
    elt_t v = elt_zero();
    if (cbase + c < p.nbatch) v = ld16(src + (long long)k * p.sk + (long long)c * p.sc);
    st16(&s[lds_slot(bitrev(k, p.logT), c, p.logT, p.logC)], v);
  }
  // the T/2 stage twiddles w_T^i go behind the tile in LDS: three twiddle reads per radix-4 step then cost an LDS
  // access instead of a vector-memory instruction each
  elt_t* const wl = s + (T << p.logC);  // stage twiddle i at Wl[i << wsh]
  if (WLDS)
    for (u32 i = tid; i < (T >> 1); i += FFT_THREADS) st16(&wl[i], ld16(&W[(size_t)i << wshift]));
  const u32 wsh = WLDS ? 0 : wshift;
  __syncthreads();
  // R stages per LDS round trip on 2^R register-resident points x[a] = s[i0 + a*m]: sub-stage t pairs (a, a + 2^t)
  // with w_T^(jj * T / (2m 2^t)), jj = j + (a mod 2^t) * m.  Same products as radix 2, 1/R of the LDS traffic and
  // barriers, and 2^(R-1) independent carry chains in flight per thread.
  u32 st = 0;
  while (st < p.logT) {
    const u32 rem = p.logT - st;
    const u32 R = (rem == 3 || rem > 4) ? 3 : (rem >= 2 ? 2 : 1);
    if (WLDS) {
      if (R == 3) fp_radix_round<O, 3, FFT_THREADS>(s, wl, wsh, p.logT, p.logC, st, tid);
      else if (R == 2) fp_radix_round<O, 2, FFT_THREADS>(s, wl, wsh, p.logT, p.logC, st, tid);
      else fp_radix_round<O, 1, FFT_THREADS>(s, wl, wsh, p.logT, p.logC, st, tid);
    } else {
      if (R == 3) fp_radix_round<O, 3, FFT_THREADS>(s, W, wsh, p.logT, p.logC, st, tid);
      else if (R == 2) fp_radix_round<O, 2, FFT_THREADS>(s, W, wsh, p.logT, p.logC, st, tid);
      else fp_radix_round<O, 1, FFT_THREADS>(s, W, wsh, p.logT, p.logC, st, tid);
    }
    __syncthreads();
    st += R;
  }
  for (u32 e = tid; e < T * C; e += FFT_THREADS) {
    u32 j, c;
    if (p.kfast_dst) { j = e & (T - 1); c = e >> p.logT; } else { c = e & (C - 1); j = e >> p.logC; }
    if (cbase + c >= p.nbatch) continue;
    elt_t v = ld16(&s[lds_slot(j, c, p.logT, p.logC)]);
    if (tw_lo) {
      const u32 col = cbase + c, ex = j * col;
      if (ex) {
        if (tw_hi) {
          elt_t t = ld16(&tw_lo[ex & 1023]);
          if (ex >> 10) t = O::mul_tw(t, ld16(&tw_hi[ex >> 10]));
          v = O::mul_tw(v, t);
        } else {
          v = O::mul_tw(v, ld16(&tw_lo[(size_t)j * p.nbatch + col]));
        }
      }
    }
    st16(dst + (long long)j * p.dk + (long long)c * p.dc, v);
  }
}

// ------------------------------------------------------------------ K2: LCH14
// Stage ii of the tile (global stage i = i_lo + ii) uses
//   tw = tbl[off[ii] + u_local]  (^ base[ii*nb + cbase + c] when base != null)
// fwd (lch14.h:219-223): b0 ^= tw*b1; b1 ^= b0.   bwd (:225-229): b1 ^= b0; b0 ^= tw*b1.
struct LchTables {
  const elt_t* tbl;
  const elt_t* base;
  u32 nb;
  u32 off[16];
};

template <int FFT_THREADS>
__global__ __launch_bounds__(FFT_THREADS) void lch_fft_tile(TilePlan p, int inverse, LchTables t) {
  extern __shared__ elt_t s[];
  const u32 T = 1u << p.logT, C = 1u << p.logC, tid = threadIdx.x;
  const u32 cbase = blockIdx.x << p.logC;
  const elt_t* src = p.src + (long long)blockIdx.y * p.src_row + (long long)blockIdx.x * p.src_tile;
  elt_t* dst = p.dst + (long long)blockIdx.y * p.dst_row + (long long)blockIdx.x * p.dst_tile;

  for (u32 e = tid; e < T * C; e += FFT_THREADS) {
    u32 k, c;
    if (p.kfast_src) { k = e & (T - 1); c = e >> p.logT; } else { c = e & (C - 1); k = e >> p.logC; }
    elt_t v = elt_zero();
    if (cbase + c < p.nbatch) v = ld16(src + (long long)k * p.sk + (long long)c * p.sc);
    st16(&s[(k << p.logC) + c], v);
  }
  __syncthreads();
  for (u32 step = 0; step < p.logT; ++step) {
    const u32 ii = inverse ? step : (p.logT - 1 - step);
    const u32 sz = 1u << ii;
    for (u32 e = tid; e < (T >> 1) * C; e += FFT_THREADS) {
      u32 c = e & (C - 1), b = e >> p.logC;
      u32 v = b & (sz - 1), u = b >> ii;
      u32 i0 = (u << (ii + 1)) + v, i1 = i0 + sz;
      elt_t tw = ld16(&t.tbl[t.off[ii] + u]);
      if (t.base) {
        u32 cb = cbase + c;
        if (cb >= t.nb) cb = 0;
        tw = gf_add(tw, ld16(&t.base[ii * t.nb + cb]));
      }
      elt_t b0 = ld16(&s[(i0 << p.logC) + c]);
      elt_t b1 = ld16(&s[(i1 << p.logC) + c]);
      if (!inverse) {
        b0 = gf_add(b0, gf_mul(tw, b1));
        b1 = gf_add(b1, b0);
      } else {
        b1 = gf_add(b1, b0);
        b0 = gf_add(b0, gf_mul(tw, b1));
      }
      st16(&s[(i0 << p.logC) + c], b0);
      st16(&s[(i1 << p.logC) + c], b1);
    }
    __syncthreads();
  }
  for (u32 e = tid; e < T * C; e += FFT_THREADS) {
    u32 j, c;
    if (p.kfast_dst) { j = e & (T - 1); c = e >> p.logT; } else { c = e & (C - 1); j = e >> p.logC; }
    if (cbase + c >= p.nbatch) continue;
    st16(dst + (long long)j * p.dk + (long long)c * p.dc, ld16(&s[(j << p.logC) + c]));
  }
}

// ------------------------------------------------------------------ host side
template <class O>
static int set_lds_limit_fp(lfgpu_ctx* c) {
  LF_HIP(c, hipFuncSetAttribute((const void*)fp_fft_tile<O, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)fp_fft_tile<O, 1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)fp_fft_tile<O, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)fp_fft_tile<O, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return LFGPU_OK;
}
static int set_lds_limit(lfgpu_ctx* c) {
  if (!(c->attr_done & 1u)) {
    if (const char* e = getenv("LFGPU_TILE_LOG")) {  // tuning knob: 12 or 13
      int v = atoi(e);
      if (v == 12 || v == 13) c->tile_log = v;
    }
    LF_TRY(set_lds_limit_fp<Fp128Ops>(c));
    LF_TRY(set_lds_limit_fp<F64x2Ops<true>>(c));
    LF_TRY(set_lds_limit_fp<F64x2Ops<false>>(c));
    LF_HIP(c, hipFuncSetAttribute((const void*)lch_fft_tile<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_ELTS * 16));
    LF_HIP(c, hipFuncSetAttribute((const void*)lch_fft_tile<512>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE_ELTS * 16));
    c->attr_done |= 1u;
  }
  return LFGPU_OK;
}

// LDS of one Fp128 tile: the elements, plus the T/2 stage twiddles when both fit (p.wlds)
static size_t fp_lds_bytes(TilePlan& p) {
  const size_t tile = ((size_t)16 << p.logT) << p.logC, tw = (size_t)8 << p.logT;
  p.wlds = tile + tw <= 160u * 1024u ? 1u : 0u;
  return p.wlds ? tile + tw : tile;
}
template <class O, class... Args>
static void launch_fp(lfgpu_ctx* c, dim3 grid, size_t lds, const TilePlan& p, Args... args) {
  if (c->tile_log == 13) {
    if (p.wlds) hipLaunchKernelGGL((fp_fft_tile<O, 1024, true>), grid, dim3(1024), lds, c->stream, p, args...);
    else hipLaunchKernelGGL((fp_fft_tile<O, 1024, false>), grid, dim3(1024), lds, c->stream, p, args...);
  } else {
    if (p.wlds) hipLaunchKernelGGL((fp_fft_tile<O, 512, true>), grid, dim3(512), lds, c->stream, p, args...);
    else hipLaunchKernelGGL((fp_fft_tile<O, 512, false>), grid, dim3(512), lds, c->stream, p, args...);
  }
}
template <class... Args>
static void launch_lch(lfgpu_ctx* c, dim3 grid, size_t lds, Args... args) {
  if (c->tile_log == 13)
    hipLaunchKernelGGL(lch_fft_tile<1024>, grid, dim3(1024), lds, c->stream, args...);
  else
    hipLaunchKernelGGL(lch_fft_tile<512>, grid, dim3(512), lds, c->stream, args...);
}

template <class O>
static elt_t fp_reroot(elt_t w, u64 n, u64 r) {  // twiddle.h:47-55
  while (r < n) {
    w = O::hmul(w, w);
    r += r;
  }
  return w;
}

static std::string keyf(const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return std::string(buf);
}

// single-pass plan: T = n points, batch = rows
static TilePlan plan_single(const lfgpu_ctx* c, void* A, size_t rows, u32 logn, size_t ld) {
  TilePlan p{};
  u32 logC = (u32)c->tile_log - logn;
  u32 need = lf_log2(rows);
  if (logC > need) logC = need;
  p.src = (const elt_t*)A;
  p.dst = (elt_t*)A;
  p.src_row = p.dst_row = 0;
  p.src_tile = p.dst_tile = (long long)ld << logC;
  p.sk = p.dk = 1;
  p.sc = p.dc = (long long)ld;
  p.logT = logn;
  p.logC = logC;
  p.nbatch = (u32)rows;
  p.kfast_src = p.kfast_dst = 1;
  return p;
}

// root table W[i] = w_Tw^i, i < Tw/2, for a transform of 2^logn points with the root wn of that order
template <class O>
static int fp_root_table(lfgpu_ctx* c, const elt_t wn, u32 logn, u32 logTw, std::string* key_out, void** dW) {
  const size_t n = (size_t)1 << logn;
  std::string key = keyf("%sW:%llx:%llx:%u:%u", O::kKey, (u64)wn.lo, (u64)wn.hi, logn, logTw);
  if (!lf_table_lookup(c, key, dW)) {
    std::vector<elt_t> W((size_t)1 << (logTw ? logTw - 1 : 0));
    elt_t wt = fp_reroot<O>(wn, n, (u64)1 << logTw), x = O::one();
    for (size_t i = 0; i < W.size(); ++i) {
      W[i] = x;
      x = O::hmul(x, wt);
    }
    LF_TRY(lf_table(c, key, W.data(), W.size() * 16, dW));
  }
  *key_out = key;
  return LFGPU_OK;
}
// the two-level inter-pass table lo[e & 1023] = wn^(e & 1023), hi[e >> 10] = wn^(1024 (e >> 10)), e < n
template <class O>
static int fp_two_level_tables(lfgpu_ctx* c, const elt_t wn, u32 logn, const std::string& key, void** dlo, void** dhi) {
  std::string klo = key + ":lo", khi = key + ":hi";
  if (!lf_table_lookup(c, klo, dlo) || !lf_table_lookup(c, khi, dhi)) {
    std::vector<elt_t> lo(1024), hi((size_t)1 << (logn > 10 ? logn - 10 : 0));
    elt_t x = O::one();
    for (size_t i = 0; i < 1024; ++i) {
      lo[i] = x;
      x = O::hmul(x, wn);
    }
    elt_t w1024 = x;  // wn^1024
    x = O::one();
    for (size_t i = 0; i < hi.size(); ++i) {
      hi[i] = x;
      x = O::hmul(x, w1024);
    }
    LF_TRY(lf_table(c, klo, lo.data(), lo.size() * 16, dlo));
    LF_TRY(lf_table(c, khi, hi.data(), hi.size() * 16, dhi));
  }
  return LFGPU_OK;
}

// Two tile passes for 2^12 < n <= 2^20: `rows` transforms with the root wn of order n, row r read at src + r*sld and
// written to X[j] = dst[r*drow + j*dstep] (dstep = 1, drow = ld: a plain row).  Goes through `scratch`.
template <class O>
static int fp_fft_two_pass(lfgpu_ctx* c, const elt_t wn, u32 logn, size_t rows, const elt_t* src, size_t sld, elt_t* dst, long long drow,
                           long long dstep) {
  const size_t n = (size_t)1 << logn;
  const u32 logn2 = 10, logn1 = logn - logn2;
  const u32 logTw = logn1 > logn2 ? logn1 : logn2;  // root table covers the larger tile
  void *dW = nullptr, *dlo = nullptr, *dhi = nullptr;
  std::string key;
  LF_TRY(fp_root_table<O>(c, wn, logn, logTw, &key, &dW));
  // Inter-pass twiddles w_n^(j1*k2): either the full [j1][k2] table (n elements, default: one product
  // per element, table slices stay in L2 because batch rows vary fastest in the grid) or, with LFGPU_FP_TW=2, the
  // two-level form lo[e & 1023] * hi[e >> 10] (2 KiB + n/64 bytes of tables, two products per element).
  static const bool two_level = getenv("LFGPU_FP_TW") && atoi(getenv("LFGPU_FP_TW")) == 2;
  const size_t n1 = (size_t)1 << logn1, n2 = (size_t)1 << logn2;
  if (two_level) {
    LF_TRY(fp_two_level_tables<O>(c, wn, logn, key, &dlo, &dhi));
  } else {
    std::string kfull = key + ":full";
    if (!lf_table_lookup(c, kfull, &dlo)) {
      std::vector<elt_t> full(n1 * n2);
      elt_t wj = O::one();  // wn^j1
      for (size_t j = 0; j < n1; ++j) {
        elt_t x = O::one();
        elt_t* row = &full[j * n2];
        for (size_t k = 0; k < n2; ++k) {
          row[k] = x;
          x = O::hmul(x, wj);
        }
        wj = O::hmul(wj, wn);
      }
      LF_TRY(lf_table(c, kfull, full.data(), full.size() * 16, &dlo));
    }
    dhi = nullptr;
  }
  void* scratch = nullptr;
  LF_TRY(lf_scratch(c, rows * n * 16, &scratch));
  {  // pass A: n1-point transforms over k1 (stride n2), C consecutive k2 per tile
    TilePlan p{};
    p.logT = logn1;
    p.logC = (u32)c->tile_log - logn1;
    if (p.logC > logn2) p.logC = logn2;
    p.src = src;
    p.dst = (elt_t*)scratch;
    p.src_row = (long long)sld;
    p.dst_row = (long long)n;
    p.src_tile = p.dst_tile = 1ll << p.logC;
    p.sk = p.dk = (long long)n2;
    p.sc = p.dc = 1;
    p.nbatch = (u32)n2;
    p.kfast_src = p.kfast_dst = 0;
    size_t lds = fp_lds_bytes(p);
    static const bool row_fast_env = !(getenv("LFGPU_FP_ROWFAST") && atoi(getenv("LFGPU_FP_ROWFAST")) == 0);
    if (!two_level && rows <= 65535 && row_fast_env)  // rows fastest: the tile's table slice is reused by every row while it is hot
      launch_fp<O>(c, dim3((u32)rows, (u32)(n2 >> p.logC)), lds, p, (const elt_t*)dW, logTw - logn1, (const elt_t*)dlo, (const elt_t*)dhi, 1u);
    else
      launch_fp<O>(c, dim3((u32)(n2 >> p.logC), (u32)rows), lds, p, (const elt_t*)dW, logTw - logn1, (const elt_t*)dlo, (const elt_t*)dhi, 0u);
    LF_HIP(c, hipGetLastError());
  }
  {  // pass B: n2-point transforms on contiguous rows j1; output X[j1 + n1*j2]
    TilePlan p{};
    p.logT = logn2;
    p.logC = (u32)c->tile_log - logn2;
    if (p.logC > logn1) p.logC = logn1;
    p.src = (const elt_t*)scratch;
    p.dst = dst;
    p.src_row = (long long)n;
    p.dst_row = drow;
    p.src_tile = (long long)n2 << p.logC;
    p.dst_tile = (1ll << p.logC) * dstep;
    p.sk = 1;
    p.sc = (long long)n2;
    p.dk = (long long)n1 * dstep;
    p.dc = dstep;
    p.nbatch = (u32)n1;
    p.kfast_src = 1;
    p.kfast_dst = 0;
    size_t lds = fp_lds_bytes(p);
    launch_fp<O>(c, dim3((u32)(n1 >> p.logC), (u32)rows), lds, p, (const elt_t*)dW, logTw - logn2, (const elt_t*)nullptr,
              (const elt_t*)nullptr, 0u);
    LF_HIP(c, hipGetLastError());
  }
  return LFGPU_OK;
}

// FFT<Field>::fftb / fftf (fft.h:185-201) for the field O describes
template <class O>
static int fft_any(lfgpu_ctx* c, const char* what, int dir, size_t rows, size_t n, const uint64_t omega[2], uint64_t omega_order, void* d_A,
                   size_t ld) {
  if (!c || !omega || (!d_A && rows && n)) return lf_fail(c, LFGPU_ERR_ARG, "%s: null argument", what);
  if (rows == 0 || n <= 1) return LFGPU_OK;
  if (n & (n - 1)) return lf_fail(c, LFGPU_ERR_ARG, "%s: n=%zu is not a power of two", what, n);
  if (omega_order < n || (omega_order & (omega_order - 1)))
    return lf_fail(c, LFGPU_ERR_ARG, "%s: omega_order must be a power of two >= n", what);
  if (ld < n) return lf_fail(c, LFGPU_ERR_ARG, "%s: ld < n", what);
  if (rows > 0x7fffffffu) return lf_fail(c, LFGPU_ERR_ARG, "%s: too many rows", what);
  const u32 logn = lf_log2(n);
  if (logn > 30) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "%s: n > 2^30", what);
  LF_HIP(c, hipSetDevice(c->device));
  LF_TRY(set_lds_limit(c));

  elt_t w{omega[0], omega[1]};
  if (dir == 1) w = O::inv(w);  // fftf = fftb with omega^-1 (fft.h:198-201)
  elt_t wn = fp_reroot<O>(w, omega_order, n);
  if (logn <= (u32)c->tile_log) {  // one pass
    void* dW = nullptr;
    std::string key;
    LF_TRY(fp_root_table<O>(c, wn, logn, logn, &key, &dW));
    TilePlan p = plan_single(c, d_A, rows, logn, ld);
    u32 ntiles = (u32)((rows + (1u << p.logC) - 1) >> p.logC);
    size_t lds = fp_lds_bytes(p);
    launch_fp<O>(c, dim3(ntiles, 1), lds, p, (const elt_t*)dW, 0u, (const elt_t*)nullptr, (const elt_t*)nullptr, 0u);
    LF_HIP(c, hipGetLastError());
    return LFGPU_OK;
  }
  if (logn <= 20) return fp_fft_two_pass<O>(c, wn, logn, rows, (const elt_t*)d_A, ld, (elt_t*)d_A, (long long)ld, 1);
  // n > 2^20: n = n1 * 2^20.  Per row: one tile pass of n1-point transforms down the stride-2^20 dimension, multiplied by
  // the twiddles w_n^(j1 * column) (two-level table), written as n1 contiguous sequences of 2^20 points; those are
  // transformed by the two-pass plan with the root w_n^n1 and written to X[j1 + n1 * J].
  const u32 logn1 = logn - 20;
  const size_t n1 = (size_t)1 << logn1, n2 = (size_t)1 << 20;
  void *dW = nullptr, *dlo = nullptr, *dhi = nullptr, *mid = nullptr;
  std::string key;
  LF_TRY(fp_root_table<O>(c, wn, logn, logn1, &key, &dW));
  LF_TRY(fp_two_level_tables<O>(c, wn, logn, key, &dlo, &dhi));
  LF_TRY(lf_scratch2(c, n * 16, &mid));
  elt_t wi = wn;  // w_n^n1: the root of order 2^20 of the inner transforms
  for (u32 i = 0; i < logn1; ++i) wi = O::hmul(wi, wi);
  for (size_t r = 0; r < rows; ++r) {
    elt_t* row = (elt_t*)d_A + r * ld;
    TilePlan p{};
    p.logT = logn1;
    p.logC = (u32)c->tile_log - logn1;
    p.src = row;
    p.dst = (elt_t*)mid;
    p.src_row = p.dst_row = 0;
    p.src_tile = p.dst_tile = 1ll << p.logC;
    p.sk = p.dk = (long long)n2;
    p.sc = p.dc = 1;
    p.nbatch = (u32)n2;
    p.kfast_src = p.kfast_dst = 0;
    size_t lds = fp_lds_bytes(p);
    launch_fp<O>(c, dim3((u32)(n2 >> p.logC), 1), lds, p, (const elt_t*)dW, 0u, (const elt_t*)dlo, (const elt_t*)dhi, 0u);
    LF_HIP(c, hipGetLastError());
    LF_TRY(fp_fft_two_pass<O>(c, wi, 20, n1, (const elt_t*)mid, n2, row, 1, (long long)n1));
  }
  return LFGPU_OK;
}

extern "C" int lfgpu_fp128_fft(lfgpu_ctx* c, int dir, size_t rows, size_t n, const uint64_t omega[2], uint64_t omega_order, void* d_A,
                               size_t ld) {
  return fft_any<Fp128Ops>(c, "fp128_fft", dir, rows, n, omega, omega_order, d_A, ld);
}
// FFT<Fp2<Fp<1>>>::fftb / fftf over p = 2^64 - 2^32 + 1 (fft_test.cc:205-229).  omega = {re, im}; the usual root is real.
extern "C" int lfgpu_f64_2_fft(lfgpu_ctx* c, int dir, size_t rows, size_t n, const uint64_t omega[2], uint64_t omega_order, void* d_A,
                               size_t ld) {
  if (omega && (omega[0] >= F64_P || omega[1] >= F64_P)) return lf_fail(c, LFGPU_ERR_ARG, "f64_2_fft: omega is not reduced");
  if (omega && omega[1] == 0) return fft_any<F64x2Ops<true>>(c, "f64_2_fft", dir, rows, n, omega, omega_order, d_A, ld);
  return fft_any<F64x2Ops<false>>(c, "f64_2_fft", dir, rows, n, omega, omega_order, d_A, ld);
}

// Build (and cache) LCH14 twiddle tables for one tile pass.
//   with_coset: fold `coset` into tbl (pass A / single pass); otherwise tbl is coset-free
//   and base[ii*nb + blk] = twiddle(i, coset ^ (blk << logT)) carries coset and block bits.
static int lch_tables(lfgpu_ctx* c, const GfHostCtx* g, u32 i_lo, u32 logT, u64 coset, bool with_coset, u32 nb,
                      LchTables* out) {
  std::string key = keyf("lch:%u:%u:%u:%llx:%d:%u", g->k, i_lo, logT, (u64)coset, (int)with_coset, nb);
  void *dt = nullptr, *db = nullptr;
  u32 off = 0;
  for (u32 ii = 0; ii < logT; ++ii) {
    out->off[ii] = off;
    off += 1u << (logT - 1 - ii);
  }
  if (!lf_table_lookup(c, key, &dt)) {
    std::vector<elt_t> tbl(off ? off : 1);
    for (u32 ii = 0; ii < logT; ++ii) {
      u32 i = i_lo + ii, cnt = 1u << (logT - 1 - ii);
      for (u32 u = 0; u < cnt; ++u) {
        u64 x = (u64)u << (i + 1);
        if (with_coset) x ^= coset;
        tbl[out->off[ii] + u] = h_lch14_twiddle(g, i, x);
      }
    }
    LF_TRY(lf_table(c, key, tbl.data(), tbl.size() * 16, &dt));
  }
  out->tbl = (const elt_t*)dt;
  out->base = nullptr;
  out->nb = nb;
  if (!with_coset) {
    std::string kb = key + ":base";
    if (!lf_table_lookup(c, kb, &db)) {
      std::vector<elt_t> base((size_t)logT * nb);
      for (u32 ii = 0; ii < logT; ++ii)
        for (u32 b = 0; b < nb; ++b)
          base[(size_t)ii * nb + b] = h_lch14_twiddle(g, i_lo + ii, coset ^ ((u64)b << (i_lo + logT)));
      LF_TRY(lf_table(c, kb, base.data(), base.size() * 16, &db));
    }
    out->base = (const elt_t*)db;
  }
  return LFGPU_OK;
}

extern "C" int lfgpu_gf2128_lch14_fft(lfgpu_ctx* c, int k, int dir, size_t rows, unsigned l, uint64_t coset,
                                      void* d_B, size_t ld) {
  if (!c || (!d_B && rows)) return lf_fail(c, LFGPU_ERR_ARG, "lch14_fft: null argument");
  const GfHostCtx* g = lf_gf_ctx(c, k);
  if (!g) return lf_fail(c, LFGPU_ERR_ARG, "lch14_fft: subfield_log_bits must be 4 or 5");
  if (l > g->sub_bits) return lf_fail(c, LFGPU_ERR_ARG, "lch14_fft: l <= kSubFieldBits violated (lch14.h:107)");
  if (rows == 0 || l == 0) return LFGPU_OK;
  // the two-pass tile plan splits l = (l - 10) + 10 with a first tile of at most 2^tile_log points: l <= 22; batches of
  // >= 32 rows take the bit-sliced path, whose passes are generic in l (bounded here by its u32 column indices and the
  // 2^l x 128-byte internal buffer per 32 rows)
  if (l > (rows >= 32 ? 24u : 10u + (unsigned)c->tile_log)) return lf_fail(c, LFGPU_ERR_UNSUPPORTED, "lch14_fft: l = %u too large for %zu rows", l, rows);
  if (ld < ((size_t)1 << l)) return lf_fail(c, LFGPU_ERR_ARG, "lch14_fft: ld < 2^l");
  if (rows > 0x7fffffffu) return lf_fail(c, LFGPU_ERR_ARG, "lch14_fft: too many rows");
  LF_HIP(c, hipSetDevice(c->device));
  LF_TRY(set_lds_limit(c));
  const int inverse = dir ? 1 : 0;
  {  // batches of >= 32 rows take the bit-sliced tower path (lch_bs.hip); LFGPU_LCH_BS=0 disables it
    static int bs = -1;
    if (bs < 0) {
      const char* e = getenv("LFGPU_LCH_BS");
      bs = (e && atoi(e) == 0) ? 0 : 1;
    }
    if (bs && rows >= 32 && l >= 7) return lf_lch14_fft_bitsliced(c, k, inverse, rows, l, coset, d_B, ld);
  }
  if (l <= (unsigned)c->tile_log) {
    TilePlan p = plan_single(c, d_B, rows, l, ld);
    LchTables t{};
    LF_TRY(lch_tables(c, g, 0, l, coset, true, 1, &t));
    u32 ntiles = (u32)((rows + (1u << p.logC) - 1) >> p.logC);
    size_t lds = ((size_t)16 << p.logT) << p.logC;
    launch_lch(c, dim3(ntiles, 1), lds, p, inverse, t);
    LF_HIP(c, hipGetLastError());
    return LFGPU_OK;
  }
  const u32 lo = 10, logn1 = l - lo;
  const size_t n1 = (size_t)1 << logn1, n2 = (size_t)1 << lo;
  TilePlan pa{}, pb{};
  LchTables ta{}, tb{};
  // pass A: stages i >= lo over k1 (stride n2); twiddle index u = k1 >> (ii+1): no block term
  pa.logT = logn1;
  pa.logC = (u32)c->tile_log - logn1;
  if (pa.logC > lo) pa.logC = lo;
  pa.src = (const elt_t*)d_B;
  pa.dst = (elt_t*)d_B;
  pa.src_row = pa.dst_row = (long long)ld;
  pa.src_tile = pa.dst_tile = 1ll << pa.logC;
  pa.sk = pa.dk = (long long)n2;
  pa.sc = pa.dc = 1;
  pa.nbatch = (u32)n2;
  pa.kfast_src = pa.kfast_dst = 0;
  LF_TRY(lch_tables(c, g, lo, logn1, coset, true, 1, &ta));
  // pass B: stages i < lo on contiguous blocks k1; C consecutive blocks per tile
  pb.logT = lo;
  pb.logC = (u32)c->tile_log - lo;
  if (pb.logC > logn1) pb.logC = logn1;
  pb.src = (const elt_t*)d_B;
  pb.dst = (elt_t*)d_B;
  pb.src_row = pb.dst_row = (long long)ld;
  pb.src_tile = pb.dst_tile = (long long)n2 << pb.logC;
  pb.sk = pb.dk = 1;
  pb.sc = pb.dc = (long long)n2;
  pb.nbatch = (u32)n1;
  pb.kfast_src = pb.kfast_dst = 1;
  LF_TRY(lch_tables(c, g, 0, lo, coset, false, (u32)n1, &tb));
  size_t lds_a = ((size_t)16 << pa.logT) << pa.logC, lds_b = ((size_t)16 << pb.logT) << pb.logC;
  dim3 ga((u32)(n2 >> pa.logC), (u32)rows), gb((u32)(n1 >> pb.logC), (u32)rows);
  if (!inverse) {  // FFT: stages l-1 .. 0
    launch_lch(c, ga, lds_a, pa, 0, ta);
    launch_lch(c, gb, lds_b, pb, 0, tb);
  } else {  // IFFT: stages 0 .. l-1
    launch_lch(c, gb, lds_b, pb, 1, tb);
    launch_lch(c, ga, lds_a, pa, 1, ta);
  }
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

extern "C" int lfgpu_fp128_fft_host(lfgpu_ctx* c, int dir, size_t n, const uint64_t omega[2], uint64_t omega_order,
                                    void* h_A) {
  if (!c || !h_A) return LFGPU_ERR_ARG;
  void* d = nullptr;
  LF_TRY(lf_scratch2(c, n * 16, &d));
  LF_HIP(c, hipMemcpyAsync(d, h_A, n * 16, hipMemcpyHostToDevice, c->stream));
  LF_TRY(lfgpu_fp128_fft(c, dir, 1, n, omega, omega_order, d, n));
  LF_HIP(c, hipMemcpyAsync(h_A, d, n * 16, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}

extern "C" int lfgpu_f64_2_fft_host(lfgpu_ctx* c, int dir, size_t n, const uint64_t omega[2], uint64_t omega_order, void* h_A) {
  if (!c || !h_A) return LFGPU_ERR_ARG;
  void* d = nullptr;
  LF_TRY(lf_scratch2(c, n * 16, &d));
  LF_HIP(c, hipMemcpyAsync(d, h_A, n * 16, hipMemcpyHostToDevice, c->stream));
  LF_TRY(lfgpu_f64_2_fft(c, dir, 1, n, omega, omega_order, d, n));
  LF_HIP(c, hipMemcpyAsync(h_A, d, n * 16, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}

extern "C" int lfgpu_gf2128_lch14_fft_host(lfgpu_ctx* c, int k, int dir, unsigned l, uint64_t coset, void* h_B) {
  if (!c || !h_B) return LFGPU_ERR_ARG;
  size_t n = (size_t)1 << l;
  void* d = nullptr;
  LF_TRY(lf_scratch2(c, n * 16, &d));
  LF_HIP(c, hipMemcpyAsync(d, h_B, n * 16, hipMemcpyHostToDevice, c->stream));
  LF_TRY(lfgpu_gf2128_lch14_fft(c, k, dir, 1, l, coset, d, n));
  LF_HIP(c, hipMemcpyAsync(h_B, d, n * 16, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}
