// lch_bs.hip -- K2, bit-sliced: batched LCH14 additive FFT over GF(2^128) for >= 32 rows.
//
// Reference semantics: LCH14<GF2_128<k>>::FFT / IFFT (lib/gf2k/lch14.h:106-144), bit-exact.
// Why a second implementation: the per-lane kernel (fft.hip) pays ~510 VALU ops per twiddle
// product because CDNA4 has no carry-less multiply.  Here 32 batch rows share one 32-bit word per
// coordinate bit of the tower basis (bitslice.h), all rows of a column pair share the twiddle, and a
// twiddle product is ~m/2 conditional plane XORs: ~45 (k=4) / ~85 (k=5) ops per butterfly.
//
// Pipeline (all device-resident, "internal" = bit-sliced tower units of m words):
//   bs_cin    : rows (reference Elt layout) -> transpose 32x32 bits -> poly->tower basis -> units
//   bs_bfly2  : one launch per group of <= 4 index bits, a lane keeps its butterfly pair in registers through the group's
//               stages, one wave per column pair (1 / 2 / 4 / 8 waves for groups of 1 / 2 / 3 / 4 bits); 87 % of the pass's
//               memory-only time.  bs_bfly is the round-1 LDS-tile kernel, kept for < 64 combos.
//   bs_cout   : units -> tower->poly basis -> transpose -> rows
// Internal buffer: unit index ((rg*D + q)*stride + c), m words each; 128-byte (k=5) / 64-byte (k=4)
// contiguous chunks in every pass.  The tower representation is also a WORK FORMAT: the Reed-Solomon encoder of large rows
// (rs.hip) converts once, runs its dozen transforms on sub-blocks of a unit buffer (lf_bs_tower_op, end of this file) and
// converts each coset of evaluations out once.
#include <string>

#include "bitslice.h"
#include "ctx.h"

#define BS_COLS 64      // columns per conversion tile (one wave = one plane row)
#define BS_NB_MAX 5     // index bits per butterfly pass (array bound)
// Butterfly tile = 2^r_log (row-group, coordinate) combos x 2^nb columns = 512 units.  r_log 5 / nb 4
// (one stage per pass whose twiddle differs between the two halves of a wave: run as two exec-masked scalar-branched
// products, bs_mac_groups) or r_log 6 / nb 3 (every stage wave-uniform, more passes):
// LFGPU_BS_RLOG selects; see DESIGN.md.
#define BS_PS(units) ((units) + 1)  // LDS plane stride (words): +1 keeps the 8x4-byte scatter of a 128-byte chunk on distinct banks

// ------------------------------------------------------------------ conversions
// Tile = one row group (32 rows) x BS_COLS columns, 256 threads, 32 KiB LDS, <= 256 VGPRs: two independent
// workgroups per CU, so one tile's HBM phase overlaps the other's XOR program (a single 512-thread WG per
// CU ran its phases strictly one after the other).  Plane layout [p][lc ^ ((p>>5)<<3)]: the transpose
// writes of the four dwords of a column land on distinct banks.
#define BS_PL(p, lc) ((p) * BS_COLS + ((lc) ^ ((((u32)(p)) >> 5) << 3)))

// n = columns per combo of the unit buffer `dst` (its stride); valid = columns of `src` that hold data (the rest reads as 0)
template <int K, int WPC>
__global__ __launch_bounds__(256, WPC) void bs_cin_kernel(const elt_t* __restrict__ src, size_t ld, u32 rows, u32 n,
                                                          u32* __restrict__ dst, u32 valid) {
  constexpr int M = Tower<K>::M, D = Tower<K>::D;
  extern __shared__ u32 lds[];  // 32 rows x BS_COLS x 4 words, then 128 planes x BS_COLS words
  const u32 t = threadIdx.x, rg = blockIdx.y, c0 = blockIdx.x * BS_COLS;
  {  // phase 1: coalesced row reads (16 B per lane, 1 KiB contiguous per row), all issued before the LDS writes
    const u32 lc = t & (BS_COLS - 1), rq = t / BS_COLS;
    const u32 rows_eff = c0 + lc < valid ? rows : 0u;  // columns past the data read as zero: same loop as without the bound
    uint4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32 row = rg * 32 + rq * 8 + i;
      v[i] = make_uint4(0, 0, 0, 0);
      if (row < rows_eff) v[i] = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + c0 + lc);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(&lds[((rq * 8 + i) * BS_COLS + lc) * 4]) = v[i];
  }
  __syncthreads();
  {  // phase 2: per (column, dword) 32x32 bit transpose
    const u32 lc = t >> 2, w = t & 3;
    u32 x[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = lds[(r * BS_COLS + lc) * 4 + w];
    __syncthreads();
    bs_transpose32(x);
    const u32 lcs = lc ^ (w << 3);
#pragma unroll
    for (int b = 0; b < 32; ++b) lds[(w * 32 + b) * BS_COLS + lcs] = x[b];
  }
  __syncthreads();
  {  // phase 3: polynomial basis -> tower basis, one coordinate per wave (wave-uniform q)
    const u32 lc = t & (BS_COLS - 1), q0 = t / BS_COLS;
#define BS_IN(i) lds[BS_PL(i, lc)]
#define BS_OUT(o) out[o]
    if (K == 5 && WPC >= 4) {
      // 8 outputs per program (a quarter of a coordinate): +14 % XORs over the 32-output programs, but the live set fits
      // 128 VGPRs without spills -> 4 waves per SIMD (the 16-output halves need 191)
      const u32 q = q0;
#define BS_CIN_Q(O)                                                                                             \
  {                                                                                                             \
    u32 out[8];                                                                                                 \
    switch (q) {                                                                                                \
      case 0: TOWER_K5_P2T_Q0O##O(BS_IN, BS_OUT); break;                                                        \
      case 1: TOWER_K5_P2T_Q1O##O(BS_IN, BS_OUT); break;                                                        \
      case 2: TOWER_K5_P2T_Q2O##O(BS_IN, BS_OUT); break;                                                        \
      default: TOWER_K5_P2T_Q3O##O(BS_IN, BS_OUT); break;                                                       \
    }                                                                                                           \
    uint4* u = reinterpret_cast<uint4*>(dst + ((size_t)(rg * D + q) * n + c0 + lc) * M + 8 * O);                \
    u[0] = make_uint4(out[0], out[1], out[2], out[3]);                                                          \
    u[1] = make_uint4(out[4], out[5], out[6], out[7]);                                                          \
  }
      BS_CIN_Q(0)
      BS_CIN_Q(1)
      BS_CIN_Q(2)
      BS_CIN_Q(3)
#undef BS_CIN_Q
    } else if (K == 5) {
      // 16 outputs per program (half a coordinate): +4 % XORs over the 32-output programs, about half the live
      // registers; a wave does both halves of its coordinate one after the other
      const u32 q = q0;  // D == 4 == waves per workgroup
      for (u32 hh = 0; hh < 2; ++hh) {
        u32 out[16];
        switch (2 * q + hh) {
          case 0: TOWER_K5_P2T_Q0H0(BS_IN, BS_OUT); break;
          case 1: TOWER_K5_P2T_Q0H1(BS_IN, BS_OUT); break;
          case 2: TOWER_K5_P2T_Q1H0(BS_IN, BS_OUT); break;
          case 3: TOWER_K5_P2T_Q1H1(BS_IN, BS_OUT); break;
          case 4: TOWER_K5_P2T_Q2H0(BS_IN, BS_OUT); break;
          case 5: TOWER_K5_P2T_Q2H1(BS_IN, BS_OUT); break;
          case 6: TOWER_K5_P2T_Q3H0(BS_IN, BS_OUT); break;
          default: TOWER_K5_P2T_Q3H1(BS_IN, BS_OUT); break;
        }
        uint4* u = reinterpret_cast<uint4*>(dst + ((size_t)(rg * D + q) * n + c0 + lc) * M + 16 * hh);
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = make_uint4(out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]);
      }
    } else {
      for (u32 q = q0; q < (u32)D; q += 256 / BS_COLS) {
        u32 out[M];
        switch (q) {
          case 0: TOWER_K4_P2T_Q0(BS_IN, BS_OUT); break;
          case 1: TOWER_K4_P2T_Q1(BS_IN, BS_OUT); break;
          case 2: TOWER_K4_P2T_Q2(BS_IN, BS_OUT); break;
          case 3: TOWER_K4_P2T_Q3(BS_IN, BS_OUT); break;
          case 4: TOWER_K4_P2T_Q4(BS_IN, BS_OUT); break;
          case 5: TOWER_K4_P2T_Q5(BS_IN, BS_OUT); break;
          case 6: TOWER_K4_P2T_Q6(BS_IN, BS_OUT); break;
          default: TOWER_K4_P2T_Q7(BS_IN, BS_OUT); break;
        }
        uint4* u = reinterpret_cast<uint4*>(dst + ((size_t)(rg * D + q) * n + c0 + lc) * M);
#pragma unroll
        for (int j = 0; j < M / 4; ++j) u[j] = make_uint4(out[4 * j], out[4 * j + 1], out[4 * j + 2], out[4 * j + 3]);
      }
    }
#undef BS_IN
#undef BS_OUT
  }
}

// n = columns per combo of the unit buffer `src`; only the columns [out_lo, out_hi) of the tile range are written to dst
template <int K, int WPC>
__global__ __launch_bounds__(256, WPC) void bs_cout_kernel(const u32* __restrict__ src, size_t ld, u32 rows, u32 n,
                                                         elt_t* __restrict__ dst, u32 out_lo, u32 out_hi) {
  constexpr int M = Tower<K>::M, D = Tower<K>::D;
  extern __shared__ u32 lds[];
  const u32 t = threadIdx.x, rg = blockIdx.y, c0 = blockIdx.x * BS_COLS;
  {  // units -> tower planes [p][lc]; all global loads first
    const u32 lc = t & (BS_COLS - 1), q0 = t / BS_COLS;
    constexpr int QN = D * BS_COLS / 256;  // coordinates per thread
    uint4 v[QN][M / 4];
#pragma unroll
    for (int qi = 0; qi < QN; ++qi) {
      const u32 q = q0 + qi * (256 / BS_COLS);
      const uint4* u = reinterpret_cast<const uint4*>(src + ((size_t)(rg * D + q) * n + c0 + lc) * M);
#pragma unroll
      for (int j = 0; j < M / 4; ++j) v[qi][j] = u[j];
    }
#pragma unroll
    for (int qi = 0; qi < QN; ++qi) {
      const u32 q = q0 + qi * (256 / BS_COLS);
#pragma unroll
      for (int j = 0; j < M / 4; ++j) {
        lds[BS_PL(q * M + 4 * j + 0, lc)] = v[qi][j].x;
        lds[BS_PL(q * M + 4 * j + 1, lc)] = v[qi][j].y;
        lds[BS_PL(q * M + 4 * j + 2, lc)] = v[qi][j].z;
        lds[BS_PL(q * M + 4 * j + 3, lc)] = v[qi][j].w;
      }
    }
  }
  __syncthreads();
  {  // tower -> poly for one dword of the element (wave-uniform w), then 32x32 transpose back to rows
    const u32 lc = t & (BS_COLS - 1), w = t / BS_COLS;
    u32 x[32];
#define BS_IN(i) lds[BS_PL(i, lc)]
#define BS_OUT(o) x[o]
    if (K == 5 && WPC >= 4) {
#undef BS_OUT
#define BS_COUT_Q(O)                                     \
  switch (w) {                                           \
    case 0: TOWER_K5_T2P_W0O##O(BS_IN, BS_OUT); break;   \
    case 1: TOWER_K5_T2P_W1O##O(BS_IN, BS_OUT); break;   \
    case 2: TOWER_K5_T2P_W2O##O(BS_IN, BS_OUT); break;   \
    default: TOWER_K5_T2P_W3O##O(BS_IN, BS_OUT); break;  \
  }
#define BS_OUT(o) x[(o)]
      BS_COUT_Q(0)
#undef BS_OUT
#define BS_OUT(o) x[8 + (o)]
      BS_COUT_Q(1)
#undef BS_OUT
#define BS_OUT(o) x[16 + (o)]
      BS_COUT_Q(2)
#undef BS_OUT
#define BS_OUT(o) x[24 + (o)]
      BS_COUT_Q(3)
#undef BS_OUT
#define BS_OUT(o) x[o]
#undef BS_COUT_Q
    } else if (K == 5) {
      switch (w) {
        case 0: TOWER_K5_T2P_W0(BS_IN, BS_OUT); break;
        case 1: TOWER_K5_T2P_W1(BS_IN, BS_OUT); break;
        case 2: TOWER_K5_T2P_W2(BS_IN, BS_OUT); break;
        default: TOWER_K5_T2P_W3(BS_IN, BS_OUT); break;
      }
    } else {
      switch (w) {
        case 0: TOWER_K4_T2P_W0(BS_IN, BS_OUT); break;
        case 1: TOWER_K4_T2P_W1(BS_IN, BS_OUT); break;
        case 2: TOWER_K4_T2P_W2(BS_IN, BS_OUT); break;
        default: TOWER_K4_T2P_W3(BS_IN, BS_OUT); break;
      }
    }
#undef BS_IN
#undef BS_OUT
    __syncthreads();
    bs_transpose32(x);
#pragma unroll
    for (int r = 0; r < 32; ++r) lds[(r * BS_COLS + lc) * 4 + w] = x[r];
  }
  __syncthreads();
  {
    const u32 lc = t & (BS_COLS - 1), rq = t / BS_COLS;
    const u32 rows_eff = (c0 + lc >= out_lo && c0 + lc < out_hi) ? rows : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const u32 r = rq * 8 + i, row = rg * 32 + r;
      if (row < rows_eff) *reinterpret_cast<uint4*>(dst + (size_t)row * ld + c0 + lc) = *reinterpret_cast<const uint4*>(&lds[(r * BS_COLS + lc) * 4]);
    }
  }
}

// ------------------------------------------------------------------ butterflies
struct BflyArgs {
  u32* data;
  long long src_off;  // words from `data` to where the pass READS its units (0: in place; else another buffer with the same geometry)
  u32 valid;          // columns >= valid of `src` read as zero (the zero-padded coefficient vector of a further coset)
  const u32* tw;      // stage tables, coset folded in: tw[off[b] + u_glob]
  u32 off[BS_NB_MAX];
  u32 n;              // columns per combo (2^l)
  u32 stride;         // columns per combo in the unit buffer (== n for a whole transform; larger when `data` points at a sub-block)
  u32 lo_bit, nb;     // this pass covers index bits [lo_bit, lo_bit + nb)
  u32 r_log;          // log2(lanes per column pair) = log2(combos per tile) + cu
  u32 cu;             // low column bits kept inside the tile (2^cu consecutive columns: 2^cu * 4m-byte segments)
  int inverse;
  u32 probe;  // timing experiments only (LFGPU_BS_PROBE): 1 = no butterflies (memory only), 2 = every tile aliases tile 0 (compute only)
};

template <int K, bool INV>
__global__ __launch_bounds__(256, 2) void bs_bfly_kernel(BflyArgs a) {
  constexpr int M = Tower<K>::M;
  constexpr u32 MU = Tower<K>::MU_LOW;
  const u32 RL = a.r_log, R = 1u << RL;
  extern __shared__ u32 lds[];
  const u32 tid = threadIdx.x;
  const u32 ncol = 1u << a.nb, units = R << a.nb, PS = BS_PS(units);
  const u32 CU = a.cu, cumask = (1u << CU) - 1;
  const u32 tc = a.probe == 2 ? 0u : blockIdx.x, cb0 = (a.probe == 2 ? 0u : blockIdx.y) << (RL - CU);
  const u32 tc_lo = tc & ((1u << (a.lo_bit - CU)) - 1), tc_hi = tc >> (a.lo_bit - CU);
  const u32 cbase = (tc_hi << (a.lo_bit + a.nb)) | (tc_lo << CU);
  constexpr u32 PIECES = M / 4;  // 16-byte pieces per unit
  // load: unit s = j*R + cl, cl = (combo offset << cu) | inner column.  All global loads of the thread are
  // issued before the first LDS write (one exposed HBM latency per tile instead of one per piece).
  constexpr u32 MAXIT = (512 * PIECES) / 256;
  const u32 total = units * PIECES;
  {
    uint4 v[MAXIT];
#pragma unroll
    for (u32 it = 0; it < MAXIT; ++it) {
      const u32 e = tid + it * 256;
      const u32 s = e / PIECES, p4 = e % PIECES;
      const u32 j = s >> RL, cl = s & (R - 1);
      v[it] = make_uint4(0, 0, 0, 0);
      if (e < total && cbase + (j << a.lo_bit) + (cl & cumask) < a.valid)
        v[it] = *reinterpret_cast<const uint4*>(a.data + a.src_off + ((size_t)(cb0 + (cl >> CU)) * a.stride + cbase + ((size_t)j << a.lo_bit) + (cl & cumask)) * M + 4 * p4);
    }
#pragma unroll
    for (u32 it = 0; it < MAXIT; ++it) {
      const u32 e = tid + it * 256;
      const u32 s = e / PIECES, p4 = e % PIECES;
      if (e < total) {
        lds[(4 * p4 + 0) * PS + s] = v[it].x;
        lds[(4 * p4 + 1) * PS + s] = v[it].y;
        lds[(4 * p4 + 2) * PS + s] = v[it].z;
        lds[(4 * p4 + 3) * PS + s] = v[it].w;
      }
    }
  }
  // this tile's twiddles for every stage of the pass (prefetched: one latency, not one per stage)
  u32 twv[BS_NB_MAX];
#pragma unroll
  for (u32 b = 0; b < BS_NB_MAX; ++b) {
    twv[b] = 0;
    // threads beyond the R << (nb - 1) butterflies of a stage hold no pair: clamp their pair index, or a short last group
    // (nb < 4) reads past the end of the stage table (a fault at l = 17, 18 where the table ends on a page boundary)
    if (b < a.nb) twv[b] = a.tw[a.off[b] + ((tc_hi << (a.nb - 1 - b)) | (((tid >> RL) & ((1u << (a.nb - 1)) - 1u)) >> b))];
  }
  __syncthreads();
  const u32 ntask = R << (a.nb - 1);
  for (u32 step = 0; step < (a.probe == 1 ? 0u : a.nb); ++step) {
    const u32 b = INV ? step : (a.nb - 1 - step);
    if (tid < ntask) {
      const u32 cl = tid & (R - 1), pv = tid >> RL;
      const u32 v = pv & ((1u << b) - 1), u = pv >> b;
      const u32 j0 = (u << (b + 1)) | v, j1 = j0 + (1u << b);
      const u32 s0 = (j0 << RL) + cl, s1 = (j1 << RL) + cl;
      // global twiddle index: (column >> (i+1)), i = lo_bit + b
      const u32 tw = b == 0 ? twv[0] : b == 1 ? twv[1] : b == 2 ? twv[2] : twv[3];  // (column >> (i+1)) table entry, i = lo_bit + b
      u32 b0[M], b1[M];
#pragma unroll
      for (int p = 0; p < M; ++p) {
        b0[p] = lds[p * PS + s0];
        b1[p] = lds[p * PS + s1];
      }
      // lanes sharing a twiddle: aligned groups of min(64, R << b); one scalar-branched MAC per group
      const u32 glog = RL + b < 6 ? RL + b : 6;
      if (INV) {  // b1 ^= b0; b0 ^= tw*b1   (lch14.h:225-229)
#pragma unroll
        for (int p = 0; p < M; ++p) b1[p] ^= b0[p];
      }
      bs_mac_groups<M, MU>(tw, glog, b1, b0);  // b0 ^= tw*b1
      if (!INV) {  // ... b1 ^= b0   (lch14.h:219-223)
#pragma unroll
        for (int p = 0; p < M; ++p) b1[p] ^= b0[p];
      }
#pragma unroll
      for (int p = 0; p < M; ++p) {
        lds[p * PS + s0] = b0[p];
        lds[p * PS + s1] = b1[p];
      }
    }
    __syncthreads();
  }
  for (u32 e = tid; e < units * PIECES; e += 256) {
    const u32 s = e / PIECES, p4 = e % PIECES;
    const u32 j = s >> RL, cl = s & (R - 1);
    uint4 v;
    v.x = lds[(4 * p4 + 0) * PS + s];
    v.y = lds[(4 * p4 + 1) * PS + s];
    v.z = lds[(4 * p4 + 2) * PS + s];
    v.w = lds[(4 * p4 + 3) * PS + s];
    *reinterpret_cast<uint4*>(a.data + ((size_t)(cb0 + (cl >> CU)) * a.stride + cbase + ((size_t)j << a.lo_bit) + (cl & cumask)) * M + 4 * p4) = v;
  }
}

// ------------------------------------------------------------------ butterflies, register-resident (v2)
// Measured on the LDS-tile kernel above (profiles/r02: probes + PMC): a pass is COMPUTE-bound -- 8.35 ms with every tile
// aliased to tile 0 (no HBM traffic) against 6.45 ms with the butterflies skipped -- because its 233 VGPRs and its
// 64 KiB tile allow 2 waves per SIMD, and one or two waves in their XOR phase issue at 4-5 cycles per instruction where
// four issue at ~2 (tools/ubench: v_xor_b32 28 / 53 / 66 T lane-ops/s at 1 / 2 / 4 waves per SIMD).
// Here a lane keeps its butterfly pair (2 units = 2 x M words) in registers through all nb stages of a pass:
//   * lanes of a wave = 64 (combo, inner column) slots that share every twiddle of the pass -> all stages are wave-uniform:
//     scalar branches on the twiddle's bits, no masked passes;
//   * wave w of the workgroup owns pair w of the tile's 2^nb columns; between two stages exactly one of its two units
//     changes owner (the constant-geometry exchange w <-> w ^ 2^s), so a stage moves ONE unit per lane through LDS
//     (M words out, M words in) instead of reading and writing both, and the tile never lives in LDS;
//   * the product is evaluated in Horner form, acc = acc * h ^ (t_k ? b : 0): no `cur` copy of the operand -> ~110 VGPRs,
//     4 waves per SIMD (two 512-thread workgroups per CU with 64 KiB of exchange buffer each);
//   * units go straight between HBM and registers (128 contiguous bytes per lane).
template <int M, u32 MU_LOW>
__device__ __forceinline__ void bs_mul_horner(u32 t, const u32 (&b)[M], u32 (&acc)[M]) {  // acc = t * b, t wave-uniform
#pragma unroll
  for (int j = 0; j < M; ++j) acc[j] = 0;
#pragma unroll
  for (int k = M - 1; k >= 0; --k) {
    if (k != M - 1) bs_mulh<M, MU_LOW>(acc);
    if ((t >> k) & 1u) {
#pragma unroll
      for (int j = 0; j < M; ++j) acc[j] ^= b[j];
    }
  }
}

// NW = waves per workgroup = column pairs per tile: 8 (<= 4 index bits per pass, two workgroups per CU) or 16 (5 bits,
// one 1024-thread workgroup per CU); either way 16 waves per CU
template <int K, bool INV, int NW>
__global__ __launch_bounds__(64 * NW, 4) void bs_bfly2_kernel(BflyArgs a) {
  constexpr int M = Tower<K>::M;
  constexpr u32 MU = Tower<K>::MU_LOW;
  extern __shared__ u32 xch[];  // [NW waves][M planes][64 + 2 lanes]
  const u32 tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const u32 CU = a.cu, cumask = (1u << CU) - 1;
  const u32 tc = blockIdx.x, cb0 = blockIdx.y << (6 - CU);
  const u32 tc_lo = tc & ((1u << (a.lo_bit - CU)) - 1), tc_hi = tc >> (a.lo_bit - CU);
  const u32 cbase = (tc_hi << (a.lo_bit + a.nb)) | (tc_lo << CU);
  const u32 npair = 1u << (a.nb - 1);
  const bool active = w < npair;
  u32 b0[M], b1[M];
  u32 j0 = 0, j1 = 0;
  {
    const u32 b = INV ? 0u : a.nb - 1, u = w >> b, v = w & ((1u << b) - 1);
    j0 = (u << (b + 1)) | v;
    j1 = j0 + (1u << b);
  }
  // A unit is 128 (K = 5) / 64 (K = 4) contiguous bytes, so HBM wants 8 / 4 lanes per unit: a wave instruction moves whole
  // 128-byte lines (8 or 16 units).  The wave's own exchange region doubles as the transposition buffer between that
  // layout and "lane = slot, registers = planes": plane stride 64 + 2 words puts the 64 lanes of a transposing write on
  // 64 different banks (bank = 8 * piece + 2 * k + slot mod 64), the per-lane plane reads are conflict-free anyway.
  constexpr u32 PIECES = M / 4, UPI = 64 / PIECES, NIT = 64 / UPI, XS = 66;
  u32* const stage = xch + (size_t)w * M * XS;
  const u32 piece = lane % PIECES, sub = lane / PIECES;
  auto slot_base = [&](u32 slot) { return a.data + ((size_t)(cb0 + (slot >> CU)) * a.stride + cbase + (slot & cumask)) * M; };
  auto unit_in = [&](u32 j, u32 (&dst)[M]) {
    uint4 v[NIT];
    if (active) {
#pragma unroll
      for (u32 it = 0; it < NIT; ++it) {
        const u32 slot = it * UPI + sub;
        v[it] = *reinterpret_cast<const uint4*>(slot_base(slot) + a.src_off + ((size_t)j << a.lo_bit) * M + 4 * piece);  // always in bounds: same geometry
      }
#pragma unroll
      for (u32 it = 0; it < NIT; ++it)  // selects after the loads: the loads of a unit stay one batch
        if (cbase + (j << a.lo_bit) + ((it * UPI + sub) & cumask) >= a.valid) v[it] = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (u32 it = 0; it < NIT; ++it) {
        u32* q = stage + (4 * piece) * XS + it * UPI + sub;
        q[0] = v[it].x; q[XS] = v[it].y; q[2 * XS] = v[it].z; q[3 * XS] = v[it].w;
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int p = 0; p < M; ++p) dst[p] = stage[p * XS + lane];
    }
    __syncthreads();
  };
  auto unit_out = [&](u32 j, const u32 (&src)[M]) {
    if (active) {
#pragma unroll
      for (int p = 0; p < M; ++p) stage[p * XS + lane] = src[p];
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (u32 it = 0; it < NIT; ++it) {
        const u32* q = stage + (4 * piece) * XS + it * UPI + sub;
        *reinterpret_cast<uint4*>(slot_base(it * UPI + sub) + ((size_t)j << a.lo_bit) * M + 4 * piece) = make_uint4(q[0], q[XS], q[2 * XS], q[3 * XS]);
      }
    }
    __syncthreads();
  };
  unit_in(j0, b0);
  unit_in(j1, b1);
  for (u32 step = 0; step < a.nb; ++step) {
    const u32 b = INV ? step : (a.nb - 1 - step);
    if (active) {
      // twiddle of (stage lo_bit + b, column block): the same for the whole wave
      const u32 t = __builtin_amdgcn_readfirstlane(a.tw[a.off[b] + ((tc_hi << (a.nb - 1 - b)) | (w >> b))]);
      u32 acc[M];
      if (INV) {  // b1 ^= b0; b0 ^= t*b1   (lch14.h:225-229)
#pragma unroll
        for (int p = 0; p < M; ++p) b1[p] ^= b0[p];
        bs_mul_horner<M, MU>(t, b1, acc);
#pragma unroll
        for (int p = 0; p < M; ++p) b0[p] ^= acc[p];
      } else {  // b0 ^= t*b1; b1 ^= b0   (lch14.h:219-223)
        bs_mul_horner<M, MU>(t, b1, acc);
#pragma unroll
        for (int p = 0; p < M; ++p) {
          b0[p] ^= acc[p];
          b1[p] ^= b0[p];
        }
      }
    }
    if (step + 1 < a.nb) {
      // next stage pairs columns that differ in bit sb' ; my two columns agree in that bit (= s): the unit that keeps
      // its slot stays, the other one is swapped with wave w ^ 2^sb
      const u32 sb = INV ? b : b - 1;
      const u32 s = (w >> sb) & 1u, partner = w ^ (1u << sb);
      __syncthreads();  // everybody has read the previous exchange
      if (active) {
        u32* out = xch + ((size_t)partner * M) * XS + lane;
        if (s == 0) {
#pragma unroll
          for (int p = 0; p < M; ++p) out[p * XS] = b1[p];
        } else {
#pragma unroll
          for (int p = 0; p < M; ++p) out[p * XS] = b0[p];
        }
      }
      __syncthreads();
      if (active) {
        const u32* in = xch + ((size_t)w * M) * XS + lane;
        if (s == 0) {
#pragma unroll
          for (int p = 0; p < M; ++p) b1[p] = in[p * XS];
        } else {
#pragma unroll
          for (int p = 0; p < M; ++p) b0[p] = in[p * XS];
        }
      }
      // the columns this wave now holds
      const u32 nbit = INV ? b + 1 : b - 1;
      const u32 u = w >> nbit, v = w & ((1u << nbit) - 1);
      j0 = (u << (nbit + 1)) | v;
      j1 = j0 + (1u << nbit);
    }
  }
  __syncthreads();  // the last exchange has been read everywhere: the regions are free for the way out
  unit_out(j0, b0);
  unit_out(j1, b1);
}

// ------------------------------------------------------------------ host
template <int K>
static int bs_tables(lfgpu_ctx* c, const GfHostCtx* g, unsigned l, u64 coset, const u32** d_tw, std::vector<u32>* offs) {
  // tw_i[u] = tower bits of twiddle(i, coset ^ (u << (i+1))), i = 0..l-1; linear in the bits of the argument
  char kb[96];
  snprintf(kb, sizeof(kb), "bstw:%d:%u:%llx", K, l, (u64)coset);
  std::string key(kb);
  offs->assign(l, 0);
  u32 off = 0;
  for (unsigned i = 0; i < l; ++i) {
    (*offs)[i] = off;
    off += 1u << (l - 1 - i);
  }
  void* d = nullptr;
  if (!lf_table_lookup(c, key, &d)) {
    std::vector<u32> tbl(off ? off : 1);
    for (unsigned i = 0; i < l; ++i) {
      elt_t t0 = h_lch14_twiddle(g, i, coset);
      u32* t = &tbl[(*offs)[i]];
      t[0] = tower_twiddle_bits<K>(t0.lo, t0.hi);
      for (unsigned kk = 0; (i + 1) + kk < l; ++kk) {
        elt_t sh = g->w_hat[i][(i + 1) + kk];
        u32 sb = tower_twiddle_bits<K>(sh.lo, sh.hi);
        for (u32 u = 0; u < (1u << kk); ++u) t[u + (1u << kk)] = t[u] ^ sb;
      }
    }
    LF_TRY(lf_table(c, key, tbl.data(), tbl.size() * 4, &d));
  }
  *d_tw = (const u32*)d;
  return LFGPU_OK;
}

// shared launch configuration of the bit-sliced kernels (once per context)
static int bs_setup(lfgpu_ctx* c) {
  if (c->attr_done & 4u) return LFGPU_OK;
  if (const char* e = getenv("LFGPU_BS_RLOG")) {
    int v = atoi(e);
    if (v >= 5 && v <= 7) c->bs_rlog = (u32)v;
  }
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly_kernel<5, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly_kernel<5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<4, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<5, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<4, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<5, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 68 * 1024));
  // NW = 1, 2, 4 (groups of 1, 2, 3 index bits: as many waves as column pairs) need <= 34 KiB: under the default limit
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<4, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<5, false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<4, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
  LF_HIP(c, hipFuncSetAttribute((const void*)bs_bfly2_kernel<5, true, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
  c->attr_done |= 4u;
  return LFGPU_OK;
}

// geometry of the unit buffer for `rows` batch rows: row groups, and combos padded to whole butterfly tiles
template <int K>
struct BsGeom {
  bool v2;
  u32 R, nbmax, nrg, combos, rlog;
};
template <int K>
static BsGeom<K> bs_geom(lfgpu_ctx* c, size_t rows) {
  constexpr int D = Tower<K>::D;
  static const int v2_env = getenv("LFGPU_BS_V2") ? atoi(getenv("LFGPU_BS_V2")) : 1;
  // 4 index bits per pass (two 512-thread workgroups per CU overlap their HBM and XOR phases): 61.0 ms per 2^20 x 1024 batch;
  // 5 bits (LFGPU_BS_V2_NB=5: one 1024-thread workgroup per CU, 4 passes instead of 5) measured 64.1 ms
  static const u32 v2_nb = getenv("LFGPU_BS_V2_NB") && atoi(getenv("LFGPU_BS_V2_NB")) == 5 ? 5u : 4u;
  BsGeom<K> gm;
  gm.nrg = (u32)((rows + 31) / 32);
  gm.v2 = v2_env && gm.nrg * D >= 64;  // register-resident butterflies: lanes = 64 (combo, inner column) slots
  gm.rlog = gm.v2 ? 6u : c->bs_rlog;
  gm.R = 1u << gm.rlog;
  gm.nbmax = gm.v2 ? v2_nb : 9 - c->bs_rlog;
  gm.combos = ((gm.nrg * D + gm.R - 1) / gm.R) * gm.R;
  return gm;
}

// the butterfly passes of one transform of 2^l columns on the unit buffer `units` (pointing at the transform's first
// column; `stride` columns per combo): bit groups of <= nbmax index bits; FFT walks stages l-1..0, IFFT 0..l-1
template <int K>
// src != nullptr: the FIRST pass reads its units from `src` (same geometry, columns >= valid as zero) and writes `units`
static int bs_passes(lfgpu_ctx* c, const GfHostCtx* g, const BsGeom<K>& gm, int inverse, unsigned l, u64 coset, u32* units, u32 stride,
                     const u32* src = nullptr, u32 valid = 0xffffffffu) {
  constexpr int M = Tower<K>::M;
  static const u32 cu_env = [] {
    const char* e = getenv("LFGPU_BS_CU");
    return e && (u32)atoi(e) <= 5 ? (u32)atoi(e) : 3u;
  }();
  const u32 n = 1u << l;
  const u32* d_tw = nullptr;
  std::vector<u32> offs;
  LF_TRY(bs_tables<K>(c, g, l, coset, &d_tw, &offs));
  std::vector<std::pair<u32, u32>> groups;  // (lo_bit, nb), ascending
  for (u32 lo = 0; lo < l; lo += gm.nbmax) groups.push_back({lo, std::min<u32>(gm.nbmax, l - lo)});
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    const auto& gr = inverse ? groups[gi] : groups[groups.size() - 1 - gi];
    BflyArgs a{};
    a.data = units;
    a.src_off = gi == 0 && src ? (long long)(src - units) : 0;
    a.valid = gi == 0 && src ? valid : 0xffffffffu;
    a.tw = d_tw;
    for (u32 b = 0; b < gr.second; ++b) a.off[b] = offs[gr.first + b];
    a.n = n;
    a.stride = stride;
    a.lo_bit = gr.first;
    a.nb = gr.second;
    a.r_log = gm.rlog;
    a.cu = std::min(std::min(cu_env, gr.first), a.r_log);  // inner bits must lie below the stage bits
    a.inverse = inverse;
    static const u32 probe_env = getenv("LFGPU_BS_PROBE") ? (u32)atoi(getenv("LFGPU_BS_PROBE")) : 0u;
    a.probe = probe_env;
    const u32 units_per_tile = gm.R << gr.second;
    const dim3 grid(n >> (gr.second + a.cu), gm.combos >> (a.r_log - a.cu));
    if (gm.v2 && gr.second == 5) {  // 16 waves x M planes x (64 + 2) words of exchange / staging buffer
      if (inverse)
        hipLaunchKernelGGL((bs_bfly2_kernel<K, true, 16>), grid, dim3(1024), (size_t)16 * M * 66 * 4, c->stream, a);
      else
        hipLaunchKernelGGL((bs_bfly2_kernel<K, false, 16>), grid, dim3(1024), (size_t)16 * M * 66 * 4, c->stream, a);
    } else if (gm.v2) {
      // one wave per column pair of the tile: a short group (the tail of l mod 4 bits, the small blocks of the truncated
      // Reed-Solomon transform) launched with 8 waves left 7 / 6 / 4 of them idle and cost 1.9x / 1.4x a full pass
      static const int match_nw = getenv("LFGPU_BS_NW_MATCH") ? atoi(getenv("LFGPU_BS_NW_MATCH")) : 1;
      const u32 nw = match_nw ? 1u << (gr.second - 1) : 8u;
#define BS_LAUNCH2(NWV)                                                                                                              \
  do {                                                                                                                               \
    if (inverse)                                                                                                                     \
      hipLaunchKernelGGL((bs_bfly2_kernel<K, true, NWV>), grid, dim3(64 * NWV), (size_t)NWV * M * 66 * 4, c->stream, a);            \
    else                                                                                                                             \
      hipLaunchKernelGGL((bs_bfly2_kernel<K, false, NWV>), grid, dim3(64 * NWV), (size_t)NWV * M * 66 * 4, c->stream, a);           \
  } while (0)
      if (nw == 1) BS_LAUNCH2(1);
      else if (nw == 2) BS_LAUNCH2(2);
      else if (nw == 4) BS_LAUNCH2(4);
      else BS_LAUNCH2(8);
#undef BS_LAUNCH2
    } else if (inverse)
      hipLaunchKernelGGL((bs_bfly_kernel<K, true>), grid, dim3(256), (size_t)M * BS_PS(units_per_tile) * 4, c->stream, a);
    else
      hipLaunchKernelGGL((bs_bfly_kernel<K, false>), grid, dim3(256), (size_t)M * BS_PS(units_per_tile) * 4, c->stream, a);
  }
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

// rows (reference layout; columns >= valid read as 0) -> units, `ncols` columns (a multiple of 64) from the buffer's first column
template <int K>
static int bs_convert_in(lfgpu_ctx* c, const BsGeom<K>& gm, size_t rows, u32 ncols, u32 valid, const elt_t* src, size_t ld, u32* units, u32 stride) {
  // GF2_128<5>: 8-output basis-change programs at 4 waves per SIMD (12.6 -> 10.4 ms per 2^30 elements); <4>: 16-plane programs, 2
  static const int cin_wpc = getenv("LFGPU_BS_CIN_WPC") ? atoi(getenv("LFGPU_BS_CIN_WPC")) : (K == 5 ? 4 : 2);
  const dim3 grid(ncols / BS_COLS, gm.nrg);
  if (cin_wpc == 3)
    hipLaunchKernelGGL((bs_cin_kernel<K, 3>), grid, dim3(256), 32768, c->stream, src, ld, (u32)rows, stride, units, valid);
  else if (cin_wpc == 4)
    hipLaunchKernelGGL((bs_cin_kernel<K, 4>), grid, dim3(256), 32768, c->stream, src, ld, (u32)rows, stride, units, valid);
  else
    hipLaunchKernelGGL((bs_cin_kernel<K, 2>), grid, dim3(256), 32768, c->stream, src, ld, (u32)rows, stride, units, valid);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}
// units -> rows: the columns [out_lo, out_hi) of the buffer's first `ncols` columns
template <int K>
static int bs_convert_out(lfgpu_ctx* c, const BsGeom<K>& gm, size_t rows, u32 ncols, const u32* units, u32 stride, elt_t* dst, size_t ld, u32 out_lo, u32 out_hi) {
  static const int cout_wpc = getenv("LFGPU_BS_COUT_WPC") ? atoi(getenv("LFGPU_BS_COUT_WPC")) : (K == 5 ? 4 : 2);
  if (out_hi > ncols) out_hi = ncols;
  if (out_lo >= out_hi) return LFGPU_OK;
  constexpr int M = Tower<K>::M;
  const u32 t0 = out_lo / BS_COLS, t1 = (out_hi + BS_COLS - 1) / BS_COLS;  // only the tiles that hold columns of the window
  const dim3 grid(t1 - t0, gm.nrg);
  units += (size_t)t0 * BS_COLS * M;
  dst += (size_t)t0 * BS_COLS;
  out_lo -= t0 * BS_COLS;
  out_hi -= t0 * BS_COLS;
  if (cout_wpc == 4)
    hipLaunchKernelGGL((bs_cout_kernel<K, 4>), grid, dim3(256), 32768, c->stream, units, ld, (u32)rows, stride, dst, out_lo, out_hi);
  else if (cout_wpc == 3)
    hipLaunchKernelGGL((bs_cout_kernel<K, 3>), grid, dim3(256), 32768, c->stream, units, ld, (u32)rows, stride, dst, out_lo, out_hi);
  else
    hipLaunchKernelGGL((bs_cout_kernel<K, 2>), grid, dim3(256), 32768, c->stream, units, ld, (u32)rows, stride, dst, out_lo, out_hi);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

template <int K>
static int lch_bs_run(lfgpu_ctx* c, const GfHostCtx* g, int inverse, size_t rows, unsigned l, u64 coset, void* d_B, size_t ld) {
  constexpr int M = Tower<K>::M, D = Tower<K>::D;
  LF_TRY(bs_setup(c));
  const BsGeom<K> gm = bs_geom<K>(c, rows);
  const u32 n = 1u << l;
  void* internal = nullptr;
  LF_TRY(lf_scratch(c, (size_t)gm.combos * n * M * 4, &internal));
  if (gm.combos > gm.nrg * D)  // padded combos: define the bits (values are never read back)
    LF_HIP(c, hipMemsetAsync((u32*)internal + (size_t)gm.nrg * D * n * M, 0, (size_t)(gm.combos - gm.nrg * D) * n * M * 4, c->stream));
  LF_TRY(bs_convert_in<K>(c, gm, rows, n, n, (const elt_t*)d_B, ld, (u32*)internal, n));
  LF_TRY(bs_passes<K>(c, g, gm, inverse, l, coset, (u32*)internal, n));
  LF_TRY(bs_convert_out<K>(c, gm, rows, n, (const u32*)internal, n, (elt_t*)d_B, ld, 0u, n));
  return LFGPU_OK;
}

// entry used by lfgpu_gf2128_lch14_fft (fft.hip) when the batch is large enough
int lf_lch14_fft_bitsliced(lfgpu_ctx* c, int k, int inverse, size_t rows, unsigned l, u64 coset, void* d_B, size_t ld) {
  const GfHostCtx* g = lf_gf_ctx(c, k);
  if (!g) return LFGPU_ERR_ARG;
  return k == 4 ? lch_bs_run<4>(c, g, inverse, rows, l, coset, d_B, ld) : lch_bs_run<5>(c, g, inverse, rows, l, coset, d_B, ld);
}

// ------------------------------------------------------------------ tower-domain building blocks for rs.hip
// The Reed-Solomon encoder of rows larger than LDS runs a truncated transform (whole FFT / IFFT blocks of shrinking size
// and partial butterfly ranges between them) and then one FFT per further coset -- about a dozen transforms on the same
// coefficients.  Converted once, they stay in the bit-sliced tower representation throughout: one bs_cin, butterfly passes
// and range kernels on sub-blocks of the unit buffer, one bs_cout per coset of evaluations.
template <int K>
__global__ __launch_bounds__(256) void bs_range_kernel(u32 kind, u32 s, u32 lo, u32 cnt, u32 t, u32 stride, u32 ncombo, u32* __restrict__ U) {
  // butterflies (uv, uv + s), uv in [lo, lo + cnt), of every combo; t = tower bits of the level's one twiddle
  // kind 0: b0 ^= t b1; b1 ^= b0 (lch14.h:219-223)   1: b1 ^= b0; b0 ^= t b1 (:225-229)   2: x = b1; b1 ^= b0; b0 ^= t x (:231-237)
  constexpr int M = Tower<K>::M;
  constexpr u32 MU = Tower<K>::MU_LOW;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)cnt * ncombo) return;
  const u32 uv = lo + (u32)(idx % cnt), combo = (u32)(idx / cnt);
  uint4* p0 = reinterpret_cast<uint4*>(U + ((size_t)combo * stride + uv) * M);
  uint4* p1 = reinterpret_cast<uint4*>(U + ((size_t)combo * stride + uv + s) * M);
  u32 b0[M], b1[M], acc[M];
#pragma unroll
  for (int j = 0; j < M / 4; ++j) {
    const uint4 x = p0[j], y = p1[j];
    b0[4 * j] = x.x; b0[4 * j + 1] = x.y; b0[4 * j + 2] = x.z; b0[4 * j + 3] = x.w;
    b1[4 * j] = y.x; b1[4 * j + 1] = y.y; b1[4 * j + 2] = y.z; b1[4 * j + 3] = y.w;
  }
  const u32 tt = __builtin_amdgcn_readfirstlane(t);
  if (kind == 1) {
#pragma unroll
    for (int p = 0; p < M; ++p) b1[p] ^= b0[p];
  }
  bs_mul_horner<M, MU>(tt, b1, acc);
  if (kind == 2) {
#pragma unroll
    for (int p = 0; p < M; ++p) b1[p] ^= b0[p];
  }
#pragma unroll
  for (int p = 0; p < M; ++p) b0[p] ^= acc[p];
  if (kind == 0) {
#pragma unroll
    for (int p = 0; p < M; ++p) b1[p] ^= b0[p];
  }
#pragma unroll
  for (int j = 0; j < M / 4; ++j) {
    p0[j] = make_uint4(b0[4 * j], b0[4 * j + 1], b0[4 * j + 2], b0[4 * j + 3]);
    p1[j] = make_uint4(b1[4 * j], b1[4 * j + 1], b1[4 * j + 2], b1[4 * j + 3]);
  }
}
// dst = src for the columns below zero_from, 0 from there on (16-byte pieces; the coefficient vector of a further coset)
__global__ __launch_bounds__(256) void bs_copy_units_kernel(size_t pieces_per_combo, size_t keep_pieces, size_t total, const uint4* __restrict__ src,
                                                            uint4* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  dst[i] = (i % pieces_per_combo) < keep_pieces ? src[i] : make_uint4(0, 0, 0, 0);
}

template <int K>
static int bs_tower_ops(lfgpu_ctx* c, const GfHostCtx* g, int op, size_t rows, u32 a0, u32 a1, u32 a2, u32 a3, u64 coset, elt_t tw, void* U, u32 stride,
                        const void* src, void* dst, size_t ld) {
  constexpr int M = Tower<K>::M, D = Tower<K>::D;
  LF_TRY(bs_setup(c));
  const BsGeom<K> gm = bs_geom<K>(c, rows);
  u32* units = (u32*)U;
  switch (op) {
    case 0:  // cin: a0 = ncols, a1 = valid
      if (gm.combos > gm.nrg * D)
        LF_HIP(c, hipMemsetAsync(units + (size_t)gm.nrg * D * stride * M, 0, (size_t)(gm.combos - gm.nrg * D) * stride * M * 4, c->stream));
      return bs_convert_in<K>(c, gm, rows, a0, a1, (const elt_t*)src, ld, units, stride);
    case 1:  // cout: a0 = ncols, a1 = out_lo, a2 = out_hi
      return bs_convert_out<K>(c, gm, rows, a0, units, stride, (elt_t*)dst, ld, a1, a2);
    case 2:  // FFT / IFFT passes: a0 = l, a1 = inverse, a2 = first column; src != nullptr: the first pass reads that unit buffer (a3 = valid columns)
      if (src && gm.combos > gm.nrg * D)
        LF_HIP(c, hipMemsetAsync(units + (size_t)gm.nrg * D * stride * M, 0, (size_t)(gm.combos - gm.nrg * D) * stride * M * 4, c->stream));
      return bs_passes<K>(c, g, gm, (int)a1, a0, coset, units + (size_t)a2 * M, stride, src ? (const u32*)src + (size_t)a2 * M : nullptr, src ? a3 : 0xffffffffu);
    case 3: {  // range: a0 = kind, a1 = s, a2 = first column of the lower half + lo, a3 = count
      if (a3 == 0) return LFGPU_OK;
      const u32 t = tower_twiddle_bits<K>(tw.lo, tw.hi);
      const size_t total = (size_t)a3 * gm.nrg * D;
      hipLaunchKernelGGL(bs_range_kernel<K>, dim3((u32)((total + 255) / 256)), dim3(256), 0, c->stream, a0, a1, a2, a3, t, stride, gm.nrg * D, units);
      LF_HIP(c, hipGetLastError());
      return LFGPU_OK;
    }
    default: {  // copy with zero tail: a0 = zero_from; dst = the other unit buffer
      const size_t ppc = (size_t)stride * M / 4, total = ppc * gm.combos;
      hipLaunchKernelGGL(bs_copy_units_kernel, dim3((u32)((total + 255) / 256)), dim3(256), 0, c->stream, ppc, (size_t)a0 * M / 4, total, (const uint4*)U, (uint4*)dst);
      LF_HIP(c, hipGetLastError());
      return LFGPU_OK;
    }
  }
}
// bytes of a unit buffer for `rows` batch rows x `stride` columns
size_t lf_bs_units_bytes(lfgpu_ctx* c, int k, size_t rows, u32 stride) {
  return k == 4 ? (size_t)bs_geom<4>(c, rows).combos * stride * Tower<4>::M * 4 : (size_t)bs_geom<5>(c, rows).combos * stride * Tower<5>::M * 4;
}
int lf_bs_tower_op(lfgpu_ctx* c, int k, int op, size_t rows, u32 a0, u32 a1, u32 a2, u32 a3, u64 coset, elt_t tw, void* U, u32 stride, const void* src,
                   void* dst, size_t ld) {
  const GfHostCtx* g = lf_gf_ctx(c, k);
  if (!g) return LFGPU_ERR_ARG;
  return k == 4 ? bs_tower_ops<4>(c, g, op, rows, a0, a1, a2, a3, coset, tw, U, stride, src, dst, ld)
                : bs_tower_ops<5>(c, g, op, rows, a0, a1, a2, a3, coset, tw, U, stride, src, dst, ld);
}
