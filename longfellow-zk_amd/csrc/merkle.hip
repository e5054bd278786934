// merkle.hip -- K5 (column leaf SHA-256) and K6 (Merkle heap tree).
//
// Reference: MerkleCommitment::commit (lib/merkle/merkle_commitment.h:50-64),
// LigeroCommon::column_hash (lib/ligero/ligero_param.h:432-439),
// MerkleTree::build_tree / Digest::hash2 (lib/merkle/merkle_tree.h:51-58,109-114),
// MerkleTree::generate_compressed_proof (:122-143).
//
// K5: one lane per tableau column; lane j walks down column col0+j, so every
// row read is a coalesced 16 B/lane wavefront load along the row.  The SHA
// message schedule lives in 16 VGPRs with static indices only.
#include "ctx.h"

#define SHA_THREADS 256

// leaf_j = SHA256(nonce_j[32] || canon(T[0][col0+j]) || ... || canon(T[nrow-1][col0+j]))
// digest j lands at out[out0 + j] (out0 = ncols: the leaf level of the heap; 0: a plain leaf array)
template <int F>
__global__ __launch_bounds__(SHA_THREADS) void column_leaves_kernel(u32 nrow, size_t ld, size_t col0, u32 ncols,
                                                                     const elt_t* __restrict__ T,
                                                                     const uint4* __restrict__ nonces,
                                                                     uint4* __restrict__ layers, size_t out0) {
  u32 j = blockIdx.x * SHA_THREADS + threadIdx.x;
  if (j >= ncols) return;
  const elt_t* col = T + col0 + j;
  sha_state st;
  sha_init(st);
  const u32 nch = 2 + nrow;                       // 16-byte chunks of message
  const u32 nblk = (nch * 16 + 9 + 63) / 64;      // SHA blocks incl. padding
  const u64 bits = (u64)nch * 128;
  for (u32 bi = 0; bi < nblk; ++bi) {
    u32 w[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      u32 ch = 4 * bi + q;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ch < 2) {
        uint4 nb = nonces[2 * (size_t)j + ch];
        v = make_uint4(bswap32(nb.x), bswap32(nb.y), bswap32(nb.z), bswap32(nb.w));
      } else if (ch < nch) {
        elt_t e = Fld<F>::canon(ld16(col + (size_t)(ch - 2) * ld));
        v = make_uint4(bswap32((u32)e.lo), bswap32((u32)(e.lo >> 32)), bswap32((u32)e.hi), bswap32((u32)(e.hi >> 32)));
      } else if (ch == nch) {
        v.x = 0x80000000u;
      }
      w[4 * q + 0] = v.x;
      w[4 * q + 1] = v.y;
      w[4 * q + 2] = v.z;
      w[4 * q + 3] = v.w;
    }
    if (bi == nblk - 1) {
      w[14] = (u32)(bits >> 32);
      w[15] = (u32)bits;
    }
    sha_compress(st, w);
  }
  uint4 o0 = make_uint4(bswap32(st.h[0]), bswap32(st.h[1]), bswap32(st.h[2]), bswap32(st.h[3]));
  uint4 o1 = make_uint4(bswap32(st.h[4]), bswap32(st.h[5]), bswap32(st.h[6]), bswap32(st.h[7]));
  layers[2 * (out0 + j)] = o0;
  layers[2 * (out0 + j) + 1] = o1;
}

// node i = SHA256(node 2i || node 2i+1)
__device__ __forceinline__ void hash2_node(uint4* layers, size_t i) {
  const uint4* ch = layers + 4 * i;  // children 2i, 2i+1 are 64 contiguous bytes
  u32 w[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint4 v = ch[q];
    w[4 * q + 0] = bswap32(v.x);
    w[4 * q + 1] = bswap32(v.y);
    w[4 * q + 2] = bswap32(v.z);
    w[4 * q + 3] = bswap32(v.w);
  }
  sha_state st;
  sha_init(st);
  sha_compress(st, w);
#pragma unroll
  for (int q = 0; q < 16; ++q) w[q] = 0;
  w[0] = 0x80000000u;
  w[15] = 512;
  sha_compress(st, w);
  layers[2 * i] = make_uint4(bswap32(st.h[0]), bswap32(st.h[1]), bswap32(st.h[2]), bswap32(st.h[3]));
  layers[2 * i + 1] = make_uint4(bswap32(st.h[4]), bswap32(st.h[5]), bswap32(st.h[6]), bswap32(st.h[7]));
}

// one heap level: nodes [lo, hi)
__global__ __launch_bounds__(SHA_THREADS) void merkle_level_kernel(uint4* layers, size_t lo, size_t hi) {
  size_t i = lo + (size_t)blockIdx.x * SHA_THREADS + threadIdx.x;
  if (i < hi) hash2_node(layers, i);
}

// the top of the heap (levels d_top .. 0, each <= 1024 nodes) in one workgroup
__global__ __launch_bounds__(1024) void merkle_top_kernel(uint4* layers, size_t n, int d_top) {
  for (int d = d_top; d >= 0; --d) {
    size_t lo = (size_t)1 << d, hi = (size_t)2 << d;
    if (hi > n) hi = n;
    size_t i = lo + threadIdx.x;
    if (i < hi) hash2_node(layers, i);
    __threadfence_block();
    __syncthreads();
  }
}

__global__ void gather_digests_kernel(const uint4* __restrict__ layers, const u64* __restrict__ idx, u32 cnt,
                                      uint4* __restrict__ out) {
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < 2 * cnt) out[t] = layers[2 * idx[t >> 1] + (t & 1)];
}

static int build_tree(lfgpu_ctx* c, size_t n, void* d_layers) {
  if (n < 2) return LFGPU_OK;
  int d = 0;
  while (((size_t)2 << d) <= n - 1) ++d;  // highest level with an internal node: 2^d <= n-1
  for (; d > 10; --d) {
    size_t lo = (size_t)1 << d, hi = (size_t)2 << d;
    if (hi > n) hi = n;
    u32 nb = (u32)((hi - lo + SHA_THREADS - 1) / SHA_THREADS);
    hipLaunchKernelGGL(merkle_level_kernel, dim3(nb), dim3(SHA_THREADS), 0, c->stream, (uint4*)d_layers, lo, hi);
  }
  hipLaunchKernelGGL(merkle_top_kernel, dim3(1), dim3(1024), 0, c->stream, (uint4*)d_layers, n, d);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

static int read_root(lfgpu_ctx* c, size_t n, const void* d_layers, uint8_t root_out[32]) {
  // n == 1: the root is the single leaf (layers[1]); build_tree's loop is empty (merkle_tree.h:110-113)
  LF_HIP(c, hipMemcpyAsync(c->mailbox_h, (const uint8_t*)d_layers + 32, 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(root_out, c->mailbox_h, 32);
  return LFGPU_OK;
}

extern "C" int lfgpu_merkle_build_tree(lfgpu_ctx* c, size_t n, void* d_layers, uint8_t root_out[32]) {
  if (!c || !d_layers || !root_out || n == 0) return lf_fail(c, LFGPU_ERR_ARG, "merkle_build_tree: bad argument");
  LF_HIP(c, hipSetDevice(c->device));
  LF_TRY(build_tree(c, n, d_layers));
  return read_root(c, n, d_layers, root_out);
}

static int column_leaves(lfgpu_ctx* c, const char* who, int field, size_t nrow, size_t ld, size_t col0, size_t ncols, const void* d_T,
                         const void* d_nonces, void* d_out, size_t out0) {
  if (!c || !d_T || !d_nonces || !d_out || ncols == 0) return lf_fail(c, LFGPU_ERR_ARG, "%s: bad argument", who);
  if (col0 + ncols > ld) return lf_fail(c, LFGPU_ERR_ARG, "%s: col0 + ncols > ld", who);
  if (nrow > 0x0fffffffu || ncols > 0x7fffffffu) return lf_fail(c, LFGPU_ERR_ARG, "%s: too large", who);
  LF_HIP(c, hipSetDevice(c->device));
  u32 nb = (u32)((ncols + SHA_THREADS - 1) / SHA_THREADS);
  if (field == LFGPU_FIELD_GF2_128)
    hipLaunchKernelGGL(column_leaves_kernel<FIELD_GF2_128>, dim3(nb), dim3(SHA_THREADS), 0, c->stream, (u32)nrow, ld,
                       col0, (u32)ncols, (const elt_t*)d_T, (const uint4*)d_nonces, (uint4*)d_out, out0);
  else if (field == LFGPU_FIELD_FP128)
    hipLaunchKernelGGL(column_leaves_kernel<FIELD_FP128>, dim3(nb), dim3(SHA_THREADS), 0, c->stream, (u32)nrow, ld,
                       col0, (u32)ncols, (const elt_t*)d_T, (const uint4*)d_nonces, (uint4*)d_out, out0);
  else if (field == LFGPU_FIELD_P256)
    return lf_column_leaves32(c, nrow, ld, col0, ncols, d_T, d_nonces, d_out, out0);
  else
    return lf_fail(c, LFGPU_ERR_ARG, "%s: unknown field %d", who, field);
  LF_HIP(c, hipGetLastError());
  return LFGPU_OK;
}

extern "C" int lfgpu_column_commit(lfgpu_ctx* c, int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                                   const void* d_T, const void* d_nonces, void* d_layers, uint8_t root_out[32]) {
  if (!root_out) return lf_fail(c, LFGPU_ERR_ARG, "column_commit: bad argument");
  LF_TRY(column_leaves(c, "column_commit", field, nrow, ld, col0, ncols, d_T, d_nonces, d_layers, ncols));
  LF_TRY(build_tree(c, ncols, d_layers));
  return read_root(c, ncols, d_layers, root_out);
}

// the leaf half alone (multi-GPU commit: a rank hashes the columns it owns); enqueue only
extern "C" int lfgpu_column_leaves(lfgpu_ctx* c, int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                                   const void* d_T, const void* d_nonces, void* d_leaves) {
  return column_leaves(c, "column_leaves", field, nrow, ld, col0, ncols, d_T, d_nonces, d_leaves, 0);
}

// MerkleTree::generate_compressed_proof (merkle_tree.h:63-84,122-143): the index walk is
// host logic on positions only; the digests are gathered on the device.
extern "C" int lfgpu_merkle_open(lfgpu_ctx* c, size_t n, const void* d_layers, const size_t* pos, size_t np,
                                 uint8_t* h_path, size_t path_cap, size_t* npath) {
  if (!c || !d_layers || !pos || !npath || np == 0 || n == 0)
    return lf_fail(c, LFGPU_ERR_ARG, "merkle_open: bad argument (a proof with 0 leaves is not defined)");
  std::vector<bool> tree(2 * n, false);
  for (size_t ip = 0; ip < np; ++ip) {
    if (pos[ip] >= n) return lf_fail(c, LFGPU_ERR_ARG, "merkle_open: invalid leaf position");
    if (tree[pos[ip] + n]) return lf_fail(c, LFGPU_ERR_ARG, "merkle_open: duplicate position");
    tree[pos[ip] + n] = true;
  }
  for (size_t i = n; i-- > 1;) tree[i] = tree[2 * i] || tree[2 * i + 1];
  std::vector<u64> idx;
  for (size_t i = n; i-- > 1;) {
    if (tree[i]) {
      size_t child = 2 * i;
      if (tree[child]) child = 2 * i + 1;
      if (!tree[child]) idx.push_back(child);
    }
  }
  *npath = idx.size();
  if (idx.empty()) return LFGPU_OK;
  if (!h_path || idx.size() > path_cap) return lf_fail(c, LFGPU_ERR_ARG, "merkle_open: path buffer too small");
  LF_HIP(c, hipSetDevice(c->device));
  void* sc = nullptr;
  LF_TRY(lf_scratch2(c, idx.size() * 40 + 64, &sc));
  u64* d_idx = (u64*)sc;
  uint4* d_out = (uint4*)((u8*)sc + ((idx.size() * 8 + 15) & ~(size_t)15));
  LF_HIP(c, hipMemcpyAsync(d_idx, idx.data(), idx.size() * 8, hipMemcpyHostToDevice, c->stream));
  u32 cnt = (u32)idx.size();
  hipLaunchKernelGGL(gather_digests_kernel, dim3((2 * cnt + 255) / 256), dim3(256), 0, c->stream,
                     (const uint4*)d_layers, (const u64*)d_idx, cnt, d_out);
  LF_HIP(c, hipGetLastError());
  LF_HIP(c, hipMemcpyAsync(h_path, d_out, idx.size() * 32, hipMemcpyDeviceToHost, c->stream));
  LF_HIP(c, hipStreamSynchronize(c->stream));
  return LFGPU_OK;
}

extern "C" int lfgpu_column_commit_host(lfgpu_ctx* c, int field, size_t nrow, size_t ld, size_t col0, size_t ncols,
                                        const void* h_T, const uint8_t* h_nonces, uint8_t* h_layers,
                                        uint8_t root_out[32]) {
  if (!c || !h_T || !h_nonces) return LFGPU_ERR_ARG;
  size_t tb = nrow * ld * (field == LFGPU_FIELD_P256 ? 32 : 16), nb = ncols * 32, lb = 2 * ncols * 32;
  void* d = nullptr;
  LF_TRY(lf_scratch(c, tb + nb + lb + 64, &d));
  u8* dT = (u8*)d;
  u8* dN = dT + ((tb + 15) & ~(size_t)15);
  u8* dL = dN + nb;
  LF_HIP(c, hipMemcpyAsync(dT, h_T, tb, hipMemcpyHostToDevice, c->stream));
  LF_HIP(c, hipMemcpyAsync(dN, h_nonces, nb, hipMemcpyHostToDevice, c->stream));
  LF_TRY(lfgpu_column_commit(c, field, nrow, ld, col0, ncols, dT, dN, dL, root_out));
  if (h_layers) {
    LF_HIP(c, hipMemcpyAsync(h_layers, dL, lb, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(c, hipStreamSynchronize(c->stream));
  }
  return LFGPU_OK;
}
