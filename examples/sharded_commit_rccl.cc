// sharded_commit_rccl.cc -- the library's multi-GPU transport hooks (lfgpu_comm_ops, include/lfgpu.h) bound to RCCL from plain
// C++, the way INTEGRATION.md section 4 describes: lfgpu_ligero_commit_sharded + the prove entry points run over the
// communicator and must give the one-GPU LigeroProver's root, y vectors and openings.  No Python, no torch.
//
// One process per GPU.  With RANKS > 1 the ncclUniqueId travels through a file (argv[3]) that rank 0 writes -- any launcher that
// starts the ranks (mpirun, srun, a shell loop) will do; on a one-GPU box only RANKS = 1 can run (RCCL does not put two ranks
// on one device), which still drives every hook through RCCL.
//
//   hipcc -std=c++17 -O2 -Iinclude examples/sharded_commit_rccl.cc -Llongfellow-zk_amd -llfgpu -lrccl -Wl,-rpath,$PWD/longfellow-zk_amd -o sharded_commit_rccl
//   ./sharded_commit_rccl RANK RANKS [id-file]
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lfgpu.h"

struct Comm {
  ncclComm_t nccl;
  int rank, world;
  hipStream_t side;  // for the host-buffer hooks (tiny payloads staged through device memory)
};
#define NCK(x)                                                                      \
  do {                                                                              \
    ncclResult_t r_ = (x);                                                          \
    if (r_ != ncclSuccess) {                                                        \
      fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));                      \
      return 1;                                                                     \
    }                                                                               \
  } while (0)
#define HCK(x)                                                                      \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

// host buffers: stage through device memory on the side stream, complete on return
static int staged(Comm* c, const void* h_send, size_t send_bytes, void* h_recv, size_t recv_bytes, int (*run)(Comm*, const void*, void*, size_t, hipStream_t, void*), void* arg) {
  void *ds = nullptr, *dr = nullptr;
  HCK(hipMalloc(&ds, send_bytes ? send_bytes : 16));
  HCK(hipMalloc(&dr, recv_bytes ? recv_bytes : 16));
  if (send_bytes) HCK(hipMemcpyAsync(ds, h_send, send_bytes, hipMemcpyHostToDevice, c->side));
  if (run(c, ds, dr, send_bytes, c->side, arg)) return 1;
  if (recv_bytes) HCK(hipMemcpyAsync(h_recv, dr, recv_bytes, hipMemcpyDeviceToHost, c->side));
  HCK(hipStreamSynchronize(c->side));
  (void)hipFree(ds);
  (void)hipFree(dr);
  return 0;
}
static int run_ag(Comm* c, const void* s, void* r, size_t n, hipStream_t st, void*) { NCK(ncclAllGather(s, r, n, ncclUint8, c->nccl, st)); return 0; }
static int hook_all_gather(void* u, const void* send, void* recv, size_t bytes, int on_device, void* stream) {
  Comm* c = (Comm*)u;
  if (!on_device) return staged(c, send, bytes, recv, bytes * (size_t)c->world, run_ag, nullptr);
  NCK(ncclAllGather(send, recv, bytes, ncclUint8, c->nccl, (hipStream_t)stream));  // enqueued on the library's stream: ordered
  return 0;
}
struct A2A {
  const size_t *so, *sb, *ro, *rb;
};
static int a2a_on(Comm* c, const void* send, const A2A& a, void* recv, hipStream_t st) {
  NCK(ncclGroupStart());
  for (int q = 0; q < c->world; ++q) {
    if (a.sb[q]) NCK(ncclSend((const char*)send + a.so[q], a.sb[q], ncclUint8, q, c->nccl, st));
    if (a.rb[q]) NCK(ncclRecv((char*)recv + a.ro[q], a.rb[q], ncclUint8, q, c->nccl, st));
  }
  NCK(ncclGroupEnd());
  return 0;
}
static int hook_all_to_all(void* u, const void* send, const size_t* so, const size_t* sb, void* recv, const size_t* ro, const size_t* rb, int on_device,
                           void* stream) {
  Comm* c = (Comm*)u;
  const A2A a{so, sb, ro, rb};
  if (on_device) return a2a_on(c, send, a, recv, (hipStream_t)stream);
  size_t st = 0, rt = 0;
  for (int q = 0; q < c->world; ++q) {
    st = st > so[q] + sb[q] ? st : so[q] + sb[q];
    rt = rt > ro[q] + rb[q] ? rt : ro[q] + rb[q];
  }
  void *ds = nullptr, *dr = nullptr;
  HCK(hipMalloc(&ds, st ? st : 16));
  HCK(hipMalloc(&dr, rt ? rt : 16));
  if (st) HCK(hipMemcpyAsync(ds, send, st, hipMemcpyHostToDevice, c->side));
  if (a2a_on(c, ds, a, dr, c->side)) return 1;
  if (rt) HCK(hipMemcpyAsync(recv, dr, rt, hipMemcpyDeviceToHost, c->side));
  HCK(hipStreamSynchronize(c->side));
  (void)hipFree(ds);
  (void)hipFree(dr);
  return 0;
}
static int hook_broadcast(void* u, void* buf, size_t bytes, int root, int on_device, void* stream) {
  Comm* c = (Comm*)u;
  if (on_device) {
    NCK(ncclBroadcast(buf, buf, bytes, ncclUint8, root, c->nccl, (hipStream_t)stream));
    return 0;
  }
  void* d = nullptr;
  HCK(hipMalloc(&d, bytes ? bytes : 16));
  if (bytes) HCK(hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, c->side));
  NCK(ncclBroadcast(d, d, bytes, ncclUint8, root, c->nccl, c->side));
  if (bytes) HCK(hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, c->side));
  HCK(hipStreamSynchronize(c->side));
  (void)hipFree(d);
  return 0;
}

// the statement: LigeroParam(nw, nq, 4, 24, 1024) over GF2_128<4> with valid quadratic constraints, an LCG engine
struct Lcg {
  uint64_t s;
};
static void lcg_bytes(void* u, uint8_t* b, size_t n) {
  Lcg* l = (Lcg*)u;
  for (size_t i = 0; i < n; ++i) {
    l->s = l->s * 6364136223846793005ull + 1442695040888963407ull;
    b[i] = (uint8_t)(l->s >> 32);
  }
}
#define LCK(ctx, x)                                                                 \
  do {                                                                              \
    int rc_ = (x);                                                                  \
    if (rc_ != LFGPU_OK) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, lfgpu_last_error(ctx));            \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main(int argc, char** argv) {
  const int rank = argc > 1 ? atoi(argv[1]) : 0, world = argc > 2 ? atoi(argv[2]) : 1;
  const char* idfile = argc > 3 ? argv[3] : "/tmp/lfgpu_nccl_id";
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    fprintf(stderr, "no MI355X / HIP device: there is no CPU fallback\n");
    return 1;
  }
  const int dev = rank % ndev;
  HCK(hipSetDevice(dev));
  ncclUniqueId id;
  if (rank == 0) {
    NCK(ncclGetUniqueId(&id));
    if (world > 1) {
      FILE* f = fopen(idfile, "wb");
      fwrite(&id, sizeof(id), 1, f);
      fclose(f);
    }
  } else {
    FILE* f = nullptr;
    for (int t = 0; t < 600 && !(f = fopen(idfile, "rb")); ++t) usleep(100000);
    if (!f || fread(&id, sizeof(id), 1, f) != 1) return 1;
    fclose(f);
  }
  Comm cm{};
  cm.rank = rank;
  cm.world = world;
  NCK(ncclCommInitRank(&cm.nccl, world, id, rank));
  HCK(hipStreamCreateWithFlags(&cm.side, hipStreamNonBlocking));
  lfgpu_comm_ops ops{&cm, rank, world, hook_all_gather, hook_all_to_all, hook_broadcast};
  LCK(nullptr, lfgpu_comm_selftest(&ops));  // every hook with host buffers, ragged blocks, every root

  lfgpu_ctx* ctx = nullptr;
  if (lfgpu_init(dev, &ctx) != LFGPU_OK) return 1;
  LCK(ctx, lfgpu_own_stream(ctx));
  const size_t nw = 3000, nq = 17;
  lfgpu_ligero_param p;
  LCK(ctx, lfgpu_ligero_param_init(&p, LFGPU_FIELD_GF2_128, 4, nw, nq, 4, 24, 1024));
  std::vector<uint64_t> W(2 * nw);
  Lcg wl{7};
  lcg_bytes(&wl, (uint8_t*)W.data(), 16 * nw);
  std::vector<size_t> lqc(3 * nq);
  // z = x * y for every constraint needs the field product: take x = 1 (the element with image 1): z = y
  W[0] = 1;
  W[1] = 0;
  for (size_t i = 0; i < nq; ++i) {
    lqc[3 * i] = 0;
    lqc[3 * i + 1] = 10 + i;
    lqc[3 * i + 2] = 1500 + i;
    W[2 * (1500 + i)] = W[2 * (10 + i)];
    W[2 * (1500 + i) + 1] = W[2 * (10 + i) + 1];
  }
  // one-GPU reference run on this rank, then the sharded run over the communicator: same engine seed
  uint8_t root1[32], rootN[32];
  lfgpu_ligero_prover *p1 = nullptr, *pN = nullptr;
  Lcg e1{100}, eN{100};
  LCK(ctx, lfgpu_ligero_commit(ctx, LFGPU_FIELD_GF2_128, 4, &p, W.data(), 0, lqc.data(), lcg_bytes, &e1, root1, &p1));
  LCK(ctx, lfgpu_ligero_commit_sharded(ctx, LFGPU_FIELD_GF2_128, 4, &p, W.data(), 0, lqc.data(), lcg_bytes, &eN, &ops, rootN, &pN));
  bool ok = memcmp(root1, rootN, 32) == 0;
  std::vector<uint64_t> u(2 * p.nwqrow), y1(2 * p.block), yN(2 * p.block);
  Lcg ul{9};
  lcg_bytes(&ul, (uint8_t*)u.data(), 16 * p.nwqrow);
  LCK(ctx, lfgpu_ligero_low_degree_proof(p1, u.data(), y1.data()));
  LCK(ctx, lfgpu_ligero_low_degree_proof(pN, u.data(), yN.data()));
  ok = ok && y1 == yN;
  std::vector<uint64_t> uq(2 * p.nqtriples), a0(2 * p.r), a2(2 * (p.dblock - p.block)), b0(2 * p.r), b2(2 * (p.dblock - p.block));
  lcg_bytes(&ul, (uint8_t*)uq.data(), 16 * p.nqtriples);
  LCK(ctx, lfgpu_ligero_quadratic_proof(p1, uq.data(), a0.data(), a2.data()));
  LCK(ctx, lfgpu_ligero_quadratic_proof(pN, uq.data(), b0.data(), b2.data()));
  ok = ok && a0 == b0 && a2 == b2;
  std::vector<size_t> idx(p.nreq);
  for (size_t i = 0; i < p.nreq; ++i) idx[i] = (i * 37 + 5) % p.block_ext;
  std::vector<uint64_t> r1(2 * p.nrow * p.nreq), rN(2 * p.nrow * p.nreq);
  std::vector<uint8_t> n1(32 * p.nreq), nN(32 * p.nreq), pa1(32 * (p.nreq * p.mc_pathlen + 1)), paN(32 * (p.nreq * p.mc_pathlen + 1));
  size_t np1 = 0, npN = 0;
  LCK(ctx, lfgpu_ligero_open(p1, idx.data(), r1.data(), n1.data(), pa1.data(), p.nreq * p.mc_pathlen + 1, &np1));
  LCK(ctx, lfgpu_ligero_open(pN, idx.data(), rN.data(), nN.data(), paN.data(), p.nreq * p.mc_pathlen + 1, &npN));
  ok = ok && r1 == rN && n1 == nN && np1 == npN && memcmp(pa1.data(), paN.data(), 32 * np1) == 0;
  printf("{\"rank\": %d, \"ranks\": %d, \"nrow\": %zu, \"block_ext\": %zu, \"sharded_equals_one_gpu\": %s}\n", rank, world, p.nrow, p.block_ext, ok ? "true" : "false");
  lfgpu_ligero_free(p1);
  lfgpu_ligero_free(pN);
  lfgpu_shutdown(ctx);
  ncclCommDestroy(cm.nccl);
  return ok ? 0 : 1;
}
