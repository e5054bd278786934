"""Multi-GPU use of the hot path from Python: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU node; "gloo" in the CPU tests and in the one-GPU rehearsal, where device buffers are staged through host memory).

The orchestration itself is C++ behind the C ABI (include/lfgpu.h: lfgpu_ligero_commit_sharded and the prove entry points,
include/lfgpu_zk.h: lfgpu_zk_prover_set_comm); what this module adds is the BINDING of the library's three transport hooks
(lfgpu_comm_ops: all_gather, all_to_all, broadcast -- the way the transcript and the RandomEngine are hooks) to
torch.distributed, and thin wrappers.  A C++ caller binds the same hooks to RCCL directly.

What shards and what is exchanged (reference lib/ligero/ligero_prover.h:58-79,171-270,
lib/merkle/merkle_commitment.h:50-64):
  * FFT / RS row encode: rows are independent -> contiguous row slabs per rank, no collective.
  * RandomEngine draws: sequential by definition.  Rank 0 draws the whole stream of LigeroProver::commit once and
    broadcasts the bytes; every rank replays them through the library's host layout and keeps its slab.
  * Merkle column commit: a leaf hashes ALL rows of one column, so the encoded slab is re-partitioned by columns with ONE
    all_to_all, leaves are hashed locally, the 32-byte digests are all_gathered and every rank builds the (small) tree ->
    identical root on every rank.  SHA-256 is not a reduction: there is no "all-reduce of roots".
  * Ligero prove: each rank combines the rows of its slab, the partial vectors are all_gathered and folded with the FIELD's
    addition (RCCL has no XOR / mod-p reduce op); opened columns are gathered slab by slab.
  * sumcheck: replicated (see DESIGN.md section 6 for why the index-range split is not built).

Bytes per collective (nrow rows, block_ext leaves, N ranks, 16-byte elements):
  broadcast of the random stream   16 (block + 2 dblock + r nwqrow) + 32 block_ext          (host, once)
  all_to_all column re-partition   nrow block_ext 16 (N-1)/N on the wire; per xGMI link and direction nrow block_ext 16 / N^2
  all_gather of leaf digests       32 block_ext received per rank
  prove partial vectors            16 (block + 2 dblock) per rank
"""
import contextlib
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

FP128_P = 2**128 - 2**108 + 1

_vp, _sz, _ci = C.c_void_p, C.c_size_t, C.c_int
_psz = C.POINTER(C.c_size_t)
AG_FN = C.CFUNCTYPE(_ci, _vp, _vp, _vp, _sz, _ci, _vp)
A2A_FN = C.CFUNCTYPE(_ci, _vp, _vp, _psz, _psz, _vp, _psz, _psz, _ci, _vp)
BC_FN = C.CFUNCTYPE(_ci, _vp, _vp, _sz, _ci, _ci, _vp)


class CommOps(C.Structure):
    """lfgpu_comm_ops (include/lfgpu.h)"""
    _fields_ = [("user", _vp), ("rank", _ci), ("world", _ci), ("all_gather", AG_FN), ("all_to_all", A2A_FN), ("broadcast", BC_FN)]


def row_shard(nrows, rank, world):
    """contiguous slab [start, start+count) of `nrows` for `rank` (sizes differ by at most 1)"""
    base, rem = divmod(nrows, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def col_shard(ncols, rank, world):
    return row_shard(ncols, rank, world)


def ligero_row_shard(p, rank, world):
    """lfgpu_ligero_row_shard: row slab [lo, hi) of a Ligero tableau for `rank` (the quadratic rows all on the last rank)"""
    from . import load_library
    lo, hi = C.c_size_t(), C.c_size_t()
    rc = load_library().lfgpu_ligero_row_shard(C.byref(p), rank, world, C.byref(lo), C.byref(hi))
    if rc != 0:
        raise RuntimeError("lfgpu_ligero_row_shard failed with code %d" % rc)
    return lo.value, hi.value


def field_add(field, a, b):
    """a, b: (lo, hi) u64 pairs.  GF(2^128): XOR; Fp128 (Montgomery images are additive): mod p."""
    if field == 4:
        return (a[0] ^ b[0], a[1] ^ b[1])
    s = ((a[0] | (a[1] << 64)) + (b[0] | (b[1] << 64))) % FP128_P
    return (s & (2**64 - 1), s >> 64)


class _DevMem:
    """a raw device pointer as a CUDA-array-interface object (zero-copy torch view)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 3, "strides": None}


class TorchComm:
    """The library's transport hooks over a torch.distributed process group.  nccl (= RCCL): device buffers go into the
    collectives as they are, host buffers are staged through a device tensor; gloo: device buffers are staged through host
    memory (the one-GPU rehearsal and the CPU tests)."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.gloo = dist.get_backend(group) == "gloo"
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
        self.error = None
        self._ag, self._a2a, self._bc = AG_FN(self._all_gather), A2A_FN(self._all_to_all), BC_FN(self._broadcast)
        self.ops = CommOps(None, self.rank, self.world, self._ag, self._a2a, self._bc)

    # -- views of the library's buffers
    def _view(self, ptr, nbytes, on_device):
        if nbytes == 0 or not ptr:
            return torch.empty(0, dtype=torch.uint8, device=self.device if on_device else "cpu")
        if on_device:
            return torch.as_tensor(_DevMem(ptr, nbytes), device=self.device)
        return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

    def _on(self, stream):
        """order after the work queued on the library's stream (it is usually torch's current one already)"""
        if stream and self.device is not None and self.device.type == "cuda":
            return torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=self.device))
        return contextlib.nullcontext()

    def _guard(self, fn):
        try:
            fn()
            return 0
        except Exception as e:  # noqa: BLE001 -- a hook must not raise through the C frames above it
            import traceback
            self.error = "%r\n%s" % (e, traceback.format_exc())
            return 1

    # -- hooks
    def _all_gather(self, _user, send, recv, nbytes, on_device, stream):
        def run():
            with self._on(stream):
                s, r = self._view(send, nbytes, on_device), self._view(recv, nbytes * self.world, on_device)
                if self.gloo or not on_device:  # (a one-rank group goes through the backend too: bench.py --force-dist rehearses RCCL that way)
                    if self.gloo:
                        hs = s.cpu() if on_device else s
                        parts = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                        dist.all_gather(parts, hs.contiguous(), group=self.group)
                        r.copy_(torch.cat(parts))
                    else:  # nccl, host buffers: through device tensors
                        ds = s.to(self.device)
                        dr = torch.empty(nbytes * self.world, dtype=torch.uint8, device=self.device)
                        dist.all_gather_into_tensor(dr, ds, group=self.group)
                        r.copy_(dr.cpu())
                else:
                    dist.all_gather_into_tensor(r, s, group=self.group)
                if on_device and self.device is not None and self.device.type == "cuda":
                    torch.cuda.current_stream().synchronize()
        return self._guard(run)

    def _all_to_all(self, _user, send, soff, sbytes, recv, roff, rbytes, on_device, stream):
        def run():
            W, me = self.world, self.rank
            so, sb = [soff[q] for q in range(W)], [sbytes[q] for q in range(W)]
            ro, rb = [roff[q] for q in range(W)], [rbytes[q] for q in range(W)]
            with self._on(stream):
                s = self._view(send, max(o + b for o, b in zip(so, sb)), on_device)
                r = self._view(recv, max(o + b for o, b in zip(ro, rb)), on_device)
                sl = [s[so[q]:so[q] + sb[q]] for q in range(W)]
                rl = [r[ro[q]:ro[q] + rb[q]] for q in range(W)]
                if self.gloo:  # no all_to_all, no device tensors: pairwise send / recv on host copies
                    hs = [t.cpu().contiguous() for t in sl]
                    hr = [torch.empty(rb[q], dtype=torch.uint8) for q in range(W)]
                    hr[me].copy_(hs[me])
                    reqs = []
                    for q in range(W):
                        if q != me:
                            if sb[q]:
                                reqs.append(dist.isend(hs[q], q, group=self.group))
                            if rb[q]:
                                reqs.append(dist.irecv(hr[q], q, group=self.group))
                    for w in reqs:
                        w.wait()
                    for t, h in zip(rl, hr):
                        t.copy_(h)
                else:
                    if on_device:
                        dist.all_to_all(rl, [t.contiguous() for t in sl], group=self.group)
                    else:
                        dr = [torch.empty(rb[q], dtype=torch.uint8, device=self.device) for q in range(W)]
                        dist.all_to_all(dr, [t.to(self.device) for t in sl], group=self.group)
                        for t, d in zip(rl, dr):
                            t.copy_(d.cpu())
                if on_device and self.device is not None and self.device.type == "cuda":
                    torch.cuda.current_stream().synchronize()
        return self._guard(run)

    def _broadcast(self, _user, buf, nbytes, root, on_device, stream):
        def run():
            with self._on(stream):
                b = self._view(buf, nbytes, on_device)
                src = dist.get_global_rank(self.group, root) if self.group is not None else root
                if self.gloo:
                    h = b.cpu().contiguous() if on_device else b
                    dist.broadcast(h, src, group=self.group)
                    if on_device:
                        b.copy_(h)
                elif on_device:
                    dist.broadcast(b, src, group=self.group)
                    torch.cuda.current_stream().synchronize()
                else:
                    d = b.to(self.device)
                    dist.broadcast(d, src, group=self.group)
                    b.copy_(d.cpu())
        return self._guard(run)

    def selftest(self):
        """lfgpu_comm_selftest: every hook with host buffers (ragged all_to_all, every broadcast root)"""
        from . import load_library
        rc = load_library().lfgpu_comm_selftest(C.byref(self.ops))
        if rc != 0:
            raise RuntimeError("lfgpu_comm_selftest failed with code %d: %s" % (rc, self.error))


# ------------------------------------------------------------------ sumcheck partial sums
def allgather_fold_partials(field, a0, a2, group=None, device="cpu"):
    """combine per-rank sumcheck partial sums.  a0, a2: (lo, hi) python ints."""
    world = dist.get_world_size(group)

    def enc(v):  # u64 -> i64 two's complement for the int64 tensor
        return v - (1 << 64) if v >= (1 << 63) else v

    mine = torch.tensor([enc(a0[0]), enc(a0[1]), enc(a2[0]), enc(a2[1])], dtype=torch.int64, device=device)
    allp = [torch.empty_like(mine) for _ in range(world)]
    if dist.get_backend(group) == "gloo" and mine.device.type != "cpu":
        hp = [torch.empty(4, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(hp, mine.cpu(), group=group)
        allp = hp
    else:
        dist.all_gather(allp, mine, group=group)
    s0, s2 = (0, 0), (0, 0)
    for t in allp:
        v = [int(x) & (2**64 - 1) for x in t.cpu().tolist()]
        s0 = field_add(field, s0, (v[0], v[1]))
        s2 = field_add(field, s2, (v[2], v[3]))
    return s0, s2


# ------------------------------------------------------------------ host layout
def _rng_cb(rng_bytes):
    from . import RNG_FN

    def cb(_user, buf, n):
        C.memmove(buf, rng_bytes(n), n)

    return RNG_FN(cb)


def layout_rows(lib, field, k, p, W, subfield_boundary, lqc, rng_bytes, row_lo, row_hi, want_nonces=True):
    """lfgpu_ligero_layout_rows (host only, no device): all RandomEngine draws of LigeroProver::commit in the reference's
    order; returns (numpy uint64 [row_hi-row_lo][dblock][2], nonces bytes or None)"""
    fn = _rng_cb(rng_bytes)
    W = np.ascontiguousarray(W, dtype=np.uint64)
    rows = np.zeros((row_hi - row_lo, p.dblock, 2), dtype=np.uint64)
    nonces = (C.c_uint8 * (32 * p.block_ext))() if want_nonces else None
    lq = (C.c_size_t * max(1, 3 * p.nq))(*[int(x) for x in np.asarray(lqc, dtype=np.uint64).reshape(-1)]) if p.nq else None
    rc = lib.lfgpu_ligero_layout_rows(field, k, C.byref(p), C.c_void_p(W.ctypes.data) if W.size else None, subfield_boundary, lq, fn, None,
                                      row_lo, row_hi, C.c_void_p(rows.ctypes.data) if rows.size else None, nonces)
    if rc != 0:
        raise RuntimeError("lfgpu_ligero_layout_rows failed with code %d" % rc)
    return rows, (bytes(nonces) if want_nonces else None)


def layout_rows_sharded(lib, field, k, p, W, subfield_boundary, lqc, rng_bytes, comm):
    """lfgpu_ligero_layout_rows_sharded (host only): rank 0 draws, the stream is broadcast through `comm`, every rank keeps
    its slab -> (numpy uint64 [my rows][dblock][2], nonces bytes)"""
    fn = _rng_cb(rng_bytes if comm.rank == 0 else (lambda n: b"\x00" * n))
    W = np.ascontiguousarray(W, dtype=np.uint64)
    lo, hi = ligero_row_shard(p, comm.rank, comm.world)
    rows = np.zeros((hi - lo, p.dblock, 2), dtype=np.uint64)
    nonces = (C.c_uint8 * (32 * p.block_ext))()
    lq = (C.c_size_t * max(1, 3 * p.nq))(*[int(x) for x in np.asarray(lqc, dtype=np.uint64).reshape(-1)]) if p.nq else None
    rc = lib.lfgpu_ligero_layout_rows_sharded(field, k, C.byref(p), C.c_void_p(W.ctypes.data) if W.size else None, subfield_boundary, lq, fn, None,
                                              C.byref(comm.ops), C.c_void_p(rows.ctypes.data) if rows.size else None, nonces)
    if rc != 0:
        raise RuntimeError("lfgpu_ligero_layout_rows_sharded failed with code %d: %s" % (rc, comm.error))
    return rows, bytes(nonces)


# ------------------------------------------------------------------ the sharded prover
class ShardedLigeroProver:
    """LigeroProver<Field, InterpolatorFactory> (reference lib/ligero/ligero_prover.h:34-359) with the tableau rows sharded
    over the ranks of `group`: lfgpu_ligero_commit_sharded, then the ordinary prove entry points (which fold the ranks'
    partial vectors inside the library).  Every rank calls every method with the same arguments (SPMD); `rng_bytes` is only
    used on rank 0.  Results (root, y vectors, opened columns, Merkle path) are identical on every rank and identical to the
    one-GPU LigeroProver fed the same RandomEngine stream."""

    def __init__(self, gpu, field, param, subfield_log_bits=4, group=None, comm=None):
        from . import LigeroProver
        self.gpu, self.field, self.p, self.k = gpu, field, param, subfield_log_bits
        self.comm = comm if comm is not None else TorchComm(group)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.spans = [ligero_row_shard(param, q, self.world) for q in range(self.world)]
        self.pr = LigeroProver(gpu, field, param, subfield_log_bits)

    def commit(self, W, subfield_boundary, lqc, rng_bytes):
        """LigeroProver::commit (:58-79) without ts.write -> 32-byte root"""
        gpu = self.gpu
        self._cb = _rng_cb(rng_bytes if self.rank == 0 else (lambda n: b"\x00" * n))
        lq = np.ascontiguousarray(np.asarray(lqc, dtype=np.uint64).reshape(-1))
        W = np.ascontiguousarray(W)
        root = (C.c_uint8 * 32)()
        h = C.c_void_p()
        rc = gpu.L.lfgpu_ligero_commit_sharded(gpu.h, self.field, self.k, C.byref(self.p), C.c_void_p(W.ctypes.data), subfield_boundary,
                                               C.c_void_p(lq.ctypes.data) if lq.size else None, self._cb, None, C.byref(self.comm.ops), root, C.byref(h))
        if rc != 0:
            raise RuntimeError("lfgpu_ligero_commit_sharded: %s | hook: %s" % (gpu.L.lfgpu_last_error(gpu.h).decode(), self.comm.error))
        self.pr.h = h
        return bytes(root)

    def low_degree_proof(self, u_ldt):
        return self.pr.low_degree_proof(u_ldt)

    def dot_proof(self, A):
        return self.pr.dot_proof(A)

    def quadratic_proof(self, u_quad):
        return self.pr.quadratic_proof(u_quad)

    def open(self, idx):
        return self.pr.open(idx)

    def close(self):
        if self.pr is not None:
            self.pr.close()
            self.pr = None
