"""The multi-GPU orchestration of the Ligero commit / prove RESTATED IN PYTHON over an engine -- test infrastructure.

The product path is C++ behind the C ABI (lfgpu_ligero_commit_sharded + the prove entry points, csrc/ligero.hip, reached from
Python through longfellow-zk_amd/parallel.py).  This module keeps the round-2 Python orchestration as its checker: the same
steps (one random stream drawn on rank 0 and replayed, row-slab RS encode, all_to_all column re-partition, local column hash,
all_gather of the leaf digests, the tree on every rank; partial y vectors all_gathered and folded with the field's addition)
written against an *engine* whose methods are kernel-level C-ABI calls (GpuEngine below) or the oracle (tests/sharded_util.py,
CPU).  tests/test_distributed_gloo.py runs it with world 2 / 3 on CPU; tests/test_sharded_gpu.py compares the C entry point
with it, with the one-GPU prover and with the reference's C++ commitment root.

Reference: lib/ligero/ligero_prover.h:58-79,171-351, lib/merkle/merkle_commitment.h:50-64.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

FP128_P = 2**128 - 2**108 + 1


def row_shard(nrows, rank, world):
    """contiguous slab [start, start+count) of `nrows` for `rank` (sizes differ by at most 1)"""
    base, rem = divmod(nrows, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def col_shard(ncols, rank, world):
    return row_shard(ncols, rank, world)


def field_add(field, a, b):
    """a, b: (lo, hi) u64 pairs.  GF(2^128): XOR; Fp128 (Montgomery images are additive): mod p."""
    if field == 4:
        return (a[0] ^ b[0], a[1] ^ b[1])
    s = ((a[0] | (a[1] << 64)) + (b[0] | (b[1] << 64))) % FP128_P
    return (s & (2**64 - 1), s >> 64)


# ------------------------------------------------------------------ transport
def _is_gloo(group):
    return dist.get_backend(group) == "gloo"


def _all_to_all(recv, send, group):
    """list form of all_to_all on device tensors; gloo (no all_to_all, no device tensors): pairwise send/recv on
    host copies"""
    if not _is_gloo(group):
        dist.all_to_all(recv, send, group=group)
        return
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    hs = [t.cpu() for t in send]
    hr = [torch.empty(t.shape, dtype=t.dtype) for t in recv]
    hr[rank].copy_(hs[rank])
    reqs = []
    for q in range(world):
        if q != rank:
            reqs.append(dist.isend(hs[q], q, group=group))
            reqs.append(dist.irecv(hr[q], q, group=group))
    for r in reqs:
        r.wait()
    for t, h in zip(recv, hr):
        t.copy_(h)


def _all_gather(out_list, t, group):
    if not _is_gloo(group) or t.device.type == "cpu":
        dist.all_gather(out_list, t, group=group)
        return
    ho = [torch.empty(o.shape, dtype=o.dtype) for o in out_list]
    dist.all_gather(ho, t.cpu(), group=group)
    for o, h in zip(out_list, ho):
        o.copy_(h)


def _broadcast_bytes(data, src, group):
    """bytes on `src` -> the same bytes on every rank (length first)"""
    rank = dist.get_rank(group)
    n = torch.tensor([len(data) if rank == src else 0], dtype=torch.int64)
    dev = None if _is_gloo(group) else torch.device("cuda", torch.cuda.current_device())
    if dev is not None:
        n = n.to(dev)
    dist.broadcast(n, src, group=group)
    size = int(n.item())
    if rank == src:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8)
    else:
        buf = torch.empty(size, dtype=torch.uint8)
    if dev is not None:
        buf = buf.to(dev)
    dist.broadcast(buf, src, group=group)
    return bytes(buf.cpu().numpy().tobytes())


# ------------------------------------------------------------------ sumcheck partial sums
def allgather_fold_partials(field, a0, a2, group=None, device="cpu"):
    """combine per-rank sumcheck partial sums.  a0, a2: (lo, hi) python ints."""
    world = dist.get_world_size(group)

    def enc(v):  # u64 -> i64 two's complement for the int64 tensor
        return v - (1 << 64) if v >= (1 << 63) else v

    mine = torch.tensor([enc(a0[0]), enc(a0[1]), enc(a2[0]), enc(a2[1])], dtype=torch.int64, device=device)
    allp = [torch.empty_like(mine) for _ in range(world)]
    _all_gather(allp, mine, group)
    s0, s2 = (0, 0), (0, 0)
    for t in allp:
        v = [int(x) & (2**64 - 1) for x in t.cpu().tolist()]
        s0 = field_add(field, s0, (v[0], v[1]))
        s2 = field_add(field, s2, (v[2], v[3]))
    return s0, s2


# ------------------------------------------------------------------ engines
class GpuEngine:
    """This rank's MI355X through the C ABI (include/lfgpu.h): every method is a HIP kernel path of liblfgpu.so.
    Tensors are flat uint8 device tensors; torch only owns the memory."""

    def __init__(self, gpu, field, subfield_log_bits=4, device=None):
        self.gpu, self.field, self.k = gpu, field, subfield_log_bits
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device

    def encode_rows(self, p, row_lo, row_hi, h_rows):
        """lfgpu_ligero_encode_rows: un-encoded slab image (numpy uint64 [nr][dblock][2]) -> device [nr, block_enc*16]"""
        nr = row_hi - row_lo
        slab = torch.empty((nr, p.block_enc * 16), dtype=torch.uint8, device=self.device)
        if nr:
            h_rows = np.ascontiguousarray(h_rows)
            self.gpu._ck(self.gpu.L.lfgpu_ligero_encode_rows(self.gpu.h, self.field, self.k, C.byref(p), row_lo, row_hi,
                                                             C.c_void_p(h_rows.ctypes.data), C.c_void_p(slab.data_ptr())))
        return slab

    def column_leaves(self, nrow, cols, nonces):
        """lfgpu_column_leaves on a [nrow, ncols*16] column block -> [ncols, 32] digests"""
        ncols = cols.shape[1] // 16
        out = torch.empty((ncols, 32), dtype=torch.uint8, device=self.device)
        if ncols:
            nz = nonces.to(self.device).contiguous()
            self.gpu._ck(self.gpu.L.lfgpu_column_leaves(self.gpu.h, self.field, nrow, ncols, 0, ncols, C.c_void_p(cols.data_ptr()),
                                                        C.c_void_p(nz.data_ptr()), C.c_void_p(out.data_ptr())))
            self.gpu.sync()  # nz may be a temporary
        return out

    def build_tree(self, leaves):
        """lfgpu_merkle_build_tree: [n, 32] leaves -> (root bytes, heap layers tensor [2n, 32])"""
        n = leaves.shape[0]
        layers = torch.zeros((2 * n, 32), dtype=torch.uint8, device=self.device)
        layers[n:] = leaves
        return self.gpu.merkle_build_tree(n, layers.data_ptr()), layers

    def merkle_open(self, n, layers, idx):
        """lfgpu_merkle_open: compressed opening of the leaves idx -> list of 32-byte digests"""
        return self.gpu.merkle_open(n, layers.data_ptr(), list(idx))

    def slab_prover(self, p, row_lo, row_hi, slab, layers, nonces):
        """lfgpu_ligero_prover_from_slab: the prove entry points on this rank's rows (partial sums)"""
        from __graft_entry__ import load_package
        pr = load_package().LigeroProver(self.gpu, self.field, p, self.k)
        h = C.c_void_p()
        nz = (C.c_uint8 * len(nonces)).from_buffer_copy(nonces)
        self.gpu._ck(self.gpu.L.lfgpu_ligero_prover_from_slab(self.gpu.h, self.field, self.k, C.byref(p), row_lo, row_hi,
                                                              C.c_void_p(slab.data_ptr()), C.c_void_p(layers.data_ptr()), nz, C.byref(h)))
        pr.h = h
        pr.rows = row_hi - row_lo
        return pr


def layout_rows(lib, field, k, p, W, subfield_boundary, lqc, rng_bytes, row_lo, row_hi, want_nonces=True):
    """lfgpu_ligero_layout_rows (host only, no device): all RandomEngine draws of LigeroProver::commit in the reference's
    order; returns (numpy uint64 [row_hi-row_lo][dblock][2], nonces bytes or None)"""
    from __graft_entry__ import load_package
    RNG_FN = load_package().RNG_FN

    def cb(_user, buf, n):
        C.memmove(buf, rng_bytes(n), n)

    fn = RNG_FN(cb)
    W = np.ascontiguousarray(W, dtype=np.uint64)
    rows = np.zeros((row_hi - row_lo, p.dblock, 2), dtype=np.uint64)
    nonces = (C.c_uint8 * (32 * p.block_ext))() if want_nonces else None
    lq = (C.c_size_t * max(1, 3 * p.nq))(*[int(x) for x in np.asarray(lqc, dtype=np.uint64).reshape(-1)]) if p.nq else None
    rc = lib.lfgpu_ligero_layout_rows(field, k, C.byref(p), C.c_void_p(W.ctypes.data) if W.size else None, subfield_boundary, lq, fn, None,
                                      row_lo, row_hi, C.c_void_p(rows.ctypes.data) if rows.size else None, nonces)
    if rc != 0:
        raise RuntimeError("lfgpu_ligero_layout_rows failed with code %d" % rc)
    return rows, (bytes(nonces) if want_nonces else None)


class _Replay:
    """a recorded RandomEngine byte stream"""

    def __init__(self, data):
        self.data, self.pos = data, 0

    def bytes(self, n):
        if self.pos + n > len(self.data):
            raise RuntimeError("replayed random stream exhausted")
        b = self.data[self.pos:self.pos + n]
        self.pos += n
        return b


# ------------------------------------------------------------------ column commit
def sharded_column_commit(engine, slab, spans, ld, col0, ncols, nonces, group=None):
    """`slab`: this rank's encoded rows, uint8 tensor [my_rows, ld*16]; spans[q] = (row_lo, row_hi) of rank q
    (contiguous, ascending; an int means an even row_shard of that many rows).
    nonces: uint8 tensor [ncols, 32], identical on every rank.  Returns (root, layers) -- same bytes on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if isinstance(spans, int):
        spans = [(lo, lo + n) for lo, n in (row_shard(spans, q, world) for q in range(world))]
    nrow_total = spans[-1][1]
    my_rows = slab.shape[0]
    # 1. re-partition by columns: send to rank q the columns it owns, for my rows
    send, recv = [], []
    for q in range(world):
        c0, cn = col_shard(ncols, q, world)
        send.append(slab[:, (col0 + c0) * 16:(col0 + c0 + cn) * 16].contiguous())
    mc0, mcn = col_shard(ncols, rank, world)
    for q in range(world):
        recv.append(torch.empty((spans[q][1] - spans[q][0], mcn * 16), dtype=torch.uint8, device=slab.device))
    assert send[rank].shape == recv[rank].shape and my_rows == recv[rank].shape[0]
    _all_to_all(recv, send, group)
    cols = torch.cat(recv, dim=0)  # [nrow_total, mycols*16], rows in global order (slabs are contiguous)
    # 2. local leaves
    my_leaves = engine.column_leaves(nrow_total, cols, nonces[mc0:mc0 + mcn])
    # 3. all_gather digests (ragged: pad to the largest shard)
    maxn = col_shard(ncols, 0, world)[1]
    pad = torch.zeros((maxn, 32), dtype=torch.uint8, device=slab.device)
    pad[:mcn] = my_leaves
    gathered = [torch.empty_like(pad) for _ in range(world)]
    _all_gather(gathered, pad, group)
    leaves = torch.cat([gathered[q][:col_shard(ncols, q, world)[1]] for q in range(world)], dim=0)
    # 4. the tree, on every rank
    return engine.build_tree(leaves)


def ligero_row_shard(p, rank, world):
    """row slab [lo, hi) of a Ligero tableau for `rank`: an even split of the rows, except that the quadratic rows
    [iq, nrow) all go to the last rank (x_i, y_i, z_i of a triple are multiplied element-wise; in the ZK use there are
    3 * ceil(nl / w) of them -- a handful)"""
    lo, cnt = row_shard(p.nrow, rank, world)
    hi = lo + cnt
    if rank == world - 1:
        return min(lo, p.iq), p.nrow
    return min(lo, p.iq), min(hi, p.iq)


class ShardedLigeroProver:
    """LigeroProver<Field, InterpolatorFactory> (reference lib/ligero/ligero_prover.h:34-359) with the tableau rows
    sharded over the ranks of `group`.  Every rank calls every method with the same arguments (SPMD); `rng_bytes` is
    only used on rank 0.  Results (root, y vectors, opened columns) are identical on every rank and identical to the
    single-GPU LigeroProver fed the same RandomEngine stream."""

    def __init__(self, engine, lib, field, param, subfield_log_bits=4, group=None):
        self.e, self.lib, self.field, self.p, self.k, self.group = engine, lib, field, param, subfield_log_bits, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.spans = [ligero_row_shard(param, q, self.world) for q in range(self.world)]
        self.row_lo, self.row_hi = self.spans[self.rank]
        self.pr = None

    def commit(self, W, subfield_boundary, lqc, rng_bytes):
        """LigeroProver::commit (:58-79) without ts.write -> 32-byte root"""
        p = self.p
        # the RandomEngine is one sequential stream: rank 0 draws all of it (layout with an empty slab), everyone replays
        stream = b""
        if self.rank == 0:
            rec = bytearray()

            def tap(n):
                b = rng_bytes(n)
                rec.extend(b)
                return b

            layout_rows(self.lib, self.field, self.k, p, W, subfield_boundary, lqc, tap, 0, 0, want_nonces=False)
            stream = bytes(rec)
        if self.world > 1:
            stream = _broadcast_bytes(stream, 0, self.group)
        rep = _Replay(stream)
        h_rows, nonces = layout_rows(self.lib, self.field, self.k, p, W, subfield_boundary, lqc, rep.bytes, self.row_lo, self.row_hi)
        self.nonces = nonces
        self.slab = self.e.encode_rows(p, self.row_lo, self.row_hi, h_rows)
        nz = torch.frombuffer(bytearray(nonces), dtype=torch.uint8).reshape(p.block_ext, 32)
        self.root, self.layers = sharded_column_commit(self.e, self.slab, self.spans, p.block_enc, p.dblock, p.block_ext, nz, self.group)
        self.pr = self.e.slab_prover(p, self.row_lo, self.row_hi, self.slab, self.layers, nonces)
        return self.root

    # -- prove side: y = T[special row] + sum over the witness / quadratic rows: partial over the slab, all_gather, fold
    def _fold(self, part):
        """part: numpy uint64 [n][2], this rank's partial vector -> the field sum over the ranks"""
        if self.world == 1:
            return part
        dev = "cpu" if _is_gloo(self.group) else self.slab.device
        t = torch.from_numpy(np.ascontiguousarray(part).view(np.uint8).reshape(-1).copy()).to(dev)
        parts = [torch.empty_like(t) for _ in range(self.world)]
        _all_gather(parts, t, self.group)
        acc = parts[0].cpu().numpy().view(np.uint64).reshape(-1, 2).copy()
        for t2 in parts[1:]:
            v = t2.cpu().numpy().view(np.uint64).reshape(-1, 2)
            if self.field == 4:
                acc ^= v
            else:
                tot = [((int(x[0]) | (int(x[1]) << 64)) + (int(y[0]) | (int(y[1]) << 64))) % FP128_P for x, y in zip(acc, v)]
                acc = np.array([[x & (2**64 - 1), x >> 64] for x in tot], dtype=np.uint64)
        return acc

    def low_degree_proof(self, u_ldt):
        """y[block] = T[ildt] + sum_i u_ldt[i] T[iw + i] (ligero_prover.h:281-291)"""
        return self._fold(self.pr.low_degree_proof(u_ldt))

    def dot_proof(self, A):
        """y[dblock] = T[idot] + sum_i RS([0^r | A_i]) (.) T[iw + i] (:293-309); A: numpy uint64 [nwqrow * w][2]"""
        return self._fold(self.pr.dot_proof(A))

    def quadratic_proof(self, u_quad):
        """(:311-344) -> (y_quad_0 [r], y_quad_2 [dblock - block])"""
        y0, y2 = self.pr.quadratic_proof(u_quad)
        return self._fold(y0), self._fold(y2)

    def open(self, idx):
        """compute_req (:346-351) + MerkleCommitment::open (merkle_commitment.h:66-73)
        -> (req numpy uint64 [nrow][nreq][2], nonces uint8 [nreq][32], path digests)"""
        p = self.p
        mine, nz, path = self.pr.open(idx, rows=self.row_hi - self.row_lo)
        maxr = max(hi - lo for lo, hi in self.spans)
        dev = "cpu" if _is_gloo(self.group) else self.slab.device
        pad = torch.zeros((maxr, p.nreq * 16), dtype=torch.uint8)
        pad[:mine.shape[0]] = torch.from_numpy(mine.view(np.uint8).reshape(mine.shape[0], -1))
        pad = pad.to(dev)
        got = [torch.empty_like(pad) for _ in range(self.world)]
        _all_gather(got, pad, self.group)
        req = torch.cat([got[q][:hi - lo] for q, (lo, hi) in enumerate(self.spans)], dim=0)
        return req.cpu().numpy().view(np.uint64).reshape(p.nrow, p.nreq, 2), nz, path

    def close(self):
        if self.pr is not None:
            self.pr.close()
            self.pr = None
