"""Multi-GPU composition of the hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU tests).

What shards and what is exchanged:
  * FFT / RS row encode: rows are independent -> contiguous row slabs per rank, no collective.
  * Merkle column commit: a leaf hashes ALL rows of one column in order, so the encoded slab is
    re-partitioned by columns with one all_to_all (each rank then owns block_ext/N columns x all
    rows), leaves are hashed locally, the 32-byte digests are all_gathered and every rank builds
    the (tiny) tree -> identical root on every rank.  SHA-256 is not a reduction, so there is no
    "all-reduce of roots".
  * sumcheck round: each rank sums its index range; the (a0, a2) pairs are all_gathered and
    folded locally with the FIELD's addition (RCCL has no XOR / mod-p reduce op).
The compute callables are injected, so the same code path runs the HIP kernels on GPUs and the
oracle in the gloo CPU tests.
"""
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


def allgather_fold_partials(field, a0, a2, group=None, device="cpu"):
    """C1: combine per-rank sumcheck partial sums.  a0, a2: (lo, hi) python ints."""
    world = dist.get_world_size(group)

    def enc(v):  # u64 -> i64 two's complement for the int64 tensor
        return v - (1 << 64) if v >= (1 << 63) else v

    mine = torch.tensor([enc(a0[0]), enc(a0[1]), enc(a2[0]), enc(a2[1])], dtype=torch.int64, device=device)
    allp = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allp, mine, group=group)
    s0, s2 = (0, 0), (0, 0)
    for t in allp:
        v = [int(x) & (2**64 - 1) for x in t.cpu().tolist()]
        s0 = field_add(field, s0, (v[0], v[1]))
        s2 = field_add(field, s2, (v[2], v[3]))
    return s0, s2


def sharded_column_commit(slab, nrow_total, col0, ncols, nonces, hash_leaves, build_tree, group=None):
    """C2 (all_to_all form).  `slab`: this rank's encoded rows, uint8 tensor [my_rows, ld*16].
    nonces: uint8 [ncols, 32] (identical on every rank: drawn by the host RandomEngine).
    hash_leaves(cols_all_rows: uint8 [nrow_total, mycols*16], nonces_slice) -> uint8 [mycols, 32]
    build_tree(leaves: uint8 [ncols, 32]) -> 32-byte root.
    Returns the root (same bytes on every rank)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    my_rows = slab.shape[0]
    # 1. re-partition by columns: send to rank q the columns it owns, for my rows
    send, recv = [], []
    for q in range(world):
        c0, cn = col_shard(ncols, q, world)
        send.append(slab[:, (col0 + c0) * 16:(col0 + c0 + cn) * 16].contiguous())
    mc0, mcn = col_shard(ncols, rank, world)
    for q in range(world):
        _, rn = row_shard(nrow_total, q, world)
        recv.append(torch.empty((rn, mcn * 16), dtype=torch.uint8, device=slab.device))
    assert send[rank].shape == recv[rank].shape and my_rows == recv[rank].shape[0]
    dist.all_to_all(recv, send, group=group) if dist.get_backend(group) != "gloo" else _gloo_all_to_all(recv, send, group)
    cols = torch.cat(recv, dim=0)  # [nrow_total, mycols*16], rows in global order (slabs are contiguous)
    # 2. local leaves
    my_leaves = hash_leaves(cols, nonces[mc0:mc0 + mcn])
    # 3. all_gather digests (ragged: pad to the largest shard)
    maxn = col_shard(ncols, 0, world)[1]
    pad = torch.zeros((maxn, 32), dtype=torch.uint8, device=slab.device)
    pad[:mcn] = my_leaves
    gathered = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(gathered, pad, group=group)
    leaves = torch.cat([gathered[q][:col_shard(ncols, q, world)[1]] for q in range(world)], dim=0)
    return build_tree(leaves)


def _gloo_all_to_all(recv, send, group):
    """gloo has no all_to_all: emulate with pairwise send/recv (CPU tests only)"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    recv[rank].copy_(send[rank])
    reqs = []
    for q in range(world):
        if q != rank:
            reqs.append(dist.isend(send[q], q, group=group))
            reqs.append(dist.irecv(recv[q], q, group=group))
    for r in reqs:
        r.wait()
