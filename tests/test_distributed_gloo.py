"""N > 1 path on CPU: world_size 2, gloo.  The collectives and the re-partitioning logic of
longfellow-zk_amd/parallel.py are exercised with the oracle as the injected compute, and the
result must equal the single-process oracle (root / folded partial sums)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as ol
from oracle_lib import FP, GF, P


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, field, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from __graft_entry__ import load_package
    load_package()
    import importlib
    par = importlib.import_module("longfellow_zk_amd.parallel")
    o = ol.oracle()
    try:
        # --- shared synthetic statement (same seed on every rank)
        rng = np.random.default_rng(123)
        nrow, ld, col0, ncols = 11, 96, 29, 67
        T = ol.rand_elts(rng, nrow * ld, field).reshape(nrow, ld, 2)
        nonces = rng.integers(0, 256, size=(ncols, 32), dtype=np.uint8)
        r0, rn = par.row_shard(nrow, rank, world)
        slab = torch.from_numpy(T[r0:r0 + rn].copy().view(np.uint8).reshape(rn, ld * 16))

        def hash_leaves(cols, nz):
            a = np.ascontiguousarray(cols.numpy()).view(np.uint64).reshape(nrow, -1, 2)
            n = a.shape[1]
            out = np.zeros((n, 32), dtype=np.uint8)
            o.lfo_column_leaves(field, nrow, n, 0, n, P(np.ascontiguousarray(a)), P(np.ascontiguousarray(nz.numpy())), P(out))
            return torch.from_numpy(out)

        def build_tree(leaves):
            lv = np.ascontiguousarray(leaves.numpy())
            lay = np.zeros((2 * len(lv), 32), dtype=np.uint8)
            o.lfo_merkle_build_tree(len(lv), P(lv), P(lay))
            return lay[1].tobytes()

        root = par.sharded_column_commit(slab, nrow, col0, ncols, torch.from_numpy(nonces), hash_leaves, build_tree)
        want = np.zeros(32, dtype=np.uint8)
        o.lfo_column_commit(field, nrow, ld, col0, ncols, P(T), P(nonces), P(want), None)
        assert root == want.tobytes(), "sharded root differs"

        # --- sumcheck partial fold: each rank sums an even-aligned index range
        n = 1000
        QW, W = ol.rand_elts(rng, n, field), ol.rand_elts(rng, n, field)
        half = (n // 2 // world) * 2
        lo = rank * half
        hi = n if rank == world - 1 else lo + half
        a0, a2 = ol.Elt(), ol.Elt()
        o.lfo_sumcheck_partials(field, hi - lo, P(np.ascontiguousarray(QW[lo:hi])), P(np.ascontiguousarray(W[lo:hi])),
                                C.byref(a0), C.byref(a2))
        s0, s2 = par.allgather_fold_partials(field, (a0.l[0], a0.l[1]), (a2.l[0], a2.l[1]))
        w0, w2 = ol.Elt(), ol.Elt()
        o.lfo_sumcheck_partials(field, n, P(QW), P(W), C.byref(w0), C.byref(w2))
        assert s0 == (w0.l[0], w0.l[1]) and s2 == (w2.l[0], w2.l[1]), "folded partials differ"
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("field", [GF, FP])
def test_world2_gloo(field):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, field, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_row_shard_partition():
    from __graft_entry__ import load_package
    load_package()
    import importlib
    par = importlib.import_module("longfellow_zk_amd.parallel")
    for n in (0, 1, 7, 150, 1024):
        for w in (1, 2, 3, 8):
            spans = [par.row_shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
