"""N > 1 path on CPU: world_size 2 (and 3), gloo.

* The product's transport hooks (longfellow-zk_amd/parallel.py: TorchComm = lfgpu_comm_ops over torch.distributed) through the
  library's own host-only self-test (lfgpu_comm_selftest: ragged all_to_all, every broadcast root, all_gather), and the host
  half of the sharded commit behind the C ABI (lfgpu_ligero_layout_rows_sharded: rank 0 draws, the stream is broadcast, every
  rank replays): the slabs must be the rows of the whole layout.
* The orchestration restated in Python over an engine backed by the oracle (tests/sharded_reference.py,
  tests/sharded_util.py) -- the checker the GPU tests compare lfgpu_ligero_commit_sharded with: results must equal the
  one-rank run and, for GF2_128, the commitment root of the reference's own C++ Ligero vector."""
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
    pkg = load_package()
    import importlib
    par = importlib.import_module("longfellow_zk_amd.parallel")
    import sharded_reference as sref
    import sharded_util as su
    o = ol.oracle()
    try:
        # --- the product's hooks over this process group (host buffers), and the host half of the sharded commit
        comm = par.TorchComm(None)
        comm.selftest()
        import ligero_fixture as lf
        p_, W_, sfb_, lqc_, seed_, _ = su.statement(pkg, field)
        lib = pkg.load_library()
        whole, nz = par.layout_rows(lib, field, 4, p_, W_, sfb_, lqc_, lf.LcgRng(seed_).bytes, 0, p_.nrow)
        mine, nz1 = par.layout_rows_sharded(lib, field, 4, p_, W_, sfb_, lqc_, lf.LcgRng(seed_).bytes, comm)
        lo, hi = par.ligero_row_shard(p_, rank, world)
        assert (mine == whole[lo:hi]).all() and nz1 == nz, "sharded host layout differs from the whole layout"
        solo = None
        for r in range(world):  # new_group is collective: every rank creates every one-rank group
            g = dist.new_group([r])
            if r == rank:
                solo = g
        # --- sharded LigeroProver: commit (layout -> encode -> all_to_all -> leaves -> all_gather -> tree) + prove
        res = su.run_rank(pkg, par, su.OracleEngine(field), field, None, solo)
        assert res["spans"][0][0] == 0 and res["spans"][-1][1] == res["p"].nrow
        # --- the column commit alone on a ragged shape
        rng = np.random.default_rng(123)
        nrow, ld, col0, ncols = 11, 96, 29, 67
        T = ol.rand_elts(rng, nrow * ld, field).reshape(nrow, ld, 2)
        nonces = rng.integers(0, 256, size=(ncols, 32), dtype=np.uint8)
        r0, rn = par.row_shard(nrow, rank, world)
        slab = torch.from_numpy(T[r0:r0 + rn].copy().view(np.uint8).reshape(rn, ld * 16))
        root, _layers = sref.sharded_column_commit(su.OracleEngine(field), slab, nrow, ld, col0, ncols, torch.from_numpy(nonces))
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
        import traceback
        q.put((rank, "FAIL: %r\n%s" % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("field,world", [(GF, 2), (FP, 2), (GF, 3)])
def test_world_gloo(field, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, field, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_row_shard_partition():
    from __graft_entry__ import load_package
    pkg = load_package()
    import importlib
    par = importlib.import_module("longfellow_zk_amd.parallel")
    for n in (0, 1, 7, 150, 1024):
        for w in (1, 2, 3, 8):
            spans = [par.row_shard(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
    # Ligero slabs: contiguous, cover [0, nrow), and the quadratic rows [iq, nrow) sit on the last rank
    for (nw, nq, be) in ((1000, 50, 4096), (700, 40, 512), (100, 3, 256), (5000, 2000, 1024)):
        p = pkg.ligero_param(GF, nw, nq, 4, 12, be)
        for w in (1, 2, 3, 8):
            spans = [par.ligero_row_shard(p, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == p.nrow
            for (_, h0), (l1, _) in zip(spans, spans[1:]):
                assert h0 == l1
            assert spans[-1][0] <= p.iq


def test_layout_rows_host_slabs_match_whole():
    """lfgpu_ligero_layout_rows (host only): slabs cut out of the stream equal the rows of the whole layout, and the
    draw count does not depend on the slab"""
    from __graft_entry__ import load_package
    pkg = load_package()
    import importlib
    par = importlib.import_module("longfellow_zk_amd.parallel")
    import ligero_fixture as lf
    import sharded_util as su
    for field in (GF, FP):
        p, W, sfb, lqc, seed, _ = su.statement(pkg, field)
        lib = pkg.load_library()
        r0 = lf.LcgRng(seed)
        whole, nz = par.layout_rows(lib, field, 4, p, W, sfb, lqc, r0.bytes, 0, p.nrow)
        for lo, hi in ((0, 0), (0, 3), (2, 5), (p.iq, p.nrow), (p.nrow - 1, p.nrow)):
            r1 = lf.LcgRng(seed)
            part, nz1 = par.layout_rows(lib, field, 4, p, W, sfb, lqc, r1.bytes, lo, hi)
            assert (part == whole[lo:hi]).all() and nz1 == nz and r1.state == r0.state


def test_comm_selftest_catches_a_broken_hook():
    """lfgpu_comm_selftest is what a C++ host runs on its RCCL binding before trusting it: a one-rank communicator whose
    all_gather copies nothing must be rejected, a correct one accepted (no process group needed: the hooks are plain callbacks)"""
    import ctypes as C
    from __graft_entry__ import load_package
    pkg = load_package()
    import importlib
    par = importlib.import_module("longfellow_zk_amd.parallel")
    lib = pkg.load_library()

    def ag_ok(_u, send, recv, n, _dev, _s):
        C.memmove(recv, send, n)
        return 0

    def ag_bad(_u, send, recv, n, _dev, _s):
        return 0  # claims success, moves nothing

    def a2a(_u, send, so, sb, recv, ro, rb, _dev, _s):
        C.memmove(recv + ro[0], send + so[0], sb[0])
        return 0

    def bc(_u, buf, n, root, _dev, _s):
        return 0

    for ag, want in ((ag_ok, 0), (ag_bad, 5)):  # 5 = LFGPU_ERR_ASSERT
        fns = (par.AG_FN(ag), par.A2A_FN(a2a), par.BC_FN(bc))
        ops = par.CommOps(None, 0, 1, *fns)
        assert lib.lfgpu_comm_selftest(C.byref(ops)) == want
    ops = par.CommOps(None, 3, 2, *fns)  # rank outside the world
    assert lib.lfgpu_comm_selftest(C.byref(ops)) == 1  # LFGPU_ERR_ARG
