"""Multi-GPU product path with the HIP kernels in the loop.

The product is lfgpu_ligero_commit_sharded + the prove entry points behind the C ABI (csrc/ligero.hip), reached through
parallel.ShardedLigeroProver with the transport hooks bound to torch.distributed (parallel.TorchComm), and
lfgpu_zk_prover_set_comm for the whole prover.  Each case compares it with (a) the same orchestration restated in Python over
kernel-level calls (tests/sharded_reference.py), (b) the one-GPU LigeroProver on the same RandomEngine stream, and (c) for
GF2_128 the reference's C++ commitment root.

  * world 1, in process;
  * world 2 / 3, one process per rank sharing this box's single GPU: the same kernels, slabs and collectives as on a
    multi-GPU node; the transport is gloo with host staging because RCCL cannot put two ranks on one device.  (The
    RCCL transport itself is exercised by bench.py --gpus N on the multi-GPU node and bench.py --force-dist here.)
  * the ZK prover with a communicator (world 2): flatsha256 proofs with the tableau rows sharded, and replicated below the
    threshold -- wire bytes of the reference on every rank."""
import os
import socket

import numpy as np
import pytest

from oracle_lib import FP, GF

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    solo = None
    for r in range(world):
        g = dist.new_group([r])
        if r == rank:
            solo = g
    return solo


def _single_gpu_results(pkg, gpu, field, res):
    """the one-GPU LigeroProver (lfgpu_ligero_commit / *_proof / open) on the same statement and RandomEngine"""
    import ligero_fixture as lf
    pr = pkg.LigeroProver(gpu, field, res["p"])
    out = {"root": pr.commit(res["W"], res["sfb"], res["lqc"], lf.LcgRng(res["seed"]).bytes)}
    out["y_ldt"] = pr.low_degree_proof(res["u"])
    out["y_dot"] = pr.dot_proof(res["A"])
    out["y_q0"], out["y_q2"] = pr.quadratic_proof(res["uq"])
    out["req"], out["nonces"], out["path"] = pr.open(res["idx"])
    pr.close()
    return out


def _compare(res, one):
    assert res["root"] == one["root"] and res["path"] == one["path"]
    for key in ("y_ldt", "y_dot", "y_q0", "y_q2", "req", "nonces"):
        assert (res[key] == one[key]).all(), key


def _worker(rank, world, port, field, q):
    try:
        import torch
        import torch.distributed as dist
        solo = _init(rank, world, port)
        import importlib
        import gpu_util as G
        import sharded_util as su
        import sharded_reference as sref
        par = importlib.import_module("longfellow_zk_amd.parallel")
        torch.cuda.set_device(0)
        res = su.run_rank(G.pkg, par, sref.GpuEngine(G.gpu(), field), field, None, solo)
        one = _single_gpu_results(G.pkg, G.gpu(), field, res)
        _compare(res, one)
        _compare(su.run_rank_c(G.pkg, par, G.gpu(), field, None), one)  # the C entry point over the same ranks
        if field == GF:
            _zk_with_comm(G.pkg, G.gpu(), par, rank)
        q.put((rank, "ok"))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %r\n%s" % (e, traceback.format_exc())))


def _zk_with_comm(pkg, gpu, par, rank):
    """lfgpu_zk_prover_set_comm: flatsha256 x 1 block, rows sharded (threshold 0) and replicated (huge threshold); only rank 0's
    RandomEngine is real -- the other ranks' engines would give other bytes if they were drawn from"""
    import hashlib
    import json
    import lzma
    import ligero_fixture as lf
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    raw = lzma.decompress(open(os.path.join(gold, "flatsha_nb1.lfc1.xz"), "rb").read())
    W = np.frombuffer(lzma.decompress(open(os.path.join(gold, "flatsha_nb1.w.xz"), "rb").read()), dtype=np.uint64).reshape(-1, 2).copy()
    info = json.load(open(os.path.join(gold, "flatsha_nb1.json")))
    circ = pkg.Circuit(gpu, raw)
    zk = pkg.ZkProver(gpu, circ, 7, 132)
    comm = par.TorchComm(None)
    for thr in (0, 1 << 40):
        zk.set_comm(comm, thr)
        ts = pkg.FsTranscript(b"test")
        zk.commit(W, lf.LcgRng(100 if rank == 0 else 4242 + rank).bytes, ts)
        assert zk.prove(W, ts), comm.error
        wire = zk.wire()
        ts.close()
        assert len(wire) == info["zk_wire_bytes"] and hashlib.sha256(wire).hexdigest() == info["zk_wire_sha256"], ("threshold", thr, comm.error)
    zk.close()
    circ.close()
    # Fp256Base (the mdoc signature circuit): one RandomEngine (rank 0's) behind the communicator
    raw = lzma.decompress(open(os.path.join(gold, "mdoc_sig.lfc1.xz"), "rb").read())
    W = np.frombuffer(lzma.decompress(open(os.path.join(gold, "mdoc_sig.w.xz"), "rb").read()), dtype=np.uint64).reshape(-1, 4).copy()
    info = json.load(open(os.path.join(gold, "mdoc.json")))["sig"]
    circ = pkg.Circuit(gpu, raw)
    zk = pkg.ZkProver(gpu, circ, 7, 132, info["block_enc"])
    for thr in (0, 1 << 40):  # 19 rows of 32-byte elements sharded over the ranks (lig256_commit), then replicated
        zk.set_comm(comm, thr)
        ts = pkg.FsTranscript(b"test")
        zk.commit(W, lf.LcgRng(100 if rank == 0 else 777 + rank).bytes, ts)
        assert zk.prove(W, ts), comm.error
        wire = zk.wire()
        ts.close()
        assert len(wire) == info["zk_wire_bytes"] and hashlib.sha256(wire).hexdigest() == info["zk_wire_sha256"], ("p256 threshold", thr, comm.error)
    zk.close()
    circ.close()


@pytest.fixture(scope="module")
def world1():
    import torch.distributed as dist
    solo = _init(0, 1, _free_port())
    yield solo
    dist.destroy_process_group()


@pytest.mark.parametrize("field", [GF, FP])
def test_sharded_ligero_world1_equals_single_gpu(world1, field):
    import importlib
    import gpu_util as G
    import sharded_util as su
    import sharded_reference as sref
    par = importlib.import_module("longfellow_zk_amd.parallel")
    res = su.run_rank(G.pkg, par, sref.GpuEngine(G.gpu(), field), field, None, world1)
    one = _single_gpu_results(G.pkg, G.gpu(), field, res)
    _compare(res, one)
    _compare(su.run_rank_c(G.pkg, par, G.gpu(), field, world1), one)


@pytest.mark.parametrize("field,world", [(GF, 2), (FP, 2), (GF, 3)])
def test_sharded_ligero_ranks_share_one_gpu(field, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, field, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p_ in procs:
        p_.join(timeout=120)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
