"""Multi-GPU product path with the HIP kernels in the loop.

  * world 1, in process: ShardedLigeroProver over GpuEngine == the single-GPU LigeroProver (lfgpu_ligero_commit etc.)
    on the same RandomEngine stream, and == the reference's C++ commitment root for the GF2_128 vector.
  * world 2 / 3, one process per rank sharing this box's single GPU: the same kernels, slabs and collectives as on a
    multi-GPU node; the transport is gloo with host staging because RCCL cannot put two ranks on one device.  (The
    RCCL transport itself is exercised by bench.py --gpus N on the multi-GPU node.)"""
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
        par = importlib.import_module("longfellow_zk_amd.parallel")
        torch.cuda.set_device(0)
        res = su.run_rank(G.pkg, par, par.GpuEngine(G.gpu(), field), field, None, solo)
        _compare(res, _single_gpu_results(G.pkg, G.gpu(), field, res))
        q.put((rank, "ok"))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "FAIL: %r\n%s" % (e, traceback.format_exc())))


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
    par = importlib.import_module("longfellow_zk_amd.parallel")
    res = su.run_rank(G.pkg, par, par.GpuEngine(G.gpu(), field), field, None, world1)
    _compare(res, _single_gpu_results(G.pkg, G.gpu(), field, res))


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
