"""Throughput mode ("independent proofs are trivially parallel"): K provers in K host threads on ONE
device -- one lfgpu context with its own stream each (lfgpu_own_stream), one copy of the circuit in HBM (lfgpu_circuit_share)
-- must every one produce the reference's wire bytes, and the resident sumcheck kernels must stay co-resident under it: the
per-device CU budget (csrc/ctx.h, lf_cu_acquire) is what guarantees that, exercised here with 8 provers of the 32-block
circuit (up to 64 workgroups per grid each) and with a budget far below what they ask for.

The reference's loop: lib/circuits/sha/flatsha256_circuit_test.cc:510-536 (one proof after the other on one core)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "zk_throughput.py")


def _run(args, env_extra, timeout=600):
    e = dict(os.environ)
    for k in list(e):
        if k.startswith("LFGPU_"):
            del e[k]
    e.update(env_extra)
    r = subprocess.run([sys.executable, TOOL] + args, env=e, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_eight_concurrent_provers_emit_reference_wire_bytes():
    """8 threads x (parity proof with the fixtures' engine + timed proofs) on flatsha-32; every worker asserts the wire SHA-256"""
    res = _run(["--jobs", "flatsha32", "--k", "8", "--seconds", "1.0"], {})
    k8 = res["flatsha32"]["k"]["8"]
    assert k8["proofs"] >= 8, k8


@pytest.mark.gpu
def test_concurrent_mdoc_pairs_emit_reference_wire_bytes():
    """4 threads, each proving the mdoc pair (GF2_128 hash circuit + Fp256Base signature circuit: both resident-grid kernels live at once)"""
    res = _run(["--jobs", "mdoc", "--k", "4", "--seconds", "1.0"], {})
    assert res["mdoc"]["k"]["4"]["proofs"] >= 4


@pytest.mark.gpu
@pytest.mark.parametrize("budget", ["0", "3", "70"])
def test_concurrent_provers_with_a_short_cu_budget(budget):
    """budget 0: no resident kernel ever; 3 / 70: the grids wait for room or start late and small -- same bytes, no timeout"""
    res = _run(["--jobs", "flatsha32", "--k", "4", "--seconds", "0.5"], {"LFGPU_CU_BUDGET": budget})
    assert res["flatsha32"]["k"]["4"]["proofs"] >= 4
