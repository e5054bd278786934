"""Every sumcheck round driver and tuning switch must produce the reference's proof bytes (DESIGN.md section 4,
"Driving the rounds").  The switches are read once per process, hence one child process per combination."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "zk_mode_child.py")

CASES = [
    ("1", {"LFGPU_SC_MODE": "grid"}),
    ("1", {"LFGPU_SC_MODE": "resident"}),
    ("1", {"LFGPU_SC_MODE": "launch"}),
    ("1", {"LFGPU_SC_MODE": "off"}),
    ("1", {"LFGPU_SC_SPLIT": "0", "LFGPU_SC_TAIL": "0"}),
    ("1", {"LFGPU_SC_PER_WG": "64", "LFGPU_SC_WGS": "128"}),
    ("1", {"LFGPU_SC_GRID_MAX": "4096", "LFGPU_SC_PER_WG": "2048"}),
    ("32", {"LFGPU_SC_GRID_MAX": "262144", "LFGPU_SC_WGS": "128", "LFGPU_SC_PER_WG": "256"}),
    ("32", {"LFGPU_SC_WGS": "7", "LFGPU_SC_PER_WG": "1024"}),
    ("1 fp128", {"LFGPU_SC_PER_WG": "128", "LFGPU_SC_SPLIT": "0"}),
    ("1 fp128", {"LFGPU_SC_MODE": "resident"}),
    # the single-wave tail (<= 64 entries; on by default, so every case above runs through it): off, for both fields,
    # and on without the LDS tail it lives in (then it must not engage)
    ("1", {"LFGPU_SC_WAVE_TAIL": "0"}),
    ("1 fp128", {"LFGPU_SC_WAVE_TAIL": "0"}),
    ("32", {"LFGPU_SC_WAVE_TAIL": "1", "LFGPU_SC_TAIL": "0"}),
    # the recorded bind offsets of the grid (second proof of every case above replays them): never recorded
    ("32", {"LFGPU_SC_OFFCACHE": "0"}),
    # Fp256Base (csrc/zk256.hip) on the mdoc signature circuit: without the single-wave tail, without the resident grid
    # (three launches per round-hand all the way), and with other hand-off points / workgroup shares
    ("sig", {"LFGPU_P256_WAVE_TAIL": "0"}),
    ("sig", {"LFGPU_P256_GRID": "0"}),
    ("sig", {"LFGPU_P256_GRID_MAX": "65536", "LFGPU_P256_PER_WG": "128"}),
    ("sig", {"LFGPU_P256_GRID_MAX": "300", "LFGPU_P256_PER_WG": "2048"}),
    # the per-device CU budget of the resident kernels (csrc/ctx.h): none at all (every round-hand on per-launch kernels), and
    # far less than the grid asks for at its hand-off point (it starts some round-hands later, when it fits)
    ("32", {"LFGPU_CU_BUDGET": "0"}),
    ("32", {"LFGPU_CU_BUDGET": "5"}),
    ("1 fp128", {"LFGPU_CU_BUDGET": "1"}),
    ("sig", {"LFGPU_CU_BUDGET": "0"}),
    ("sig", {"LFGPU_CU_BUDGET": "2", "LFGPU_P256_GRID_MAX": "65536", "LFGPU_P256_PER_WG": "128"}),
    # a grid that is not placed whole (test hook: its last workgroup never shows up): the first barrier times out, the kernel
    # reports it, and the layer continues on per-launch kernels with identical bytes; after two strikes the context stops
    # launching grids
    ("1", {"LFGPU_SC_TEST_DROP": "1", "LFGPU_SC_PLACE_MS": "20"}),
    ("32", {"LFGPU_SC_TEST_DROP": "5", "LFGPU_SC_PLACE_MS": "20"}),
    ("1 fp128", {"LFGPU_SC_TEST_DROP": "1", "LFGPU_SC_PLACE_MS": "20"}),
    ("sig", {"LFGPU_P256_TEST_DROP": "1", "LFGPU_SC_PLACE_MS": "20", "LFGPU_P256_GRID_MAX": "65536", "LFGPU_P256_PER_WG": "128"}),
    # dispatch-saving paths: the EQ factor tables and their combination in ONE launch (default only with >= 6
    # provers on the device: forced here, both fields, and forced off), Dense::bind + HQuad::bind_h in two launches instead of
    # one, and both with the per-launch kernels all the way (every round-hand through bind_both_kernel from the second proof on)
    ("1", {"LFGPU_EQ_FUSED": "1"}),
    ("32", {"LFGPU_EQ_FUSED": "1"}),
    ("1 fp128", {"LFGPU_EQ_FUSED": "1"}),
    ("32", {"LFGPU_EQ_FUSED": "0", "LFGPU_SC_BIND_SPLIT": "1"}),
    ("32", {"LFGPU_SC_MODE": "off", "LFGPU_EQ_FUSED": "1"}),
    ("1 fp128", {"LFGPU_SC_MODE": "off"}),
    ("32", {"LFGPU_SC_MODE": "off", "LFGPU_SC_BIND_SPLIT": "1"}),
]


@pytest.mark.gpu
@pytest.mark.parametrize("args,env", CASES, ids=[a + " " + ",".join("%s=%s" % kv for kv in e.items()) for a, e in CASES])
def test_proof_bytes_under_every_driver(args, env):
    e = dict(os.environ)
    for k in list(e):
        if k.startswith("LFGPU_SC_") or k.startswith("LFGPU_P256_") or k.startswith("LFGPU_CU_") or k.startswith("LFGPU_EQ_"):
            del e[k]
    e.update(env)
    r = subprocess.run([sys.executable, CHILD] + args.split(), env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
