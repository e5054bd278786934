"""The integration of INTEGRATION.md section 2a, executed: the reference's OWN ZkProver / LigeroProver / sumcheck prover /
transcript / ZkProof::write, compiled from /root/reference in the build container with one template argument swapped --
InterpolatorFactory = lfgpu::GpuReedSolomonFactory<Field> (include/lfgpu_adapters.h) -- and linked against liblfgpu.so
(oracle/ref_zk_adapters.cc -> oracle/_ref/zk_adapters[_fp], built by `make -C oracle ref`).  On the GPU box the binary
proves the fixture circuits with every Reed-Solomon row extension running in the HIP kernels; its wire bytes must hash to
what the unmodified reference produced (tests/golden/flatsha_*.json)."""
import json
import lzma
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.gpu
@pytest.mark.parametrize("stem,binary", [("flatsha_nb1", "zk_adapters"), ("flatsha_nb32", "zk_adapters"), ("flatsha_nb33", "zk_adapters"),
                                         ("flatsha_fp_nb1", "zk_adapters_fp")])
def test_reference_zkprover_with_gpu_interpolator_emits_reference_wire_bytes(stem, binary):
    exe = os.path.join(ROOT, "oracle", "_ref", binary)
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/%s not built (needs the reference sources: make -C oracle ref in the build container)" % binary)
    info = json.load(open(os.path.join(GOLD, stem + ".json")))
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for ext in (".lfc1", ".w"):
            p = os.path.join(td, "x" + ext)
            with open(p, "wb") as f:
                f.write(lzma.decompress(open(os.path.join(GOLD, stem + ext + ".xz"), "rb").read()))
            paths.append(p)
        out = subprocess.run([exe] + paths, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    res = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert res["wire_bytes"] == info["zk_wire_bytes"]
    assert res["wire_sha256"] == info["zk_wire_sha256"]
    assert (res["block_enc"], res["nrow"]) == (info["zk_block_enc"], info["zk_nrow"])
