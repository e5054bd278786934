"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/lfgpu.h declares (no compute without a GPU), error behaviour of argument checks that
need no device, and (build container only) include/lfgpu_adapters.h instantiates the
reference's own LigeroProver template."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

from __graft_entry__ import ROOT, build, load_package


@pytest.fixture(scope="module")
def pkg():
    if not os.path.exists(os.path.join(ROOT, "longfellow-zk_amd", "liblfgpu.so")):
        if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
            pytest.skip("liblfgpu.so not built and no hipcc")
        build()
    return load_package()


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.load_library()
    hdr = open(os.path.join(ROOT, "include", "lfgpu.h")).read() + open(os.path.join(ROOT, "include", "lfgpu_zk.h")).read()
    declared = sorted(set(re.findall(r"\b(lfgpu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "missing export " + name
    for name in pkg.ABI_SYMBOLS:
        assert name in declared


def test_ligero_param_matches_reference_layout(pkg):
    """LigeroParam::layout (ligero_param.h:185-295) values measured on the reference"""
    p = pkg.ligero_param(pkg.FIELD_GF2_128, 4368, 13, 7, 132, 4096)
    assert (p.block, p.dblock, p.block_ext, p.r, p.w, p.nrow) == (455, 909, 3187, 132, 323, 20)
    p = pkg.ligero_param(pkg.FIELD_GF2_128, 111760, 0, 7, 132, 8192)
    assert (p.block, p.dblock, p.block_ext, p.w, p.nrow) == (910, 1819, 6373, 778, 147)
    p = pkg.ligero_param(pkg.FIELD_GF2_128, 1000, 50, 4, 36, 4096)
    assert (p.block, p.dblock, p.nrow, p.r, p.w, p.block_ext, p.nqtriples, p.nwrow) == (682, 1363, 8, 36, 646, 2733, 1, 2)
    with pytest.raises(pkg.LfGpuError):  # block_enc >= 2^16 is rejected for GF2_128<4> (ligero_param.h:197-202)
        pkg.ligero_param(pkg.FIELD_GF2_128, 1000, 0, 4, 36, 1 << 16, subfield_log_bits=4)
    with pytest.raises(pkg.LfGpuError):  # block < r
        pkg.ligero_param(pkg.FIELD_GF2_128, 1000, 0, 4, 36, 64)


def test_no_cpu_fallback_without_gpu(pkg):
    """the product path must fail loudly when no GPU is usable (never route through the oracle)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.LfGpuError):
        pkg.LfGpu(0)


@pytest.mark.ref
@pytest.mark.skipif(not os.path.isdir("/root/reference/lib"), reason="reference checkout not present")
def test_adapters_instantiate_reference_ligero_prover(tmp_path):
    cmd = ["g++", "-std=c++17", "-O0", "-mpclmul", "-DOPENSSL_SUPPRESS_DEPRECATED=1", "-Wno-deprecated-declarations",
           "-Wno-ignored-attributes", "-I/root/reference/lib", "-I" + os.path.join(ROOT, "include"), "-c",
           os.path.join(ROOT, "tests", "adapters_compile_check.cc"), "-o", str(tmp_path / "acc.o")]
    subprocess.check_call(cmd)
