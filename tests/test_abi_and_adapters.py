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


def test_post_publish_waits_for_payload(tmp_path):
    """The resident kernels post {payload words ..., sequence word} to pinned host memory with relaxed system-scope stores; the
    host acquires on the sequence word and then reads the payload, so the payload must have left the CU first.  Pins the ISA:
    between the last payload store and the sequence store hipcc must emit `s_waitcnt vmcnt(0)` (csrc/fields.h,
    lf_wait_stores_before_publish) -- a workgroup-scope release fence, which the code used before, emits none."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "post.hip"
    src.write_text('#include "%s"\n' % os.path.join(ROOT, "longfellow-zk_amd", "csrc", "fields.h") + r'''
__global__ void post_kernel(u64* po, u64 a, u64 b, u64 seq) {
  __hip_atomic_store(&po[0], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&po[9], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  lf_wait_stores_before_publish();
  __hip_atomic_store(&po[16], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
''')
    out = tmp_path / "post.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), str(src)])
    body = out.read_text().split("post_kernel", 1)[1]
    ins = [l.strip() for l in body.splitlines() if l.strip().startswith(("global_store", "flat_store", "s_waitcnt"))]
    stores = [i for i, l in enumerate(ins) if "_store" in l]
    assert len(stores) >= 3, ins
    between = ins[stores[-2] + 1:stores[-1]]
    assert any(l.startswith("s_waitcnt") and "vmcnt(0)" in l for l in between), ins
