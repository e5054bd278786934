"""step-by-step probe of the S-lig shape (bench.py ligero_commit_slig) with a progress line after every device step"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from __graft_entry__ import load_package
pkg = load_package()
gpu = pkg.LfGpu(0)
gpu.set_stream(torch.cuda.current_stream().cuda_stream)
out = open(sys.argv[1], "a")
def say(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for rows in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "64,256,1024").split(",")]:
    be = 1 << logn
    block = (be + 1) // 6
    dblock = 2 * block - 1
    ext = be - dblock
    A = torch.randint(-2**63, 2**63 - 1, (rows * be, 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize(); say("rows", rows, "alloc ok")
    t0 = time.perf_counter()
    gpu.gf2128_rs_encode_rows(A.data_ptr(), rows, block, be, subfield_log_bits=5)
    torch.cuda.synchronize(); say("rows", rows, "rs ok", round((time.perf_counter() - t0) * 1e3, 1), "ms (first call)")
    t0 = time.perf_counter()
    gpu.gf2128_rs_encode_rows(A.data_ptr(), rows, block, be, subfield_log_bits=5)
    torch.cuda.synchronize(); say("rows", rows, "rs ok", round((time.perf_counter() - t0) * 1e3, 1), "ms")
    nonces = torch.randint(0, 256, (ext, 32), dtype=torch.uint8, device="cuda")
    layers = torch.zeros(2 * ext * 32, dtype=torch.uint8, device="cuda")
    t0 = time.perf_counter()
    root = gpu.column_commit(4, rows, be, dblock, ext, A.data_ptr(), nonces.data_ptr(), layers.data_ptr())
    torch.cuda.synchronize(); say("rows", rows, "commit ok", round((time.perf_counter() - t0) * 1e3, 1), "ms", root.hex()[:16])
    del A, nonces, layers
    torch.cuda.empty_cache()
say("done")
