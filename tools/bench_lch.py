"""quick A/B timing of the LCH14 FFT paths on the GPU (not the headline bench)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
rows, l, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
gpu = pkg.LfGpu(0)
gpu.set_stream(torch.cuda.current_stream().cuda_stream)
A = torch.randint(-2**63, 2**63 - 1, (rows << l, 2), dtype=torch.int64, device="cuda")
gpu.gf2128_lch14_fft(A.data_ptr(), rows, l, subfield_log_bits=k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    gpu.gf2128_lch14_fft(A.data_ptr(), rows, l, subfield_log_bits=k)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("rows %d l %d k %d RLOG=%s BS=%s: %.2f ms  %.2f G elem/s" % (rows, l, k, os.environ.get("LFGPU_BS_RLOG"), os.environ.get("LFGPU_LCH_BS"), ms, (rows << l) / ms / 1e6))
