"""helpers for the -m gpu parity tests: torch is used only for device memory."""
import numpy as np
import torch

import os

import __graft_entry__ as _ge
from __graft_entry__ import load_package

if not os.path.exists(_ge.LIB):  # fresh checkout on a GPU box: build the HIP library in-tree (hipcc is in the image)
    _ge.build()
pkg = load_package()
_gpu = None


def gpu():
    global _gpu
    if _gpu is None:
        _gpu = pkg.LfGpu(0)
        _gpu.set_stream(torch.cuda.current_stream().cuda_stream)
    return _gpu


def to_dev(a):
    """numpy array (any dtype) -> flat device byte tensor holding the same bytes"""
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda()


def from_dev(t, dtype, shape):
    torch.cuda.synchronize()
    return t.cpu().numpy().view(dtype).reshape(shape)
