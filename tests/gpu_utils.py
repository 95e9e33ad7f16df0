"""Helpers for the -m gpu parity tests: call the C ABI on numpy data, return numpy results."""
import ctypes as C
import numpy as np
import torch

from dynamic_multiview_3d_amd import _lib

DEV = 'cuda'


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


def host(t):
    torch.cuda.synchronize()
    return t.detach().cpu().numpy()


def stream():
    return torch.cuda.current_stream().cuda_stream


class Ws:
    def __init__(self, nbytes):
        self.t = torch.empty(max(nbytes // 4, 4) + 4, dtype=torch.float32, device=DEV)
        self.ptr, self.bytes = self.t.data_ptr(), nbytes


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def conv_ws(g):
    return Ws(int(_lib.lib().conv_workspace_bytes(C.byref(g))))
