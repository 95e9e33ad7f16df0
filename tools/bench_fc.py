#!/usr/bin/env python
"""Micro-benchmark of the fc entry points (GPU box only): python tools/bench_fc.py [B in out]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_multiview_3d_amd import _lib
lib = _lib.lib()
B, n_in, n_out = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 4096, 4096)
x = torch.randn(B, n_in, device='cuda'); dy = torch.randn(B, n_out, device='cuda')
M = torch.randn(n_in, n_out, device='cuda') * 0.02; dM = torch.empty_like(M); db = torch.empty(n_out, device='cuda')
y = torch.empty(B, n_out, device='cuda'); dx = torch.empty(B, n_in, device='cuda'); b = torch.zeros(n_out, device='cuda')
wsb = int(lib.fc_workspace_bytes(B, n_in, n_out)); ws = torch.empty(max(wsb // 4, 4), device='cuda')
st = torch.cuda.current_stream().cuda_stream
epi = _lib.epilogue(b.data_ptr(), 1, 0.2)
epi0 = _lib.epilogue()
ops = {'fwd': lambda: lib.fc_fwd(B, n_in, n_out, x.data_ptr(), n_in, M.data_ptr(), y.data_ptr(), n_out, C.byref(epi), ws.data_ptr(), wsb, st),
       'dgrad': lambda: lib.fc_dgrad(B, n_in, n_out, dy.data_ptr(), n_out, M.data_ptr(), dx.data_ptr(), n_in, C.byref(epi0), ws.data_ptr(), wsb, st),
       'wgrad': lambda: lib.fc_wgrad(B, n_in, n_out, x.data_ptr(), n_in, dy.data_ptr(), n_out, dM.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, st)}
for name, fn in ops.items():
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-6s %7.1f us  %6.2f TB/s (matrix bytes / time)' % (name, us, n_in * n_out * 4 / us / 1e6), flush=True)
