#!/usr/bin/env python
"""Runs ONE conv / deconv layer call through the C ABI in a loop (for rocprofv3 --pmc passes and quick timings).
usage: python tools/layer_probe.py conv|deconv fwd|dgrad|wgrad n h w c k ksz stride [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dynamic_multiview_3d_amd import _lib
kind, op = sys.argv[1], sys.argv[2]
n, h, w, c, k, ksz, s = [int(a) for a in sys.argv[3:10]]
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 20
L = _lib.lib()
g = _lib.conv_geom(n, h, w, c, k, ksz, ksz, s, s)
ho, wo = -(-h // s), -(-w // s)
wsb = int(L.conv_workspace_bytes(C.byref(g)))
ws = torch.empty(wsb // 4 + 64, device='cuda')
img = torch.randn(n, h, w, c, device='cuda')
feat = torch.randn(n, ho, wo, k, device='cuda')
wt = torch.randn(ksz, ksz, c, k, device='cuda') * 0.05 if kind == 'conv' else torch.randn(ksz, ksz, c, k, device='cuda') * 0.05
b = torch.randn(k if kind == 'conv' else c, device='cuda')
dw, db = torch.empty_like(wt), torch.empty(k, device='cuda')
st = torch.cuda.current_stream().cuda_stream
epi = _lib.epilogue(b.data_ptr(), _lib.ACT_LRELU, 0.2)
epi0 = _lib.epilogue(None, 0, 0.2)


def call():
    if kind == 'conv':
        if op == 'fwd': L.conv2d_fwd(C.byref(g), img.data_ptr(), wt.data_ptr(), feat.data_ptr(), C.byref(epi), ws.data_ptr(), wsb, st)
        elif op == 'dgrad': L.conv2d_dgrad(C.byref(g), feat.data_ptr(), wt.data_ptr(), img.data_ptr(), C.byref(epi0), ws.data_ptr(), wsb, st)
        else: L.conv2d_wgrad(C.byref(g), img.data_ptr(), feat.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, st)
    else:
        if op == 'fwd': L.deconv2d_fwd(C.byref(g), feat.data_ptr(), wt.data_ptr(), img.data_ptr(), C.byref(epi0), ws.data_ptr(), wsb, st)
        elif op == 'dgrad': L.deconv2d_dgrad(C.byref(g), img.data_ptr(), wt.data_ptr(), feat.data_ptr(), C.byref(epi0), ws.data_ptr(), wsb, st)
        else: L.deconv2d_wgrad(C.byref(g), feat.data_ptr(), img.data_ptr(), dw.data_ptr(), ws.data_ptr(), wsb, st)


for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
print("%s %s %s: %.1f us per call (back to back, filter conversion inside where the layer has one)" % (kind, op, sys.argv[3:10], e0.elapsed_time(e1) / reps * 1e3))
