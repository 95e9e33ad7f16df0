#!/usr/bin/env python
"""In-kernel stamps of the thin filter-gradient kernel (thin.hip) on the e0 layer: MV3D_DBG=32 python tools/thinwg_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynamic_multiview_3d_amd import _lib
lib = _lib.lib()
B = 64
g = _lib.conv_geom(B, 128, 128, 3, 32, 5, 5, 2, 2)
img = torch.rand(B, 128, 128, 3, device='cuda'); dy = torch.randn(B, 64, 64, 32, device='cuda')
dw = torch.empty(5, 5, 3, 32, device='cuda'); db = torch.empty(32, device='cuda')
wsb = int(lib.conv_workspace_bytes(C.byref(g))); ws = torch.empty(wsb // 4 + 4, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    lib.conv2d_wgrad(C.byref(g), img.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, st)
torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 16, np.uint64)
lib.debug_band_stamps(buf.ctypes.data, buf.nbytes)
s = buf.reshape(512, 4, 16).astype(np.int64)
names = ['start', 'dY requested', 'planes written', 'barrier', 'products', 'barrier', 'stored']
for k in range(1, 7):
    d = s[:, :, k] - s[:, :, k - 1]
    print("%-16s median %6d  p90 %6d  max %6d cycles" % (names[k], np.median(d), np.percentile(d, 90), d.max()))
print("wave life: median %d  max %d" % (np.median(s[:, :, 6] - s[:, :, 0]), (s[:, :, 6] - s[:, :, 0]).max()))
