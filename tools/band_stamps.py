#!/usr/bin/env python
"""In-kernel stamps of the row-band kernel (thin.hip) on the e0 layer: MV3D_DBG=32 python tools/band_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dynamic_multiview_3d_amd import _lib
lib = _lib.lib()
B = 64
g = _lib.conv_geom(B, 128, 128, 3, 32, 5, 5, 2, 2)
img = torch.rand(B, 128, 128, 3, device='cuda'); w = torch.randn(5, 5, 3, 32, device='cuda') * 0.1
feat = torch.empty(B, 64, 64, 32, device='cuda'); bias = torch.zeros(32, device='cuda')
epi = _lib.epilogue(bias.data_ptr(), _lib.ACT_LRELU, 0.2)
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    lib.conv2d_fwd(C.byref(g), img.data_ptr(), w.data_ptr(), feat.data_ptr(), C.byref(epi), None, 0, st)
torch.cuda.synchronize()
buf = np.zeros(512 * 4 * 16, np.uint64)
lib.debug_band_stamps(buf.ctypes.data, buf.nbytes)
s = buf.reshape(512, 4, 16).astype(np.int64)
t0 = s[:, :, 0].min()
names = ['start', 'rows split', 'pads zeroed', 'filter frags', 'barrier', 'tile0', 'tile1', 'tile2', 'tile3', 'stores acked']
print("kernel span (first start -> last stamp): %d cycles" % (s[:, :, :10].max() - t0))
print("workgroup start times (cycles after the first): median %d  p90 %d  max %d" % (np.median(s[:, 0, 0] - t0), np.percentile(s[:, 0, 0] - t0, 90), (s[:, 0, 0] - t0).max()))
for k in range(1, 10):
    d = s[:, :, k] - s[:, :, k - 1]
    print("%-14s median %6d  p90 %6d  max %6d cycles" % (names[k], np.median(d), np.percentile(d, 90), d.max()))
print("wave life: median %d  max %d" % (np.median(s[:, :, 9] - s[:, :, 0]), (s[:, :, 9] - s[:, :, 0]).max()))
