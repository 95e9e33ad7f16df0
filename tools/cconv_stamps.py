#!/usr/bin/env python
"""Times one conv layer through the C ABI and prints the in-kernel stamps of the pipelined kernel (cconv.hip, MV3D_DBG=32).

    MV3D_DBG=32 python tools/cconv_stamps.py [n h w c k ksz] [--dgrad | --wgrad]
(--wgrad: the filter gradient of the same layer, cwgrad_kernel: for the SQ counter passes; it writes no stamps)
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dynamic_multiview_3d_amd import _lib

args = [a for a in sys.argv[1:] if not a.startswith('--')]
n, h, w, c, k, ksz = [int(a) for a in args] if args else (64, 64, 64, 32, 32, 5)
dgrad = '--dgrad' in sys.argv
wgrad = '--wgrad' in sys.argv
L = _lib.lib()
g = _lib.conv_geom(n, h, w, c, k, ksz, ksz, 1, 1)
wsb = int(L.conv_workspace_bytes(C.byref(g)))
ws = torch.empty(wsb // 4 + 64, device='cuda')
x = torch.randn(n, h, w, c, device='cuda')
y = torch.randn(n, h, w, k, device='cuda')
wt = torch.randn(ksz, ksz, c, k, device='cuda') * 0.05
b = torch.randn(k, device='cuda')
ref = torch.randn(n, h, w, c, device='cuda')
st = torch.cuda.current_stream().cuda_stream


dw = torch.empty(ksz, ksz, c, k, device='cuda')
db = torch.empty(k, device='cuda')


def call():
    if wgrad:
        L.conv2d_wgrad(C.byref(g), x.data_ptr(), y.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, st)
    elif dgrad:
        epi = _lib.epilogue(None, 0, 0.2, _lib.ACT_LRELU, 0.2, ref.data_ptr(), c)
        L.conv2d_dgrad(C.byref(g), y.data_ptr(), wt.data_ptr(), x.data_ptr(), C.byref(epi), ws.data_ptr(), wsb, st)
    else:
        epi = _lib.epilogue(b.data_ptr(), _lib.ACT_LRELU, 0.2)
        L.conv2d_fwd(C.byref(g), x.data_ptr(), wt.data_ptr(), y.data_ptr(), C.byref(epi), ws.data_ptr(), wsb, st)


for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * n * h * w * ksz * ksz * c * k
print("%s %dx%dx%dx%d->%d %dx%d: %.1f us per call (incl. the per-call filter split), %.1f TF/s" % ('wgrad' if wgrad else 'dgrad' if dgrad else 'fwd', n, h, w, c, k, ksz, ksz, ms * 1e3, fl / ms / 1e9))
if int(os.environ.get('MV3D_DBG', '0')) & 32 and not wgrad:
    NS = 64
    buf = np.zeros(256 * 8 * NS, np.uint64)
    L.dll.mv3d_debug_cconv_stamps.argtypes = [C.c_void_p, C.c_size_t]
    rc = L.dll.mv3d_debug_cconv_stamps(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    s = buf.reshape(256, 8, NS).astype(np.int64)
    t0 = s[:, :, 0].min()
    for wg in (0, 1, 100, 255):
        print("workgroup", wg)
        for wv in (0, 3, 4, 7):
            row = s[wg, wv]
            row = row[row > 0] - t0
            print("  wave %d (%s):" % (wv, 'M' if wv < 4 else 'D'), ' '.join("%6d" % v for v in np.diff(np.concatenate([[0], row]))[:40]))
    m_end = s[:, 0, :].max(axis=1) - t0
    print("last stamp of wave 0 per workgroup: min %d median %d max %d cycles (100 MHz realtime? no: shader clock)" % (m_end.min(), np.median(m_end), m_end.max()))
if int(os.environ.get('MV3D_DBG', '0')) & 32 and wgrad:
    NS = 64
    buf = np.zeros(128 * 8 * NS, np.uint64)
    rc = L.dll.mv3d_debug_cwgrad_stamps(buf.ctypes.data, buf.nbytes)
    assert rc == 0, rc
    s = buf.reshape(128, 8, NS).astype(np.int64)
    t0 = s[:, :, 0][s[:, :, 0] > 0].min()
    for wg in (0, 1, 64, 127):
        print("slab", wg)
        for wv in (0, 3, 4, 7):
            row = s[wg, wv]
            row = row[row > 0] - t0
            print("  wave %d (%s):" % (wv, 'M' if wv < 4 else 'D'), ' '.join("%6d" % v for v in np.diff(np.concatenate([[0], row]))[:44]))
