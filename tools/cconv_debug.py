#!/usr/bin/env python
"""Debug helper: one conv forward through the pipelined kernel vs the oracle, with an error map."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dynamic_multiview_3d_amd import _lib
from oracle import ops
n, h, w, c, k, ksz = 4, 32, 32, 32, 32, 5
L = _lib.lib()
g = _lib.conv_geom(n, h, w, c, k, ksz, ksz, 1, 1)
wsb = int(L.conv_workspace_bytes(C.byref(g)))
ws = torch.empty(wsb // 4 + 64, device='cuda')
rng = np.random.default_rng(0)
x = rng.standard_normal((n, h, w, c)).astype(np.float32)
wt = (rng.standard_normal((ksz, ksz, c, k)) / np.sqrt(ksz * ksz * c)).astype(np.float32)
b = rng.standard_normal(k).astype(np.float32)
dx, dw, db = torch.tensor(x).cuda(), torch.tensor(wt).cuda(), torch.tensor(b).cuda()
y = torch.full((n, h, w, k), 7.0, device='cuda')
epi = _lib.epilogue(db.data_ptr(), _lib.ACT_LRELU, 0.2)
plan = L.plan_create(); L.plan_begin(plan)
L.conv2d_fwd(C.byref(g), dx.data_ptr(), dw.data_ptr(), y.data_ptr(), C.byref(epi), ws.data_ptr(), wsb, None)
L.plan_end(); print([o[0] for o in _lib.plan_ops(plan)])
L.conv2d_fwd(C.byref(g), dx.data_ptr(), dw.data_ptr(), y.data_ptr(), C.byref(epi), ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
got = y.cpu().numpy()
ref = ops.absact_fwd(ops.conv2d_fwd(x, wt, b, 1, 1), 'lrelu')
err = np.abs(got - ref)
print("max err", err.max(), "max ref", np.abs(ref).max(), "frac == 7.0:", (got == 7.0).mean(), "frac == 0:", (got == 0).mean(), "nan:", np.isnan(got).mean())
print("err by image:", err.reshape(n, -1).max(1))
e2 = err.max(axis=(0, 3))
print("err map (rows x cols, max over images/channels), >1e-3 marked:")
for r in range(h):
    print(''.join('#' if v > 1e-3 else '.' for v in e2[r]))
print("err by channel:", np.round(err.max(axis=(0, 1, 2)), 3))
print("sample got/ref at [0,5,5,:4]", got[0, 5, 5, :4], ref[0, 5, 5, :4], " at [0,0,0,:4]", got[0, 0, 0, :4], ref[0, 0, 0, :4])
