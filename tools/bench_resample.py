#!/usr/bin/env python
"""Micro-benchmark of the resampler entry points at the bench shape (GPU box only): python tools/bench_resample.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dynamic_multiview_3d_amd import _lib
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = 128
src = torch.rand(B, H, H, 3, device='cuda'); tgt = torch.rand(B, H, H, 3, device='cuda')
flow = torch.randn(B, H, H, 2, device='cuda') * 3
warp = torch.empty_like(flow); dflow = torch.empty_like(flow); gen = torch.empty_like(src); dgen = torch.randn_like(src)
loss = torch.zeros(4, device='cuda')
st = torch.cuda.current_stream().cuda_stream
ops = {
    'fused': (lambda: lib.warp_resample_loss(B, H, H, H, H, 3, src.data_ptr(), flow.data_ptr(), 2, tgt.data_ptr(), 3, 2, 1.0, warp.data_ptr(),
                                             gen.data_ptr(), dflow.data_ptr(), 2, loss.data_ptr(), st), 52.0),
    'fwd': (lambda: lib.warp_resample_fwd(B, H, H, H, H, 3, src.data_ptr(), flow.data_ptr(), 2, warp.data_ptr(), gen.data_ptr(), st), 40.0),
    'bwd': (lambda: lib.warp_resample_bwd(B, H, H, H, H, 3, src.data_ptr(), flow.data_ptr(), 2, dgen.data_ptr(), dflow.data_ptr(), 2, st), 40.0),
    'loss': (lambda: lib.pixel_loss(B * H * H, 3, gen.data_ptr(), tgt.data_ptr(), None, 2, 1.0, loss.data_ptr(), dgen.data_ptr(), st), 36.0),
}
for name, (fn, bpp) in ops.items():
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('%-6s %7.1f us  %6.2f TB/s (%g B/pixel)' % (name, us, B * H * H * bpp / us / 1e6, bpp), flush=True)
