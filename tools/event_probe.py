#!/usr/bin/env python
"""Does a HIP timing event leave its queue slower?  Times the multi-stream train step before and after timing events
were recorded on (a) another stream, (b) the main stream, (c) around every launch of four serial steps (GPU box only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
N = 30
m = AppearanceFlowModel({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda:0', seed=1234)
g = m.graph
lib = g.lib
rng = np.random.default_rng(0)
m.feed(**{k: rng.uniform(0, 1, t.shape).astype(np.float32) for k, t in g.inputs.items()})
for _ in range(5):
    g.train_step()
torch.cuda.synchronize()
def run():
    t0 = time.perf_counter()
    for _ in range(N):
        g.train_step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / N
print('baseline                         %.3f %.3f' % (run(), run()), flush=True)
other = torch.cuda.Stream()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(200)]
for e in ev:
    e.record(other)
torch.cuda.synchronize()
print('200 timing events, other stream  %.3f %.3f' % (run(), run()), flush=True)
ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for e in ev2:
    e.record()
torch.cuda.synchronize()
print('2 timing events, main stream     %.3f %.3f' % (run(), run()), flush=True)
ev3 = [torch.cuda.Event(enable_timing=True) for _ in range(1000)]
for e in ev3:
    e.record()
torch.cuda.synchronize()
print('1000 timing events, main stream  %.3f %.3f' % (run(), run()), flush=True)
del ev3, ev2, ev
torch.cuda.synchronize()
print('after deleting them              %.3f %.3f' % (run(), run()), flush=True)
for plan in (g.plan_fwd, g.plan_bwd):
    lib.plan_profile_reset(plan); lib.plan_profile_select(plan, None); lib.plan_profile(plan, 1)
for _ in range(4):
    g.run_forward(); lib.plan_run(g.plan_bwd, torch.cuda.current_stream().cuda_stream); g.apply_adam()
torch.cuda.synchronize()
for plan in (g.plan_fwd, g.plan_bwd):
    lib.plan_profile_collect(plan); lib.plan_profile(plan, 0)
print('after a profiled table pass      %.3f %.3f' % (run(), run()), flush=True)
