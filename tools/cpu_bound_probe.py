#!/usr/bin/env python
"""Is the train step launch-bound on this host?  Times the host-side enqueue of N steps (no sync) against the GPU
completion time of the same N steps (GPU box only): python tools/cpu_bound_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = AppearanceFlowModel({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda:0', seed=1234)
rng = np.random.default_rng(0)
m.feed(**{k: rng.uniform(0, 1, t.shape).astype(np.float32) for k, t in m.graph.inputs.items()})
for _ in range(5):
    m.graph.train_step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N):
        m.graph.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.3f ms/step   until idle %.3f ms/step   (host ahead by %.1f ms at the end)' %
          ((t1 - t0) * 1e3 / N, (t2 - t0) * 1e3 / N, (t2 - t1) * 1e3), flush=True)
print('loadavg', open('/proc/loadavg').read().strip(), 'cpus', os.cpu_count())
