#!/usr/bin/env python
"""Step time of the data-parallel schedule (bucket segments, communication stream, sharded optimiser) at world size 1 through RCCL:
what the per-GPU compute side of a multi-GPU step costs next to the single-GPU step (no exchange time: world 1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dynamic_multiview_3d_amd import parallel
from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
from bench import synth_batch
m = AppearanceFlowModel({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda:0', seed=1234)
g = m.graph
m.feed(**synth_batch(np.random.default_rng(0), 64))
if '--single' not in sys.argv:
    m.enable_data_parallel(1, comm=parallel.RcclComm(0, 1), mode='sharded')
    g.world_size = 2 if False else 1
def step():
    if '--single' in sys.argv:
        g.train_step()
    else:
        g.run_forward(); g.run_backward_overlapped(with_adam=True)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    step()
torch.cuda.synchronize()
print("%s: %.4f ms per step" % ('single-GPU step' if '--single' in sys.argv else 'data-parallel schedule, world 1 (MV3D_DP_JOIN=%s)' % os.environ.get('MV3D_DP_JOIN', '0'), (time.perf_counter() - t0) / 40 * 1e3))
