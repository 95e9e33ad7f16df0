#!/usr/bin/env python
"""1500 train steps with rotating feeds, once with the fc optimiser pipelined under the next step and once joined: the two runs must end
with bit-identical parameters (a missed cross-step dependency would show) and a falling loss.  ~10 s on an MI355X."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
from bench import synth_batch
res = {}
for pipe in ('1', '0'):
    os.environ['MV3D_PIPELINE_FCADAM'] = pipe
    m = AppearanceFlowModel({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda:0', seed=1234)
    g = m.graph
    rng = np.random.default_rng(0)
    batches = [synth_batch(rng, 64) for _ in range(4)]
    losses = []
    t0 = time.time()
    for i in range(1500):
        m.feed(**batches[i % 4])
        l = g.train_step()
        if i % 250 == 0 or i == 1499:
            losses.append(float(l))
    torch.cuda.synchronize()
    g.settle()
    res[pipe] = (losses, g.params.cpu().numpy().copy())
    print("pipeline=%s  %.1f s  losses %s" % (pipe, time.time() - t0, ["%.5f" % x for x in losses]), flush=True)
a, b = res['1'], res['0']
print("params bit-identical after 1500 steps:", bool(np.array_equal(a[1], b[1])), " finite:", bool(np.isfinite(a[1]).all()))
