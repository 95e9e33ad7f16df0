#!/usr/bin/env python
"""Does replaying the forward / reverse launch sequences from a captured hipGraph shorten the gaps between dependent kernels?
(informational: prints stream-launch vs graph-replay times of the forward plan and of the single-stream reverse plan)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['MV3D_PIPELINE_FCADAM'] = '0'
import numpy as np
import torch


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from bench import synth_batch
    model = AppearanceFlowModel({'batch_size': 64, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda:0', seed=1234)
    g = model.graph
    model.feed(**synth_batch(np.random.default_rng(0), 64))
    for _ in range(3):
        g.train_step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for name, fn in (('forward plan', lambda: g.lib.plan_run(g.plan_fwd, torch.cuda.current_stream().cuda_stream)),
                     ('reverse plan, one stream', lambda: g.lib.plan_run_range_multi(g.plan_bwd, 0, g.n_launch_bwd, torch.cuda.current_stream().cuda_stream, None, 0, 0))):
        with torch.cuda.stream(side):
            t_stream = timeit(fn)
            gr = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(gr, stream=side):
                fn()
            t_graph = timeit(gr.replay)
        print("%-28s stream launches %.3f ms   hipGraph replay %.3f ms" % (name, t_stream, t_graph))


if __name__ == '__main__':
    main()
