#!/usr/bin/env python
"""bench.py -- train images/sec of the appearance-flow model (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: forward, loss, hand-scheduled backward,
(gradient all-reduce over RCCL when N > 1), fused TF-Adam.  Workload = BASELINE.json configs[1]
`appflow_offset`: AppearanceFlowModel, 128x128x3, batch 64 per GPU (weak scaling), synthetic
car-render-like batches resident in HBM before the timed region, reference initialisers.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel of the step: algorithmic
FLOPs of its launches (2*N*Ho*Wo*kh*kw*Cin*Cout each) over their HIP-event durations
measured inside the timed region on the launch stream.  `cpu_baseline` times the numpy oracle
(CPU restatement of the TF-1.3 graph; TensorFlow itself is unavailable offline) on the host
cores, on a bounded sample (batch 8).
"""
import argparse
import gc
import collections
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no sparsity)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec peak


def is_split_bf16(label):
    """Kernels that evaluate every fp32 product as three bf16 MFMA products (bconv.hip)."""
    return label.startswith('bconv') or '_b3' in label
TRAIN_MFLOP_PER_IMAGE_CONV = 3035.6   # BASELINE.md section 2 (fwd + dgrad + wgrad, no dgrad for e0)
TRAIN_MFLOP_PER_IMAGE_ALL = 3440.0


def synth_batch(rng, b, h=128):
    """grey background, one filled ellipse, N(0,2) noise, uint8 -> /255."""
    yy, xx = np.mgrid[0:h, 0:h]

    def imgs():
        img = np.full((b, h, h, 3), 127.0, np.float32)
        for i in range(b):
            cy, cx = rng.uniform(40, 88, 2) * h / 128.0
            ay, ax = rng.uniform(15, 45, 2) * h / 128.0
            m = ((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1
            img[i][m] = rng.uniform(0, 255, 3)
        img += rng.normal(0, 2, img.shape).astype(np.float32)
        return (np.clip(np.rint(img), 0, 255) / 255.0).astype(np.float32)
    disp = np.stack([rng.uniform(-1, 1, b), rng.uniform(-6.28, 6.28, b)], 1).astype(np.float32)
    return dict(image0=imgs(), image1=imgs(), disp=disp)


def cpu_baseline(batch=8, steps=3):
    """numpy oracle (oracle/: CPU restatement of the reference graph) timed on the host cores."""
    from oracle import models as omodels
    from oracle.graph import Tape
    rng = np.random.default_rng(0)
    feeds = synth_batch(rng, batch)
    builder = omodels.appearance_flow_builder('base')
    t = Tape(None, rng=np.random.default_rng(1234))
    builder(t, {k: t.const(v) for k, v in feeds.items()})          # creates the variables
    variables, adam = t.vars, omodels.AdamState(1e-4)
    omodels.step(builder, variables, adam, feeds)                   # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        omodels.step(builder, variables, adam, feeds)
    dt = time.perf_counter() - t0
    # threads actually used: the BLAS pool numpy multiplies on (im2col GEMMs dominate), bounded by this process's CPU set
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    cores = avail
    try:
        from threadpoolctl import threadpool_info
        blas = [i.get('num_threads', 0) for i in threadpool_info() if i.get('user_api') == 'blas']
        if blas:
            cores = min(avail, max(blas))
    except Exception:
        pass
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "numpy/BLAS oracle, AppearanceFlowModel fwd+bwd+Adam, %d timed steps at batch %d after 1 warm-up "
                      "(TensorFlow 1.3 reference cannot run offline)" % (steps, batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='per-GPU batch (weak scaling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true', help='do not bracket launches with HIP events')
    ap.add_argument('--dump-kernels', action='store_true', help='print the per-kernel table to stderr')
    ap.add_argument('--dump-ops', action='store_true', help='print every launch of the step in order to stderr')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument('--single-device', action='store_true', help='rehearsal: all ranks share GPU 0')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from dynamic_multiview_3d_amd import _lib

    conf = {'batch_size': args.batch, 'learning_rate': 1e-4, 'experiment_name': 'appflow_offset'}
    model = AppearanceFlowModel(conf, load_tfrec=False, build_loss=True, device=dev, seed=1234)   # same init on all ranks
    g = model.graph
    model.feed(**synth_batch(np.random.default_rng(rank), args.batch))                           # resident in HBM
    if world > 1:
        model.enable_data_parallel(world)
    lib = g.lib

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    adam_ms = []        # per step: sum of the optimiser launches' HIP-event durations (filled by adam_collect)

    def one_step(adam_events=None):
        """The product's train step (Graph.train_step).  adam_events: time the optimiser launches of this step."""
        if world > 1:
            if adam_events is not None:         # kernel-table pass: optimiser timed as one launch behind the collectives
                g.run_forward()
                g.run_backward_overlapped()
                adam_events[0].record()
                g.apply_adam()
                adam_events[1].record()
                adam_ms.append([tuple(adam_events)])
            else:
                g.train_step()
        else:
            g.adam_timing = [] if adam_events is not None else None
            g.train_step()
            if adam_events is not None:
                adam_ms.append(g.adam_timing)
            g.adam_timing = None

    def serial_step(adam_events):
        """Kernel-table pass: the same launches on ONE stream (no side streams, Adam as a single launch behind the
        reverse pass), so that every event-bracketed duration is the kernel's own and the shares add up to the step."""
        g.run_forward()
        lib.plan_run(g.plan_bwd, torch.cuda.current_stream(dev).cuda_stream)
        if world > 1:
            g.allreduce_grads()
        adam_events[0].record()
        g.apply_adam()
        adam_events[1].record()
        adam_ms.append([tuple(adam_events)])

    def adam_collect():
        """mean ms per step over the steps timed so far (call after a synchronize); clears the list"""
        tot = [sum(a.elapsed_time(b) for a, b in step) for step in adam_ms]
        adam_ms.clear()
        return sum(tot) / max(len(tot), 1)

    def collect(kern):
        for plan in (g.plan_fwd, g.plan_bwd):
            lib.plan_profile_collect(plan)
            for name, fl, by, ms, runs in _lib.plan_ops(plan):
                if runs == 0:
                    continue
                if args.dump_ops and rank == 0:
                    m1 = ms / runs
                    print("%-34s %9.1f us  %8.3f GFLOP %8.1f MB  %7.1f TF/s %7.0f GB/s" % (name, m1 * 1e3, fl / 1e9, by / 1e6, fl / max(m1, 1e-9) / 1e9, by / max(m1, 1e-9) / 1e6), file=sys.stderr)
                k = kern.setdefault(name, dict(launches=0, flops=0.0, bytes=0.0, ms=0.0))
                k['launches'] += 1
                k['flops'] += fl
                k['bytes'] += by
                k['ms'] += ms / runs

    timing = not args.no_kernel_timing
    # ---- warm-up (untimed).  With kernel timing on, the warm-up steps after the first run the same launches on one
    # stream with every launch bracketed by HIP events: that gives the per-kernel table (own durations, shares that
    # add up) and names the dominant kernel.  The timed region then runs the product's multi-stream train step with
    # events on that kernel's launches only.
    table = collections.OrderedDict()
    dominant = None
    if args.warmup > 0 or timing:
        one_step()
    if timing:
        n_table = max(args.warmup - 1, 2)       # the table (and with it `roofline`) exists for any --warmup; extra steps are untimed
        for plan in (g.plan_fwd, g.plan_bwd):
            lib.plan_profile_reset(plan)
            lib.plan_profile_select(plan, None)
            lib.plan_profile(plan, 1)
        wev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_table)]
        for i in range(n_table):
            serial_step(wev[i])
        torch.cuda.synchronize()
        collect(table)
        table['adam'] = dict(launches=1, flops=0.0, bytes=28.0 * g.flat_size, ms=adam_collect())
        dominant = max(table.items(), key=lambda kv: kv[1]['ms'])[0]
    else:
        for _ in range(max(args.warmup - 1, 0)):
            one_step()
    # ---- timed region: EXACTLY K steps.  Only the dominant kernel's launches carry events (a handful per
    # step), so `value` and `roofline` come from the same region.
    for plan in (g.plan_fwd, g.plan_bwd):
        lib.plan_profile_reset(plan)
        if timing and dominant is not None and dominant != 'adam':
            lib.plan_profile_select(plan, dominant.encode())
            lib.plan_profile(plan, 1)
        else:
            lib.plan_profile(plan, 0)
    # N > 1: the product's step runs Adam bucket by bucket behind the all-reduces on the compute stream; bracketing it there
    # would need the un-bucketed form, so the timed region carries no events and Adam's duration is the warm-up table's
    time_adam = timing and dominant == 'adam' and world == 1
    adam_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if time_adam else []
    # Host hygiene, as `timeit` does: no cyclic-GC pass inside the timed region.  The launch thread runs ~45 ms ahead of
    # the GPU; a full collection over the interpreter's ~10^6 live objects (triggered by the per-step event objects)
    # stalls it for longer than that and the queues run dry -- measured: 2.6 -> 2.95 ms/step.
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(adam_ev[i] if (time_adam and i % 3 == 0) else None)      # every third step carries the events (each costs a queue barrier)
    sync()
    elapsed = time.perf_counter() - t0
    if gc_was_enabled:
        gc.enable()
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = float(g.loss_buf[0])

    # ---- dominant kernel inside the timed region (rank 0's GPU)
    kern = collections.OrderedDict()
    if timing and dominant is not None:
        if time_adam:
            kern['adam'] = dict(launches=1, flops=0.0, bytes=28.0 * g.flat_size, ms=adam_collect())
        elif dominant == 'adam':
            kern['adam'] = dict(table['adam'])
        else:
            collect(kern)
        for plan in (g.plan_fwd, g.plan_bwd):
            lib.plan_profile(plan, 0)
            lib.plan_profile_select(plan, None)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed

    out = {
        "metric": "train images/sec, appearance-flow encoder-decoder 128x128x3 (fwd+bwd+Adam)",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if (os.environ.get('MV3D_DISABLE') and int(os.environ['MV3D_DISABLE']) & 4096) else "bf16x3",
        "data": "synthetic",
        "config": {"workload": "appflow_offset: AppearanceFlowModel 128x128x3, batch %d per GPU, Adam lr 1e-4, "
                               "random-init weights (reference initialisers)" % args.batch,
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                   "precision": "conv / deconv / fc GEMMs: fp32 operands split into bf16 hi + lo, three v_mfma_f32_32x32x16_bf16 products per "
                                "fp32 product, fp32 accumulation (error ~1e-6 of the tensor scale; MV3D_DISABLE=4096 selects the exact "
                                "fp32-MFMA kernels); activations, loss, resampler, Adam and all stored tensors fp32",
                   "launches_per_step": g.n_launch_fwd + g.n_launch_bwd + 1},
        "loss": round(loss, 6),
    }
    if timing and kern and dominant in kern:
        dom_name, dom = dominant, kern[dominant]
        if dom['flops'] > 0:
            ach = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            if is_split_bf16(dom_name):
                # algorithmic FLOPs against the dense bf16 peak; the kernel executes 3 MFMA FLOPs per algorithmic one
                roof = {"bound": "mfma", "kernel": dom_name, "achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4), "mfma_dtype": "bf16 (split: 3 products per f32 product)",
                        "executed_tflops": round(3 * ach, 2), "frac_executed": round(3 * ach / PEAK_BF16_MFMA_TFLOPS, 4),
                        "frac_vs_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4)}
            else:
                roof = {"bound": "mfma", "kernel": dom_name, "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "mfma_dtype": "f32"}
        else:
            ach = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4)}
        gpu_ms = sum(k['ms'] for k in table.values())
        roof.update({"launches_per_step": dom['launches'], "avg_launch_ms": round(dom['ms'] / dom['launches'], 5),
                     "share_of_gpu_time": round(table[dom_name]['ms'] / gpu_ms, 4), "traffic": None,
                     "measured": ("HIP events around the optimiser launches (their own stream) inside the timed region, on every third step; "
                                  "the reverse pass runs filter-gradient kernels and Adam on side streams, so a launch shares the GPU with concurrent kernels")
                                 if time_adam else
                                 ("HIP events around this kernel's launches inside the timed region (a launch may share the GPU with "
                                  "side-stream kernels)" if dom_name != 'adam' else
                                  "HIP events around the optimiser launch of the warm-up steps (N > 1: the timed region runs the product's "
                                  "bucketed Adam behind the all-reduces and carries no events)")})
        # the same kernel with the GPU to itself (warm-up table pass, streams serialized): separates kernel quality from the
        # cost of sharing HBM / CUs with the kernels it overlaps in the product schedule
        tk = table.get(dom_name)
        if tk and tk['ms'] > 0:
            if tk['flops'] > 0:
                a1, pk = tk['flops'] / (tk['ms'] * 1e-3) / 1e12, roof['peak']
            else:
                a1, pk = tk['bytes'] / (tk['ms'] * 1e-3) / 1e9, PEAK_HBM_GBS
            roof["alone"] = {"avg_launch_ms": round(tk['ms'] / tk['launches'], 5), "achieved": round(a1, 1), "frac": round(a1 / pk, 4),
                             "measured": "HIP events, warm-up steps, one stream"}
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                ent = json.load(open(tfile)).get(dom_name)
                if ent:      # HBM bytes per launch from committed rocprofv3 PMC passes (profiles/), FETCH_SIZE x2-corrected
                    # per launch as counted in `launches_per_step` above (the optimiser is one logical launch here, 4-5 slices in the product)
                    roof["traffic"] = round(ent["hbm_bytes_per_launch"] * (ent.get("launches_per_step") or dom['launches']) / dom['launches'])
                    roof["algorithmic_bytes_per_launch"] = round(dom['bytes'] / dom['launches'])
            except Exception:
                pass
        out["roofline"] = roof
        mfma = {n: k for n, k in table.items() if k['flops'] > 0}
        conv_ms = sum(k['ms'] for k in mfma.values())
        conv_fl = sum(k['flops'] for k in mfma.values())
        hbm = {n: k for n, k in table.items() if k['flops'] <= 0 and k['bytes'] > 0}
        hbm_ms = sum(k['ms'] for k in hbm.values())
        out["stack"] = {"source": "HIP events around every launch during the warm-up steps, streams serialized (each duration is the kernel's own)",
                        "gpu_ms_per_step_sum_of_kernels": round(gpu_ms, 4),
                        "mfma_kernels_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 2),
                        "mfma_kernels_frac_of_f32_peak": round(conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                        "mfma_kernels_frac_of_bf16_peak": round(conv_fl / (conv_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                        "mfma_kernels_share_of_gpu_time": round(conv_ms / gpu_ms, 4),
                        "hbm_kernels_gbs": round(sum(k['bytes'] for k in hbm.values()) / (hbm_ms * 1e-3) / 1e9, 1) if hbm_ms > 0 else None,
                        "hbm_kernels_share_of_gpu_time": round(hbm_ms / gpu_ms, 4)}
        if args.dump_kernels and rank == 0:
            for n, k in sorted(table.items(), key=lambda kv: -kv[1]['ms']):
                rate = ("%7.1f TF/s" % (k['flops'] / k['ms'] / 1e9)) if k['flops'] > 0 else ("%7.0f GB/s" % (k['bytes'] / k['ms'] / 1e6))
                print("%-34s launches=%3d  ms/step=%8.4f  %s" % (n, k['launches'], k['ms'], rate), file=sys.stderr)
    # whole step against the roofline the reference's arithmetic implies: algorithmic fp32 FLOPs (3 440 MFLOP per trained
    # image) over wall time, vs the fp32 MFMA peak -- independent of which matrix-core type carries them
    out["step_f32_equivalent"] = {"tflops": round(value / world * TRAIN_MFLOP_PER_IMAGE_ALL * 1e6 / 1e12, 2),
                                  "frac_of_f32_mfma_peak": round(value / world * TRAIN_MFLOP_PER_IMAGE_ALL * 1e6 / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                  "note": "per GPU; includes fc, Adam and every elementwise kernel in the time"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
