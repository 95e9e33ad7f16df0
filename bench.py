#!/usr/bin/env python
"""bench.py -- train images/sec of the appearance-flow model (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch: forward, loss, hand-scheduled backward,
(gradient all-reduce over RCCL when N > 1), fused TF-Adam.  Workload = BASELINE.json configs[1]
`appflow_offset`: AppearanceFlowModel, 128x128x3, batch 64 per GPU (weak scaling), synthetic
car-render-like batches resident in HBM before the timed region, reference initialisers.

Prints ONE JSON line (rank 0).  `roofline` describes the conv / deconv stack -- the kernel family north_star's
MFMA target is set on -- as ONE aggregate: algorithmic FLOPs of all its launches (2*N*Ho*Wo*kh*kw*Cin*Cout each,
DESIGN.md; slab reductions, split-K tails and the filter conversion count with 0 FLOPs but their time) over their
HIP-event durations measured inside the timed region on the streams they are launched on; `roofline.top_kernel`
is the same for the family's largest single kernel.  `cpu_baseline` times the CPU restatement of the TF-1.3 graph
(oracle/torch_tape.py on torch's CPU convolution library; TensorFlow itself is unavailable offline) on the host cores:
batch 8 and batch 64, 3 warm-up + 10 timed steps each, median.

`--gpus N` without a torch.distributed launcher in the environment starts the N rank processes itself (fresh children,
created before this process touches the GPU) and relays rank 0's JSON line.
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
    return label.startswith(('bconv', 'cconv', 'cwgrad')) or '_b3' in label


# conv / deconv family: forward, data-gradient and filter-gradient kernels of conv2d_msra / deconv2d_msra
# (tf_utils.py:70-98) plus the launches that only exist to serve them (mv3d_plan_profile_select pattern syntax)
CONV_FAMILY = ('cconv*|cwgrad*|bconv*|sconv*|s2conv*|wgrad_b3*|wgrad_tile*|hconv*|igemm*|smallc_*|thin_*|filtgrad*|reduce_slabs|grad_finalize*|transpose_filter')


def in_conv_family(label):
    for pat in CONV_FAMILY.split('|'):
        if (pat.endswith('*') and label.startswith(pat[:-1])) or label == pat:
            return True
    return False

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


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_baseline(batches=(8, 64), warmup=3, steps=10, budget_s=25.0):
    """the CPU restatement of the TF-1.3 graph (oracle/torch_tape.py: torch CPU ops + autograd, fp32, TF-Adam)
    on the host cores; configs appflow_firsttry (batch 8) and appflow_offset (batch 64), 3 warm-up + 10 timed steps, median."""
    from oracle import models as omodels
    from oracle.graph import Tape
    from oracle.torch_tape import TorchTrainer
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    threads = max(1, min(avail, torch.get_num_threads()))
    per_batch = {}
    for b in batches:
        feeds = synth_batch(np.random.default_rng(0), b)
        builder = omodels.appearance_flow_builder('base')
        t = Tape(None, rng=np.random.default_rng(1234))
        builder(t, {k: t.const(v) for k, v in feeds.items()})          # creates the variables (reference initialisers)
        trainer = TorchTrainer(builder, t.vars, lr=1e-4)
        for _ in range(warmup):
            trainer.step(feeds)
        times = []
        t_begin = time.perf_counter()
        for _ in range(steps):
            t0 = time.perf_counter()
            trainer.step(feeds)
            times.append(time.perf_counter() - t0)
            if len(times) >= 3 and time.perf_counter() - t_begin > budget_s:      # bounded sample: stop early on a slow host
                break
        n_timed = len(times)
        times.sort()
        per_batch[b] = {"images_per_sec": round(b / times[len(times) // 2], 2), "median_step_s": round(times[len(times) // 2], 4),
                        "min_step_s": round(times[0], 4), "timed_steps": n_timed}
    main_b = batches[-1]
    return {"value": per_batch[main_b]["images_per_sec"], "unit": "images/sec", "cores": threads, "kind": "port",
            "cpu_model": _cpu_model(), "cpus_available": avail,
            "by_batch": {str(b): v for b, v in per_batch.items()},
            "sample": "CPU restatement of the TF-1.3 graph (oracle/torch_tape.py: torch CPU convolution / matmul kernels + autograd, "
                      "fp32, TF-Adam; TensorFlow 1.3 itself cannot run offline), AppearanceFlowModel fwd+bwd+Adam, %d warm-up + up to %d timed "
                      "steps per batch size (at least 3, stops after %.0f s), median; value = batch %d, batch %d beside it"
                      % (warmup, steps, budget_s, main_b, batches[0])}


def self_launch(args):
    """`bench.py --gpus N` from a plain shell: start N ranks (one process per GPU) under torch.distributed.run as fresh
    children -- this process has not touched the GPU -- and exit with their status; rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='per-GPU batch (weak scaling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true', help='do not bracket launches with HIP events')
    ap.add_argument('--event-every', type=int, default=8, help='steps of the timed region that carry HIP events on the conv/deconv family: every n-th')
    ap.add_argument('--dump-kernels', action='store_true', help='print the per-kernel table to stderr')
    ap.add_argument('--dump-ops', action='store_true', help='print every launch of the step in order to stderr')
    ap.add_argument('--backend', default='rccl', help="gradient exchange: rccl = RCCL through the C ABI (mv3d_comm_*), nccl = torch.distributed's RCCL binding, "
                                                      "gloo = CPU collectives (rehearsals on one GPU)")
    ap.add_argument('--dp-mode', default='sharded', choices=['sharded', 'allreduce'],
                    help="sharded: reduce-scatter -> Adam on 1/N of every bucket -> all-gather; allreduce: SUM + redundant Adam")
    ap.add_argument('--single-device', action='store_true', help='rehearsal: all ranks share GPU 0')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args)
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)      # control plane: rendezvous, RCCL id, barriers, the max over ranks

    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    from dynamic_multiview_3d_amd import _lib

    conf = {'batch_size': args.batch, 'learning_rate': 1e-4, 'experiment_name': 'appflow_offset'}
    model = AppearanceFlowModel(conf, load_tfrec=False, build_loss=True, device=dev, seed=1234)   # same init on all ranks
    g = model.graph
    model.feed(**synth_batch(np.random.default_rng(rank), args.batch))                           # resident in HBM
    if world > 1:
        from dynamic_multiview_3d_amd import parallel
        comm = parallel.make_comm(rank, world, args.backend)
        model.enable_data_parallel(world, comm=comm, mode=args.dp_mode)
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
        fused = world == 1 and g.plan_bwd_fused is not None
        lib.plan_run(g.plan_bwd_fused if fused else g.plan_bwd, torch.cuda.current_stream(dev).cuda_stream)
        if world > 1:
            g.allreduce_grads()
        adam_events[0].record()
        if fused and g._finalized_in_plan:
            # the plan's last launch (grad_finalize_adam) summed the filter-gradient slabs and updated everything but the fc
            # matrices (fused into their filter-gradient kernels) and their biases: those four bias vectors in one launch here
            lo, hi = g._bias_span
            g._adam_range(lo, hi, g._stream_ptr(), g._bias_skip)
            g._adam_advance()
        elif fused:     # the fc matrices were updated inside their filter-gradient kernels: the rest in one launch
            g._adam_range(0, g.flat_size, g._stream_ptr(), (len(g._skip_lo), g._skip_lo, g._skip_hi))
            g._adam_advance()
        else:
            g.apply_adam()
        adam_events[1].record()
        adam_ms.append([tuple(adam_events)])

    def adam_collect():
        """mean ms per step over the steps timed so far (call after a synchronize); clears the list"""
        tot = [sum(a.elapsed_time(b) for a, b in step) for step in adam_ms]
        adam_ms.clear()
        return sum(tot) / max(len(tot), 1)

    bwd_plan = g.plan_bwd_fused if (world == 1 and g.plan_bwd_fused is not None) else g.plan_bwd

    def collect(kern):
        for plan in (g.plan_fwd, bwd_plan):
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
    # ---- warm-up (untimed).  With kernel timing on, the warm-up steps after the first run the same launches on ONE
    # stream with every launch bracketed by HIP events: that gives the per-kernel table of OWN durations (shares that add
    # up; `stack`, `roofline.alone`).  The timed region then runs the product's multi-stream train step; every
    # `--event-every`-th step of it carries events on the conv / deconv family's launches (and on the optimiser's), on the
    # streams they are issued on, which is what `roofline` reports.
    table = collections.OrderedDict()
    if args.warmup > 0 or timing:
        one_step()
    if timing:
        n_table = max(args.warmup - 1, 2)       # the table (and with it `roofline`) exists for any --warmup; extra steps are untimed
        for plan in (g.plan_fwd, bwd_plan):
            lib.plan_profile_reset(plan)
            lib.plan_profile_select(plan, None)
            lib.plan_profile(plan, 1)
        wev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_table)]
        for i in range(n_table):
            serial_step(wev[i])
        torch.cuda.synchronize()
        collect(table)
        adam_params = g.flat_size - (sum(int(h) - int(l) for l, h in zip(g._skip_lo, g._skip_hi)) if bwd_plan is not g.plan_bwd else 0)
        if bwd_plan is not g.plan_bwd and g._finalized_in_plan:
            adam_params = sum(n.b.size for n in g._fused_nodes)
        table['adam'] = dict(launches=1, flops=0.0, bytes=28.0 * adam_params, ms=adam_collect())
    else:
        for _ in range(max(args.warmup - 1, 0)):
            one_step()
    # ---- timed region: EXACTLY K steps
    for plan in (g.plan_fwd, bwd_plan):
        lib.plan_profile_reset(plan)
        lib.plan_profile(plan, 0)
        if timing:
            lib.plan_profile_select(plan, CONV_FAMILY.encode())
    every = max(1, args.event_every)
    # N > 1: the product's step runs Adam bucket by bucket behind the collectives; it carries no events there
    time_adam = timing and world == 1 and g.overlap_adam and bwd_plan is g.plan_bwd
    adam_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if time_adam else []
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # Host hygiene, as `timeit` does: no cyclic-GC pass inside the timed region.  The launch thread runs ~45 ms ahead of
    # the GPU; a full collection over the interpreter's ~10^6 live objects (triggered by the per-step event objects)
    # stalls it for longer than that and the queues run dry -- measured: 2.6 -> 2.95 ms/step.
    gc.collect()
    gc_was_enabled = gc.isenabled()
    gc.disable()
    sync()
    t0 = time.perf_counter()
    step_ev[0].record()
    n_evsteps = 0
    for i in range(args.steps):
        evstep = timing and (i % every == every - 1)
        if timing:
            for plan in (g.plan_fwd, bwd_plan):
                lib.plan_profile(plan, 1 if evstep else 0)
        n_evsteps += int(evstep)
        one_step(adam_ev[i] if (time_adam and evstep) else None)
        step_ev[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    if gc_was_enabled:
        gc.enable()
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = float(g.loss_buf[0])
    step_ms = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps))

    # ---- the conv / deconv family inside the timed region (rank 0's GPU)
    kern = collections.OrderedDict()
    adam_in_region = None
    if timing:
        if n_evsteps:
            collect(kern)
            if time_adam:
                adam_in_region = adam_collect()
        for plan in (g.plan_fwd, bwd_plan):
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
        "value_note": ("all %d timed steps count, including the %d that carry HIP events on every conv / deconv launch (queue barriers: those "
                       "steps run 10-20 %% slower); --no-kernel-timing gives the undisturbed rate" % (args.steps, n_evsteps)) if timing else "no per-kernel events inside the timed region",
        "config": {"workload": "appflow_offset: AppearanceFlowModel 128x128x3, batch %d per GPU, Adam lr 1e-4, "
                               "random-init weights (reference initialisers)" % args.batch,
                   "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                   "exchange": (None if world == 1 else "%s, %s" % (type(g.comm).__name__, g.dp_mode)),
                   "precision": "conv / deconv / fc GEMMs: fp32 operands split into bf16 hi + lo, three v_mfma_f32_32x32x16_bf16 products per "
                                "fp32 product, fp32 accumulation (error ~1e-6 of the tensor scale; MV3D_DISABLE=4096 selects the exact "
                                "fp32-MFMA kernels); activations, loss, resampler, Adam and all stored tensors fp32",
                   # recorded forward + reverse launches, then: fused step = Adam of the fc biases + its record's advance, the remaining
                   # Adam launch + advance (4); otherwise one Adam launch per bucket + 2 advances (data parallel: + the collectives)
                   "launches_per_step": g.n_launch_fwd + lib.plan_size(bwd_plan) + ((3 if getattr(g, '_finalized_in_plan', False) else 4) if bwd_plan is not g.plan_bwd else len([b for b in g.grad_buckets if b[2] > b[1]]) + 2),
                   "optimiser": ("Adam of the four large fc matrices (97 % of the parameters) fused into their filter-gradient kernels "
                                 "(mv3d_fc_wgrad_adam) and left running under the next step's encoder, one launch for the rest") if bwd_plan is not g.plan_bwd else "bucketed Adam launches"},
        "loss": round(loss, 6),
        "step_ms": {"mean": round(ms_per_step, 4), "median": round(step_ms[len(step_ms) // 2], 4), "min": round(step_ms[0], 4),
                    "max": round(step_ms[-1], 4),
                    "source": "HIP events on the main stream at step boundaries (rank 0); mean = host clock over the region / steps"},
    }

    def family(tab):
        fam = {n: k for n, k in tab.items() if in_conv_family(n)}
        return fam, sum(k['ms'] for k in fam.values()), sum(k['flops'] for k in fam.values()), sum(k['launches'] for k in fam.values())

    def mfma_rates(flops, ms, b3):
        ach = flops / (ms * 1e-3) / 1e12
        if b3:
            return {"achieved": round(ach, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                    "executed_tflops": round(3 * ach, 2), "frac_executed": round(3 * ach / PEAK_BF16_MFMA_TFLOPS, 4),
                    "frac_vs_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4)}
        return {"achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4)}

    if timing and table:
        gpu_ms = sum(k['ms'] for k in table.values())
        b3 = out["dtype"] == "bf16x3"
        src = kern if kern else table
        fam, fam_ms, fam_fl, fam_n = family(src)
        tfam, tfam_ms, tfam_fl, _ = family(table)
        if fam_ms > 0:
            roof = {"bound": "mfma", "kernel": "conv/deconv stack: all %d launches per step of conv2d / conv2d_transpose forward, data-gradient and "
                                               "filter-gradient kernels (%s), slab reductions, split-K tails and filter conversion included at 0 FLOP"
                                               % (fam_n, ", ".join(sorted({n.split('<')[0] for n in fam}))),
                    "mfma_dtype": "bf16 (split: 3 executed products per algorithmic f32 product)" if b3 else "f32"}
            roof.update(mfma_rates(fam_fl, fam_ms, b3))
            roof.update({"launches_per_step": fam_n, "avg_launch_ms": round(fam_ms / max(fam_n, 1), 5), "ms_per_step": round(fam_ms, 4),
                         "algorithmic_gflop_per_step": round(fam_fl / 1e9, 2),
                         "share_of_gpu_time": round(tfam_ms / gpu_ms, 4), "traffic": None,
                         "measured": ("HIP events around every launch of the family inside the timed region, on the stream it is issued on, on every "
                                      "%d-th step (%d of %d steps); filter-gradient kernels run on a side stream next to the data-gradient kernels and "
                                      "Adam, so a launch shares the GPU with concurrent kernels" % (every, n_evsteps, args.steps)) if kern else
                                     "HIP events, warm-up steps, one stream (no event steps inside the timed region)"})
            if tfam_ms > 0:
                al = mfma_rates(tfam_fl, tfam_ms, b3)
                al.update({"ms_per_step": round(tfam_ms, 4), "measured": "HIP events, warm-up steps, all launches on one stream (own durations)"})
                roof["alone"] = al
            # largest single kernel of the family (own durations)
            top = max(((n, k) for n, k in tfam.items() if k['flops'] > 0), key=lambda kv: kv[1]['ms'], default=None)
            if top is not None:
                tn, tk = top
                tr = mfma_rates(tk['flops'], tk['ms'], is_split_bf16(tn))
                tr.update({"kernel": tn, "launches_per_step": tk['launches'], "avg_launch_ms": round(tk['ms'] / tk['launches'], 5)})
                if kern.get(tn, {}).get('ms', 0) > 0:
                    tr["in_timed_region"] = mfma_rates(kern[tn]['flops'], kern[tn]['ms'], is_split_bf16(tn))
                roof["top_kernel"] = tr
            tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    famt = tj.get('_family', {}).get('conv')
                    if famt:      # HBM bytes per step of the family from committed rocprofv3 PMC passes (profiles/), FETCH_SIZE x2-corrected
                        roof["traffic"] = round(famt["hbm_bytes_per_step"] / max(fam_n, 1))      # per average launch, like `achieved` (the per-step figure follows)
                        roof["traffic_per_step"] = famt["hbm_bytes_per_step"]
                        roof["traffic_source"] = famt.get("source")
                    roof["algorithmic_bytes_per_step"] = round(sum(k['bytes'] for k in tfam.values()))
                except Exception:
                    pass
            out["roofline"] = roof
        ad = table.get('adam')
        if ad and ad['ms'] > 0:
            out["adam"] = {"bound": "hbm", "algorithmic_bytes": ad['bytes'], "alone_ms": round(ad['ms'], 4),
                           "alone_gbs": round(ad['bytes'] / (ad['ms'] * 1e-3) / 1e9, 1), "alone_frac_of_hbm_peak": round(ad['bytes'] / (ad['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            if adam_in_region:
                out["adam"].update({"in_region_ms": round(adam_in_region, 4), "in_region_gbs": round(ad['bytes'] / (adam_in_region * 1e-3) / 1e9, 1)})
        # achieved HBM GB/s of the bandwidth-bound pieces of the path -- the sampler (fused with the loss and its
        # gradient), the 3-channel input layer e0 and the 2-channel flow head (forward, data gradient, filter gradients), the
        # gradient finalisation and the fused fc optimiser -- from their ALGORITHMIC bytes and own durations
        hk = {}
        for label, what in (('resample_loss', 'warp + bilinear sampler + pixel loss + flow gradient (tf_utils.py:35-42,18-19)'),
                            ('smallc_band', 'e0: conv 128x128x3 -> 64x64x32, 5x5 stride 2, forward (appearance_flow_model.py:88)'),
                            ('thin_head<2>', 'flow_field: deconv 64x64x32 -> 128x128x2, 5x5 stride 2, forward (appearance_flow_model.py:125)'),
                            ('smallc_band<gmask>', 'flow_field: data gradient'),
                            ('thin_wgrad', 'filter gradients of e0 and flow_field (two launches)'),
                            ('grad_finalize_adam', 'slab sums of all conv filter gradients + their optimiser update'),
                            ('fc_wgrad_adam_b3', 'fc filter gradient + ApplyAdam of the matrix, 24 B per parameter (four launches)')):
            k = table.get(label)
            if k and k['ms'] > 0 and k['bytes'] > 0:
                gbs = k['bytes'] / (k['ms'] * 1e-3) / 1e9
                hk[label] = {"what": what, "launches": k['launches'], "algorithmic_mb": round(k['bytes'] / 1e6, 2), "ms": round(k['ms'], 4),
                             "gbs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4)}
        if hk:
            hk["_measured"] = "HIP events, warm-up steps, all launches on one stream (own durations, launch overhead included); peak %.0f GB/s" % PEAK_HBM_GBS
            out["hbm_kernels"] = hk
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
                rate = ("%7.1f TF/s" % (k['flops'] / k['ms'] / 1e9)) if k['flops'] > 0 else ("%7.0f GB/s" % (k['bytes'] / max(k['ms'], 1e-9) / 1e6))
                inr = kern.get(n, {}).get('ms', 0.0) if kern else 0.0      # the same label inside the timed region (shares the GPU with the other streams)
                print("%-40s launches=%3d  ms/step=%8.4f  %s%s" % (n, k['launches'], k['ms'], rate, ("   in-region %8.4f" % inr) if inr > 0 else ""), file=sys.stderr)
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
        if g.comm is not None:
            g.comm.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
