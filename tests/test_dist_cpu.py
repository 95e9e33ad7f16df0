"""world_size-2 gloo test of the data-parallel gradient exchange: SUM all-reduce of
the flat gradient buffer + 1/world scaling reproduces the gradient of the global-batch mean loss.
The per-rank gradients come from the oracle (no GPU here); the collective and the scaling are the
product's (dynamic_multiview_3d_amd.parallel)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from dynamic_multiview_3d_amd import parallel
    from oracle import models as omodels, ops
    r, w, _ = parallel.init_from_env('gloo')
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(0)
    n, h = 4, 8
    img = rng.standard_normal((n, h, h, 3)).astype(np.float32)
    tgt = rng.standard_normal((n, h, h, 3)).astype(np.float32)
    wgt = (rng.standard_normal((3, 3, 3, 4)) * 0.2).astype(np.float32)
    bias = np.zeros(4, np.float32)
    wd = (rng.standard_normal((3, 3, 3, 4)) * 0.2).astype(np.float32)       # deconv back to 3 channels

    def grads(lo, hi):
        x, t = img[lo:hi], tgt[lo:hi]
        y = ops.conv2d_fwd(x, wgt, bias, 1, 1)
        a = ops.absact_fwd(y, 'lrelu')
        z = ops.deconv2d_fwd(a, wd, (h, h), 1, 1)
        dz = ops.euclidean_loss_bwd(z, t)
        da, dwd = ops.deconv2d_bwd(a, wd, dz, 1, 1)
        dy = ops.absact_bwd(y, da, 'lrelu')
        _, dw, db = ops.conv2d_bwd(x, wgt, dy, 1, 1, need_dx=False)
        return np.concatenate([dw.ravel(), db.ravel(), dwd.ravel()])

    lo, hi = parallel.shard_batch(n, rank, world)
    flat = torch.from_numpy(grads(lo, hi).copy())
    parallel.allreduce_sum_(flat, bucket_elems=50)
    flat *= 1.0 / world                                   # the grad_scale the Adam kernel applies
    full = grads(0, n)
    q.put((rank, float(np.abs(flat.numpy() - full).max()), float(np.abs(full).max())))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_global_batch_gradient():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, scale in res:
        assert err < 1e-6 * max(scale, 1.0), (rank, err, scale)


def _overlap_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from dynamic_multiview_3d_amd import parallel
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    parallel.init_from_env('gloo')
    m = AppearanceFlowModel({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, device='cpu')
    g = m.graph
    m.enable_data_parallel(world)
    # the recorded kernels cannot run without a GPU: replace the segment launcher, keep the product's
    # bucket schedule and collectives
    segments = []

    class FakeLib:
        def plan_run_range_multi(self, plan, begin, end, stream, side_streams, nside, flags):
            segments.append((begin, end))
    real_lib, g.lib = g.lib, FakeLib()
    g._stream_ptr = lambda: None
    gen = torch.Generator().manual_seed(100 + rank)
    g.grads.copy_(torch.randn(g.flat_size, generator=gen))
    mine = g.grads.clone()
    g.run_backward_overlapped()
    g.lib = real_lib
    other = torch.randn(g.flat_size, generator=torch.Generator().manual_seed(100 + (1 - rank)))
    err = float((g.grads - (mine + other)).abs().max())
    q.put((rank, err, segments, [(e, lo, hi) for e, lo, hi in g.grad_buckets], g.n_launch_bwd, g.flat_size))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_overlapped_bucket_allreduce_two_ranks():
    """The product's bucket schedule (Graph.grad_buckets + run_backward_overlapped) on gloo, world 2:
    segments partition the backward plan, buckets partition the flat gradient buffer from the end
    towards the start, and after the exchange every element is the sum over ranks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, err, segments, buckets, nbwd, flat in res:
        assert err < 1e-5, (rank, err)
        assert segments[0][0] == 0 and segments[-1][1] == nbwd
        assert all(a[1] == b[0] for a, b in zip(segments, segments[1:]))
        assert buckets[0][2] == flat and buckets[-1][1] == 0
        assert all(a[1] == b[2] for a, b in zip(buckets, buckets[1:]))          # contiguous, descending
        full = [b for b in buckets if b[2] > b[1]]          # an empty bucket marks the end of the fc layers (Adam gate)
        # the decoder's conv filters (first) and the encoder's (last) are small buckets of their own -- the next step's first
        # launch reads them --, the fc matrices in between travel in >= 60 MB buckets
        assert len(full) >= 5 and all((hi - lo) * 4 >= 60e6 for _, lo, hi in full[1:-1])
        assert (full[0][2] - full[0][1]) * 4 < 8e6


def test_late_allgather_set_is_the_fc_buckets():
    """The sharded data-parallel step gathers a bucket's updated slices late (under the next step's encoder) only when no early
    forward launch reads it: conv filters are read by launch 0 (filter conversion), so both conv buckets stay early; the four fc
    buckets are first read in forward order fc1 < a3 < a4 < a5, all behind the encoder."""
    from dynamic_multiview_3d_amd.appearance_flow_model import AppearanceFlowModel
    m = AppearanceFlowModel({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cpu')
    g = m.graph
    full = [(b, u) for b, u in zip(g.grad_buckets, g._bucket_first_use) if b[2] > b[1]]
    assert full[0][1] == 0 and full[-1][1] == 0
    late = [u for _, u in full[1:-1]]
    assert len(late) == 4 and all(u >= g.pipeline_dp_min_idx for u in late)
    assert late == sorted(late, reverse=True)          # buckets come in reverse-pass order, uses in forward order
    names = {v.name: v for v in g.variables.values()}
    fc1 = next(b for b, _ in full if b[1] <= names['fc1/Matrix'].offset < b[2])
    assert g._first_param_use(fc1[1], fc1[2]) == min(late)


def test_shard_batch():
    from dynamic_multiview_3d_amd.parallel import shard_batch, bucket_views
    assert shard_batch(512, 3, 8) == (192, 256)
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)
    v = bucket_views(torch.arange(10.0), 4)
    assert [x.numel() for x in v] == [4, 4, 2]


class _CpuLib:
    """Stands in for libmv3d_hip.so on the CPU: segments are no-ops (the test supplies the gradients), the optimiser is the
    oracle's TF-Adam on the flat buffers -- so the product's data-parallel schedule (Graph.run_backward_overlapped) runs end to
    end without a GPU."""

    def __init__(self, g):
        self.g = g
        self.adam_calls = []

    def plan_run_range_multi(self, plan, begin, end, stream, side_streams, nside, flags):
        pass

    def adam_step_dev(self, count, p, gr, m, v, state, nskip, slo, shi, stream):
        from oracle import ops
        g = self.g
        lo = (p - g.params.data_ptr()) // 4
        st = g.adam_state.numpy()
        self.adam_calls.append((lo, count))
        sl = slice(lo, lo + count)
        grad = g.grads.numpy()[sl] * st[6]
        ops.adam_step(g.params.numpy()[sl], grad, g.adam_m.numpy()[sl], g.adam_v.numpy()[sl], st[4], st[5], float(st[0]))

    def adam_advance(self, state, stream):
        rec = (state - self.g.adam_state.data_ptr()) // 4          # 0: the main record, 8: the fused fc optimiser's copy
        st = self.g.adam_state.numpy()[rec:rec + 8]
        st[4] *= st[1]
        st[5] *= st[2]


def _dp_modes_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from dynamic_multiview_3d_amd import parallel
    from dynamic_multiview_3d_amd.lowdim_angle import AppFlowLowDimAngle
    parallel.init_from_env('gloo')
    out = {}
    for mode in ('allreduce', 'sharded'):
        m = AppFlowLowDimAngle({'batch_size': 2, 'learning_rate': 1e-4}, load_tfrec=False, device='cpu', seed=7)
        g = m.graph
        m.enable_data_parallel(world, mode=mode)
        fake = _CpuLib(g)
        g.lib = fake
        g._stream_ptr = lambda: None
        gen = torch.Generator().manual_seed(1000 + rank)
        for step in range(3):
            g.grads.copy_(torch.randn(g.flat_size, generator=gen) * 1e-2)      # this rank's gradients of the step
            g.run_backward_overlapped(with_adam=True)
        if mode == 'sharded':
            # a rank's slots are complete on its own slices only: state_dict() refuses until the collective gather has run
            try:
                g.state_dict()
                refused = False
            except RuntimeError:
                refused = True
            g.gather_optimizer_state()
            sd = g.state_dict()
        else:
            refused = True
            sd = g.state_dict()
        out[mode] = (g.params.clone(), g.adam_m.clone(), g.adam_v.clone(), float(g.adam_state[4]), sum(c for _, c in fake.adam_calls), g.flat_size, sd, refused)
    pa, ma, va, b1a, na, flat, sda, _ = out['allreduce']
    ps, ms, vs, b1s, ns, _, sds, refused = out['sharded']
    # checkpoints: every entry (weights, Adam, Adam_1, beta powers) of the sharded run equals the all-reduce run's
    ckpt_same = refused and list(sda) == list(sds) and all(torch.equal(sda[k], sds[k]) for k in sda) and \
        any(k.endswith('/Adam_1') and float(sds[k].abs().sum()) > 0 for k in sds)
    # every rank ends with the same weights in both modes; in sharded mode a rank only updated 1/world of every bucket, so its
    # Adam slots are complete on its own slices only -- compare the weights (all-gathered) everywhere, the slots via a SUM
    import torch.distributed as dist
    same = bool(torch.equal(pa, ps))
    other = pa.clone()
    dist.broadcast(other, src=0)
    q.put((rank, same, bool(torch.equal(other, pa)), b1a, b1s, na, ns, flat, ckpt_same))
    dist.destroy_process_group()


def test_sharded_optimiser_equals_allreduce_two_ranks():
    """reduce-scatter -> Adam on 1/world of each bucket -> all-gather leaves bit-identical weights to all-reduce + redundant Adam,
    on both ranks, after three steps (gloo, world 2; the product's bucket schedule, the oracle's TF-Adam in place of the HIP
    kernel); the sharded ranks ran the optimiser over exactly half of the parameters each."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_modes_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, same, ranks_equal, b1a, b1s, na, ns, flat, ckpt_same in res:
        assert same, "sharded and all-reduce modes diverged on rank %d" % rank
        assert ckpt_same, "sharded checkpoint (after gather_optimizer_state) differs from the all-reduce one on rank %d" % rank
        assert ranks_equal, "ranks hold different weights"
        assert abs(b1a - 0.9 ** 4) < 1e-6 and b1a == b1s
        assert na == 3 * flat and ns * 2 == na


def _make_comm_fault_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ['MV3D_COMM_INIT_TIMEOUT'] = '20'
    import time
    from dynamic_multiview_3d_amd import parallel
    parallel.init_from_env('gloo')
    res = []
    # a failure on ONE rank at each blocking step: the peers must not be left inside a collective the failed rank never enters
    for fault in ((0, 'available'), (1, 'available'), (0, 'id'), (1, 'init'), (0, 'init')):
        t0 = time.time()
        comm = parallel.make_comm(rank, world, 'rccl', fallback='gloo', _fault=fault)
        buf = torch.full((8,), float(rank + 1))
        comm.allreduce_sum_(buf, 0, 8)                      # the fallback communicator works, on every rank
        res.append((fault, type(comm).__name__, float(buf[0]), time.time() - t0))
    q.put((rank, res))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_make_comm_falls_back_together_when_one_rank_fails():
    """ADVICE r2 / VERDICT r2 weak #8: a rank that fails before the id broadcast or before ncclCommInitRank used to leave its
    peers in a mismatched collective (hang until the gloo timeout).  The ranks now agree before every blocking step, so a
    failure on either rank at any step makes BOTH fall back, promptly (no GPU here: the surviving rank's own join fails or
    times out under the 20 s watchdog, never hangs)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_make_comm_fault_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in (0, 1):
        for fault, kind, total, dt in res[rank]:
            assert kind == 'TorchComm', (rank, fault, kind)
            assert total == 3.0
            assert dt < 60, "rank %d waited %.0f s at fault %r" % (rank, dt, fault)
