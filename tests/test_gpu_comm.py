"""RCCL through the C ABI (include/mv3d_hip.h mv3d_comm_*) and the data-parallel step's stream schedule, on the one GPU a test
box has: a communicator of world size 1 exercises symbol resolution, every call signature and the comm-stream ordering of
Graph.run_backward_overlapped on real streams; the multi-rank arithmetic of the same schedule is tests/test_dist_cpu.py (gloo,
world 2).  Multi-GPU runs are the driver's (bench.py --gpus N)."""
import numpy as np
import pytest
import torch

from dynamic_multiview_3d_amd import _lib, parallel

pytestmark = pytest.mark.gpu


def test_rccl_collectives_world_one():
    assert _lib.lib().comm_available() == 1
    comm = parallel.RcclComm(0, 1)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.arange(4096, dtype=torch.float32, device='cuda')
    ref = a.clone()
    comm.allreduce_sum_(a, 64, 1024, st)
    comm.reduce_scatter_sum_(a, 0, 4096, st)
    comm.allgather_(a, 128, 512, st)
    torch.cuda.synchronize()
    assert torch.equal(a, ref)
    comm.close()
    comm.close()                                           # idempotent


@pytest.mark.parametrize("mode", ['sharded', 'allreduce'])
def test_data_parallel_schedule_world_one_equals_single_gpu_step(mode, monkeypatch):
    """Graph.run_backward_overlapped (bucket segments -> communication stream: reduce-scatter / Adam / all-gather, or all-reduce /
    Adam) with a world-size-1 RCCL communicator leaves the weights and Adam slots of the plain single-GPU step, bit for bit, over
    three steps: the event / stream ordering between the main stream, the filter-gradient side streams and the communication
    stream loses or reorders nothing."""
    from dynamic_multiview_3d_amd.lowdim_angle import AppFlowLowDimAngle
    from tests.synth import appflow_feeds
    monkeypatch.setenv('MV3D_FUSE_FC_ADAM', '0')
    feeds = appflow_feeds(np.random.default_rng(3), 4)
    res = []
    for dp in (False, True):
        m = AppFlowLowDimAngle({'batch_size': 4, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda', seed=11)
        g = m.graph
        m.feed(**feeds)
        if dp:
            comm = parallel.RcclComm(0, 1)
            m.enable_data_parallel(1, comm=comm, mode=mode)
        for _ in range(3):
            if dp:
                g.run_forward()
                g.run_backward_overlapped(with_adam=True)
            else:
                g.train_step()
        torch.cuda.synchronize()
        g.settle()
        res.append((g.params.cpu().numpy().copy(), g.adam_m.cpu().numpy().copy(), g.adam_v.cpu().numpy().copy(), float(g.adam_state[4])))
        if dp:
            comm.close()
    (p0, m0, v0, b0), (p1, m1, v1, b1) = res
    assert b0 == b1
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_array_equal(v1, v0)
    np.testing.assert_array_equal(p1, p0)


def _two_gpu_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from dynamic_multiview_3d_amd.lowdim_angle import AppFlowLowDimAngle
    from tests.synth import appflow_feeds
    torch.cuda.set_device(rank)                            # before any other GPU call of this process
    parallel.init_from_env('gloo')
    comm = parallel.make_comm(rank, world, 'rccl')
    feeds = appflow_feeds(np.random.default_rng(3), 4 * world)
    shard = {k: v[4 * rank:4 * (rank + 1)] for k, v in feeds.items()}
    m = AppFlowLowDimAngle({'batch_size': 4, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda:%d' % rank, seed=11)
    m.enable_data_parallel(world, comm=comm, mode='sharded')
    m.feed(**shard)
    for _ in range(2):
        m.graph.train_step()
    torch.cuda.synchronize()
    q.put((rank, type(comm).__name__, m.graph.params.cpu().numpy().copy()))
    comm.close()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device: tools/rccl_two_ranks_one_gpu.py)")
def test_two_gpus_data_parallel_equals_one_process_on_the_whole_batch():
    """two ranks on two GPUs, two sharded-optimiser train steps on batch shards; rank 0 == rank 1 bitwise, and both
    equal (to summation-order rounding) a single process that trains on the concatenated batch -- grad_scale 1 / world, the ordering
    between the filter-gradient side streams and the communication stream, and the all-gather of the updated weights, on hardware.
    Skipped on the one-GPU test boxes; the world-2 arithmetic of the same schedule runs on gloo in tests/test_dist_cpu.py."""
    import socket
    import torch.multiprocessing as mp
    from dynamic_multiview_3d_amd.lowdim_angle import AppFlowLowDimAngle
    from tests.synth import appflow_feeds
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        rank, kind, params = q.get(timeout=600)
        got[rank] = params
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(got[0], got[1])
    feeds = appflow_feeds(np.random.default_rng(3), 8)
    m = AppFlowLowDimAngle({'batch_size': 8, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda', seed=11)
    m.feed(**feeds)
    for _ in range(2):
        m.graph.train_step()
    torch.cuda.synchronize()
    m.graph.settle()
    ref = m.graph.params.cpu().numpy()
    # two Adam steps move a weight by at most 2 lr; gradients that differ in the last bits (different summation order of the
    # batch) move it by a tiny fraction of that
    assert np.abs(got[0] - ref).max() <= 0.05 * 2e-4


def _one_gpu_gloo_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from dynamic_multiview_3d_amd.lowdim_angle import AppFlowLowDimAngle
    from tests.synth import appflow_feeds
    try:
        torch.cuda.set_device(0)
        parallel.init_from_env('gloo')
        out = {}
        for mode in ('allreduce', 'sharded'):
            comm = parallel.make_comm(rank, world, 'gloo')
            m = AppFlowLowDimAngle({'batch_size': 4, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda', seed=11)
            g = m.graph
            m.enable_data_parallel(world, comm=comm, mode=mode)
            rng = np.random.default_rng(100 + rank)
            late_seen = False
            for _ in range(3):
                m.feed(**appflow_feeds(rng, 4))
                g.train_step()
                late_seen = late_seen or g._fc_pending
            torch.cuda.synchronize()
            g.settle()
            out[mode] = (g.params.cpu().numpy().copy(), late_seen)
        q.put((rank, None, out))
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001 -- reported to the parent, which fails the test with it
        import traceback
        q.put((rank, traceback.format_exc()[-1500:], None))


def test_two_ranks_on_one_gpu_late_allgather_equals_allreduce():
    """Two ranks share the test box's one GPU and exchange through gloo (RCCL refuses two ranks per device): the sharded step
    with the fc buckets' all-gathers deferred under the next step's encoder -- real kernels, real streams, a world size at which
    a rank only ever computes HALF of every updated bucket -- leaves the weights of the all-reduce schedule, bit for bit, on both
    ranks after three steps with fresh per-rank batches."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_one_gpu_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, err, out = q.get(timeout=900)
        assert err is None, err
        res[rank] = out
    for p in procs:
        p.join(120)
    for r in (0, 1):
        assert res[r]['sharded'][1], "the late all-gather never happened"
        assert not res[r]['allreduce'][1]
        np.testing.assert_array_equal(res[r]['sharded'][0], res[r]['allreduce'][0])
    np.testing.assert_array_equal(res[0]['sharded'][0], res[1]['sharded'][0])


def _one_gpu_mesh_worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from dynamic_multiview_3d_amd.lowdim_angle import AppFlowLowDimAngle
    from tests.synth import appflow_feeds
    try:
        torch.cuda.set_device(0)
        parallel.init_from_env('gloo')
        # the three collectives on a plain buffer first: against the sums computed on the host
        mesh = parallel.make_comm(rank, world, 'mesh')
        n = 4096 + 8
        mine = torch.arange(2 * n, dtype=torch.float32, device='cuda') * (rank + 1) + 0.25 * rank
        both = [torch.arange(2 * n, dtype=torch.float32) * (r + 1) + 0.25 * r for r in range(world)]
        st = torch.cuda.current_stream().cuda_stream
        buf = mine.clone()
        mesh.reduce_scatter_sum_(buf, 0, n, st)
        torch.cuda.synchronize()
        want = both[0] + both[1]
        ok_rs = bool(torch.equal(buf[rank * n:(rank + 1) * n].cpu(), want[rank * n:(rank + 1) * n]))
        mesh.allgather_(buf, 0, n, st)
        torch.cuda.synchronize()
        ok_ag = bool(torch.equal(buf.cpu(), want))
        buf2 = mine[:2 * n - 6].clone()                      # a length that is not a multiple of 4 * world
        mesh.allreduce_sum_(buf2, 0, buf2.numel(), st)
        torch.cuda.synchronize()
        ok_ar = bool(torch.equal(buf2.cpu(), want[:2 * n - 6]))
        out = {'collectives': (ok_rs, ok_ag, ok_ar)}
        for kind in ('gloo', 'mesh'):
            comm = mesh if kind == 'mesh' else parallel.make_comm(rank, world, 'gloo')
            m = AppFlowLowDimAngle({'batch_size': 4, 'learning_rate': 1e-4}, load_tfrec=False, build_loss=True, device='cuda', seed=11)
            g = m.graph
            m.enable_data_parallel(world, comm=comm, mode='sharded')
            rng = np.random.default_rng(100 + rank)
            for _ in range(3):
                m.feed(**appflow_feeds(rng, 4))
                g.train_step()
            torch.cuda.synchronize()
            g.settle()
            g.gather_optimizer_state()
            out[kind] = (g.params.cpu().numpy().copy(), g.adam_m.cpu().numpy().copy())
        mesh.close()
        q.put((rank, None, out))
        dist.destroy_process_group()
    except Exception as e:          # noqa: BLE001 -- reported to the parent, which fails the test with it
        import traceback
        q.put((rank, traceback.format_exc()[-1500:], None))


def test_two_ranks_on_one_gpu_mesh_exchange_equals_gloo():
    """parallel.MeshComm (the mesh-direct option of SURVEY 5 behind the Comm interface): two processes share
    the box's GPU, map each other's flat buffers through hipIpc (mv3d_ipc_*) and pull slices point to point
    (mv3d_mesh_reduce_sum / mv3d_mesh_copy).  The three collectives match host-computed sums exactly, and three sharded train
    steps through it leave the parameters AND the gathered Adam slots of the gloo exchange, bit for bit, on both ranks."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_one_gpu_mesh_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, err, out = q.get(timeout=900)
        assert err is None, err
        res[rank] = out
    for p in procs:
        p.join(120)
    for r in (0, 1):
        assert res[r]['collectives'] == (True, True, True), res[r]['collectives']
        np.testing.assert_array_equal(res[r]['mesh'][0], res[r]['gloo'][0])
        np.testing.assert_array_equal(res[r]['mesh'][1], res[r]['gloo'][1])
    np.testing.assert_array_equal(res[0]['mesh'][0], res[1]['mesh'][0])
