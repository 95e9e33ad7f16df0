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
