#!/usr/bin/env python
"""Can two RCCL ranks share one GPU on this box?  (informational: decides whether a 2-rank GPU test is possible there)"""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from dynamic_multiview_3d_amd import parallel
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        comm = parallel.RcclComm(rank, world)
        a = torch.full((1024,), float(rank + 1), device='cuda')
        comm.allreduce_sum_(a, 0, 1024, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print("rank", rank, "allreduce ->", float(a[0]), flush=True)
        comm.close()
    except Exception as e:
        print("rank", rank, "failed:", repr(e)[:300], flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
