"""Data-parallel plumbing: one process per GPU; the gradient exchange runs on RCCL over xGMI through the C ABI
(include/mv3d_hip.h mv3d_comm_*: RcclComm), torch.distributed only carries the control plane (rendezvous, the 128-byte
RCCL id, barriers) -- and the whole exchange in the CPU rehearsals (TorchComm on gloo).

The reference has no distributed code: training is single process / single device.
The train step shards naturally over the batch -- every sample is independent and the only batch
coupling is the reduce_mean in the loss (tf_utils.py:19) -- so each rank runs the unchanged
forward/backward on its shard and the flat fp32 gradient buffer is SUM-all-reduced, then scaled
by 1/world inside the fused Adam kernel (grad_scale).  That equals the gradient of the global-batch
mean loss; Adam then runs redundantly on every rank from identical weights.
"""
import os

import torch

_MESH_MAX = 8        # include/mv3d_hip.h MV3D_MESH_MAX_RANKS


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank)."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'gloo'            # control plane only: the gradient exchange has its own communicator (make_comm)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


_CONTROL = {}


def control_group():
    """The gloo process group that carries the control plane (agreement flags, the RCCL id, barriers).  The default group when
    it is gloo; otherwise (the caller initialised torch.distributed with nccl) an explicit gloo group, created once -- collective:
    every rank must call it."""
    import torch.distributed as dist
    if 'g' not in _CONTROL:
        _CONTROL['g'] = None if dist.get_backend() == 'gloo' else dist.new_group(backend='gloo')
    return _CONTROL['g']


def _agree(flag, group):
    """True iff `flag` holds on every rank (MIN all-reduce on the control plane)."""
    import torch.distributed as dist
    ok = torch.tensor([1 if flag else 0])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    return int(ok.item()) == 1


def make_comm(rank, world, backend='rccl', fallback='nccl', _fault=None):
    """Communicator of the data-parallel step.  backend 'rccl': RCCL through the C ABI (csrc/comm.hip), the id travels over the
    control-plane group (gloo); 'nccl': torch.distributed's own RCCL binding; 'gloo': the control-plane group itself (CPU
    rehearsals).  torch.distributed must be initialised.

    'mesh': MeshComm below -- peers' buffers mapped by hipIpc, slices pulled point to point, synchronised over the control
    plane.

    The 'rccl' route either succeeds on EVERY rank or every rank falls back together to `fallback`: the ranks agree on the
    control plane BEFORE each blocking step, so an asymmetric failure never leaves one rank in a collective its peers do not enter --
      1. MIN over ranks of "the library and an RCCL are loadable here";
      2. rank 0 creates the id -- or fails -- and broadcasts the id or None;
      3. every rank joins (ncclCommInitRank, itself a blocking collective) under a watchdog (MV3D_COMM_INIT_TIMEOUT seconds,
         default 120): a rank whose peers never arrive gives up instead of waiting forever; MIN over ranks of "joined".
    `_fault` = (rank, 'available' | 'id' | 'init') makes that step raise on that rank (tests/test_dist_cpu.py)."""
    import sys
    import torch.distributed as dist
    if backend == 'rccl':
        grp = control_group()
        comm, err = None, None

        def faulty(step):
            if _fault is not None and _fault[0] == rank and _fault[1] == step:
                raise RuntimeError("injected failure at step %r on rank %d" % (step, rank))

        try:
            faulty('available')
            from . import _lib
            avail = bool(_lib.lib().comm_available())
        except Exception as e:          # noqa: BLE001 -- any failure must be agreed on by all ranks before anyone proceeds
            avail, err = False, e
        stage = 'available'
        good = _agree(avail, grp)
        if good:
            stage = 'id'
            ids = [None]
            if rank == 0:
                try:
                    faulty('id')
                    ids = [RcclComm.unique_id()]
                except Exception as e:  # noqa: BLE001
                    err = e
            if world > 1:
                dist.broadcast_object_list(ids, src=0, group=grp)
            good = ids[0] is not None
        if good:
            stage = 'init'
            try:
                faulty('init')
                comm = RcclComm(rank, world, unique_id=ids[0], timeout=float(os.environ.get('MV3D_COMM_INIT_TIMEOUT', '120')))
            except Exception as e:      # noqa: BLE001
                err = e
            good = _agree(comm is not None, grp)
        if good:
            return comm
        if comm is not None:
            comm.close()
        print("[mv3d] rank %d: RCCL through the C ABI unavailable on some rank (step %r, local error %r): all ranks fall back to %s"
              % (rank, stage, err, fallback), file=sys.stderr)
        backend = fallback
    if backend == 'mesh':
        return MeshComm(rank, world, control_group())
    if backend == 'nccl':
        return TorchComm(dist.new_group(backend='nccl'))
    return TorchComm(control_group())


def bucket_views(flat, bucket_elems):
    """Contiguous views of the flat gradient buffer, largest-first friendly: [flat[a:b], ...]."""
    n = flat.numel()
    return [flat[a:min(a + bucket_elems, n)] for a in range(0, n, bucket_elems)]


def allreduce_sum_(flat, group=None, bucket_elems=None):
    """In-place SUM all-reduce of the flat gradient buffer (optionally in buckets)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    if bucket_elems is None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    else:
        works = [dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group, async_op=True) for v in bucket_views(flat, bucket_elems)]
        for w in works:
            w.wait()
    return flat


def shard_batch(global_batch, rank, world):
    """Even split of the global batch (config 4: 512 = 8 x 64).  Returns (start, stop)."""
    if global_batch % world:
        raise ValueError("global batch %d not divisible by world size %d" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


# ----------------------------------------------------------------------------------------------------- communicators
class TorchComm:
    """The three collectives of the data-parallel step on torch.distributed (gloo on the CPU: tests and rehearsals; the
    product path on GPUs is RcclComm).  Tensors are flat fp32 buffers; `lo`, `n` are element offsets / counts."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allreduce_sum_(self, buf, lo, n, stream=None):
        self.dist.all_reduce(buf[lo:lo + n], op=self.dist.ReduceOp.SUM, group=self.group)

    def reduce_scatter_sum_(self, buf, lo, n, stream=None):
        """in place: slice `rank` of buf[lo : lo + world*n] receives the sum over ranks of that slice"""
        w = self.world
        tmp = buf[lo:lo + w * n].clone()                   # gloo has no reduce-scatter: sum everything, keep the own slice
        self.dist.all_reduce(tmp, op=self.dist.ReduceOp.SUM, group=self.group)
        a = lo + self.rank * n
        buf[a:a + n].copy_(tmp[self.rank * n:(self.rank + 1) * n])

    def allgather_(self, buf, lo, n, stream=None):
        """in place: every rank contributes slice `rank` of buf[lo : lo + world*n]"""
        w = self.world
        a = lo + self.rank * n
        parts = [torch.empty(n, dtype=buf.dtype, device=buf.device) for _ in range(w)]
        self.dist.all_gather(parts, buf[a:a + n].clone(), group=self.group)
        for r, t in enumerate(parts):
            buf[lo + r * n:lo + (r + 1) * n].copy_(t)

    def close(self):
        pass


class RcclComm:
    """RCCL behind the C ABI (csrc/comm.hip).  The 128-byte id comes from rank 0 (`unique_id()`) over the control plane
    (make_comm); every rank then joins on its own, already selected, device."""

    @staticmethod
    def unique_id():
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        if not L.comm_available():
            raise _lib.Mv3dError("no RCCL in this process (librccl.so not found)")
        buf = C.create_string_buffer(128)
        L.comm_unique_id(buf)
        return buf.raw

    def __init__(self, rank, world, control_group=None, unique_id=None, timeout=None):
        import ctypes as C
        import threading
        import torch.distributed as dist
        from . import _lib
        self.lib = _lib.lib()
        self.rank, self.world = rank, world
        self._comm = None
        if not self.lib.comm_available():
            raise _lib.Mv3dError("no RCCL in this process (librccl.so not found)")
        if unique_id is None:            # stand-alone use (world 1, tools): rank 0's id over the given control group
            ids = [RcclComm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0, group=control_group)
            unique_id = ids[0]
        self._id = C.create_string_buffer(unique_id, 128)
        comm = C.c_void_p()
        box = {}
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def join():
            try:
                if dev is not None:
                    torch.cuda.set_device(dev)          # the HIP device is per thread
                self.lib.comm_init(C.byref(comm), rank, world, self._id)
                box['ok'] = True
            except Exception as e:      # noqa: BLE001
                box['err'] = e

        if timeout is None:
            join()
        else:                           # ncclCommInitRank blocks until every rank has arrived: give up after `timeout` seconds
            t = threading.Thread(target=join, daemon=True)
            t.start()
            t.join(timeout)
            if t.is_alive():
                raise TimeoutError("ncclCommInitRank did not return within %.0f s (a peer never joined)" % timeout)
        if 'err' in box:
            raise box['err']
        self._comm = comm

    def allreduce_sum_(self, buf, lo, n, stream):
        self.lib.comm_allreduce_sum(self._comm, buf.data_ptr() + 4 * lo, n, stream)

    def reduce_scatter_sum_(self, buf, lo, n, stream):
        base = buf.data_ptr() + 4 * lo
        self.lib.comm_reduce_scatter_sum(self._comm, base, base + 4 * self.rank * n, n, stream)      # in place: recv = send + rank*n

    def allgather_(self, buf, lo, n, stream):
        base = buf.data_ptr() + 4 * lo
        self.lib.comm_allgather(self._comm, base + 4 * self.rank * n, base, n, stream)

    def close(self):
        if self._comm:
            self.lib.comm_destroy(self._comm)
            self._comm = None


class MeshComm:
    """Mesh-direct reduce-scatter / all-gather: xGMI is a full mesh of point-to-point links (7 x ~153 GB/s per GPU), so
    a ring collective is bound by one link while a rank that pulls its slice of every peer's buffer directly uses all of them.

    Every rank maps its peers' flat buffers through hipIpc (mv3d_ipc_*: handles travel over the control plane the first time a
    buffer is used) and then
      reduce_scatter: sums slice `rank` of all W buffers, in rank order, into its own buffer (mv3d_mesh_reduce_sum: W - 1 remote
                      reads of n elements each, one per link);
      allgather:      pulls the W - 1 other slices from their owners (mv3d_mesh_copy);
      allreduce:      the two in sequence.
    Who may read what when is settled on the control plane: a collective starts with a barrier behind a stream synchronize
    (every rank's producer kernels have finished) and ends with one (every rank's pulls have finished before an owner writes
    again).  That makes this the REFERENCE form of the mesh exchange -- correct by construction, host-synchronous, and what the
    two-ranks-on-one-GPU rehearsal (tests/test_gpu_comm.py) holds equal to the gloo / RCCL result; replacing the two barriers by
    flags in the mapped buffers is the production form once a multi-GPU node is there to measure it on."""

    def __init__(self, rank, world, group=None):
        import torch.distributed as dist
        from . import _lib
        if world > _MESH_MAX:
            raise ValueError("MeshComm: at most %d ranks" % _MESH_MAX)
        self.lib = _lib.lib()
        self.dist, self.group = dist, group
        self.rank, self.world = rank, world
        self._maps = {}             # (local base pointer, bytes) -> [pointer of that buffer on rank r, mapped here]
        self._opened = []

    def _peers(self, buf):
        """device pointers of `buf` on every rank (the local one for this rank), mapping the peers' on first use (collective)"""
        import ctypes as C
        key = (buf.data_ptr(), buf.numel())
        if key in self._maps:
            return self._maps[key]
        handle = C.create_string_buffer(64)
        off = C.c_int64()
        self.lib.ipc_export(buf.data_ptr(), handle, C.byref(off))
        mine = (handle.raw, int(off.value), buf.numel())
        allh = [None] * self.world
        self.dist.all_gather_object(allh, mine, group=self.group)
        ptrs = []
        for r, (h, o, numel) in enumerate(allh):
            if numel != buf.numel():
                raise RuntimeError("MeshComm: rank %d registered %d elements, rank %d %d" % (self.rank, buf.numel(), r, numel))
            if r == self.rank:
                ptrs.append(buf.data_ptr())
            else:
                base = C.c_void_p()
                self.lib.ipc_open(C.create_string_buffer(h, 64), C.byref(base))
                self._opened.append(base.value)
                ptrs.append(base.value + o)
        self._maps[key] = ptrs
        return ptrs

    def _fence(self, stream):
        """every kernel this rank has issued on `stream` has finished, and every rank has got here"""
        if stream is not None:
            torch.cuda.synchronize()
        self.dist.barrier(group=self.group)

    def reduce_scatter_sum_(self, buf, lo, n, stream):
        import ctypes as C
        ptrs = self._peers(buf)
        self._fence(stream)                                  # all ranks' gradients are final
        off = 4 * (lo + self.rank * n)
        srcs = (C.c_void_p * self.world)(*[p + off for p in ptrs])
        self.lib.mesh_reduce_sum(srcs, self.world, ptrs[self.rank] + off, n, stream)
        self._fence(stream)                                  # nobody still reads a slice its owner is about to overwrite

    def allgather_(self, buf, lo, n, stream):
        ptrs = self._peers(buf)
        self._fence(stream)                                  # every owner's slice is final
        for d in range(1, self.world):
            r = (self.rank + d) % self.world                 # every rank starts at a different peer: all links busy at once
            off = 4 * (lo + r * n)
            self.lib.mesh_copy(ptrs[self.rank] + off, ptrs[r] + off, n, stream)
        self._fence(stream)

    def allreduce_sum_(self, buf, lo, n, stream):
        W = self.world
        per = -(-n // (4 * W)) * 4                           # slice length: multiple of 4 elements; the last slice may be shorter
        ptrs = self._peers(buf)
        import ctypes as C
        self._fence(stream)
        a = min(self.rank * per, n)
        cnt = min(per, n - a)
        if cnt > 0:
            off = 4 * (lo + a)
            srcs = (C.c_void_p * W)(*[p + off for p in ptrs])
            self.lib.mesh_reduce_sum(srcs, W, ptrs[self.rank] + off, cnt, stream)
        self._fence(stream)
        for d in range(1, W):
            r = (self.rank + d) % W
            a = min(r * per, n)
            cnt = min(per, n - a)
            if cnt > 0:
                self.lib.mesh_copy(ptrs[self.rank] + 4 * (lo + a), ptrs[r] + 4 * (lo + a), cnt, stream)
        self._fence(stream)

    def close(self):
        for b in self._opened:
            try:
                self.lib.ipc_close(b)
            except Exception:       # noqa: BLE001 -- the owner may already be gone at teardown
                pass
        self._opened, self._maps = [], {}
