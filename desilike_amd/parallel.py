"""Walker / parameter-point sharding across the GPUs of a node.

The reference's only parallelism is data-parallel over points (``vmap(..., backend='mpi')`` desilike/base.py:291-378:
``Scatterv`` of the points, local loop, gather; ``mpi.local_size`` desilike/mpi.py:145-149; samplers broadcast the log-posteriors,
desilike/samplers/base.py:196-200).  Here: one process per GPU, rank r evaluates a contiguous share of the rows and the only exchange is ONE
all-gather of the per-point results (log-posteriors: B / G doubles per rank).

Two process groups carry that exchange:

* :class:`RcclGroup` -- the product path on GPUs: RCCL over xGMI through the library's own C ABI (``dl_comm_*``, include/desilike_amd.h), device buffers,
  enqueued on the caller's HIP stream (no host synchronisation, no ``torch.distributed``);
* :class:`TorchGroup` -- ``torch.distributed`` (``gloo``) for the CPU tests of the host logic and for smoke runs of the N > 1 code path on a 1-GPU box.

Payloads are kilobytes: the exchange is latency-bound, so it is never split.
"""
import os

import numpy as np


def local_slice(size, rank, world):
    """Contiguous share of ``size`` items for ``rank`` out of ``world`` (same rule as desilike/mpi.py:145-149)."""
    return slice(rank * size // world, (rank + 1) * size // world)


def chunk_rows(size, world):
    """Rows per rank when the shares must have equal length (in-place all-gather): ``ceil(size / world)``; rank r holds rows
    ``[r c, min((r + 1) c, size))`` (the last ranks' shares may be shorter or empty)."""
    return (int(size) + world - 1) // world


_stores = []   # TCP stores of this process (rank 0 hosts them: they must outlive the other ranks' reads)
_store_calls = [0]


def _exchange_bytes(payload, rank, world):
    """Ship ``payload`` (bytes, made by rank 0) to every rank over a host channel: the initialised ``torch.distributed`` group if there is one, else a TCP store at
    ``MASTER_ADDR : DL_COMM_PORT`` (default ``MASTER_PORT + 1``; torchrun's own store sits on MASTER_PORT)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        box = [payload if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    from datetime import timedelta
    host = os.environ.get('MASTER_ADDR', '127.0.0.1')
    if not _stores:
        if os.environ.get('TORCHELASTIC_USE_AGENT_STORE', '') == 'True' and 'DL_COMM_PORT' not in os.environ:
            # under torchrun the agent already hosts a store at MASTER_ADDR:MASTER_PORT (what torch's own env:// rendezvous connects to): every rank is a client
            _stores.append(dist.TCPStore(host, int(os.environ['MASTER_PORT']), world, is_master=False, timeout=timedelta(seconds=300)))
        else:
            port = int(os.environ.get('DL_COMM_PORT', int(os.environ.get('MASTER_PORT', 29500)) + 1))
            _stores.append(dist.TCPStore(host, port, world, is_master=(rank == 0), timeout=timedelta(seconds=300), wait_for_workers=False))
    store = _stores[0]
    # (the restart count keeps the attempts of an elastic job apart: same run id and agent store, but a stale id of the previous attempt must not be picked up)
    key = 'desilike_amd/comm_id/{}/{}/{:d}'.format(os.environ.get('TORCHELASTIC_RUN_ID', 'job'), os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'), _store_calls[0])
    _store_calls[0] += 1
    if rank == 0:
        store.set(key, payload)
        return payload
    return bytes(store.get(key))


class RcclGroup(object):
    """The GPUs of one node as a process group: direct RCCL binding of the C ABI (``dl_comm_*``).  All arrays are device tensors; every call is enqueued on a HIP
    stream (default: torch's current stream of the device) and returns at once."""
    backend = 'rccl'

    def __init__(self, device, rank=None, world=None):
        import ctypes
        import torch
        from ._lib import load, LibraryError, rccl_library_path
        self._lib = lib = load()
        self.rank = int(os.environ.get('RANK', 0)) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', 1)) if world is None else int(world)
        self.device = int(device)
        self._torch_device = torch.device('cuda', self.device)
        path = rccl_library_path()
        cpath = path.encode() if path else None
        uid = None
        if self.rank == 0:
            buf = ctypes.create_string_buffer(128)
            if lib.dl_comm_unique_id(buf, cpath) != 0: raise LibraryError(lib.dl_last_error(None).decode())
            uid = buf.raw
        uid = _exchange_bytes(uid, self.rank, self.world)
        handle = ctypes.c_void_p()
        if lib.dl_comm_create(ctypes.byref(handle), self.device, self.rank, self.world, uid, cpath) != 0:
            raise LibraryError(lib.dl_last_error(None).decode())
        self._handle = handle
        self.rccl_version = int(lib.dl_comm_info(handle, b'rccl_version'))
        self._token = None

    @property
    def handle(self):
        return self._handle

    def _stream(self, stream):
        import torch
        return torch.cuda.current_stream(self._torch_device).cuda_stream if stream is None else stream

    def _check(self, rc):
        from ._lib import LibraryError
        if rc != 0: raise LibraryError(self._lib.dl_last_error(None).decode())

    def allgather_into(self, recv, send, stream=None):
        """``recv [world * count]`` <- every rank's ``send [count]`` (float64 device tensors; in place when ``send`` is this rank's slice of ``recv``)."""
        import ctypes
        import torch
        count = send.numel()
        assert send.is_cuda and recv.is_cuda and send.dtype == recv.dtype == torch.float64 and send.is_contiguous() and recv.is_contiguous()
        assert recv.numel() == self.world * count, (recv.shape, send.shape, self.world)
        self._check(self._lib.dl_comm_allgather_f64(self._handle, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()), count, ctypes.c_void_p(self._stream(stream))))
        return recv

    def allgather(self, tensor, stream=None):
        """``tensor [count, ...]`` (same shape on every rank; numpy arrays are staged through the device) -> ``[world * count, ...]``."""
        import torch
        is_numpy = isinstance(tensor, np.ndarray)
        send = torch.as_tensor(np.ascontiguousarray(tensor, dtype='f8') if is_numpy else tensor).to(self._torch_device, torch.float64).contiguous()
        recv = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=torch.float64, device=self._torch_device)
        self.allgather_into(recv, send, stream=stream)
        return recv.cpu().numpy() if is_numpy else recv

    def broadcast(self, tensor, src=0, stream=None):
        import ctypes
        import torch
        is_numpy = isinstance(tensor, np.ndarray)
        buf = torch.as_tensor(np.ascontiguousarray(tensor, dtype='f8') if is_numpy else tensor).to(self._torch_device, torch.float64).contiguous()
        self._check(self._lib.dl_comm_broadcast_f64(self._handle, ctypes.c_void_p(buf.data_ptr()), buf.numel(), int(src), ctypes.c_void_p(self._stream(stream))))
        if is_numpy: return buf.cpu().numpy().reshape(np.shape(tensor))
        if buf.data_ptr() != tensor.data_ptr(): tensor.copy_(buf)
        return tensor

    def barrier(self):
        """Every rank has enqueued and finished everything before this point (one 8-byte all-gather + a device synchronisation)."""
        import torch
        if self._token is None:
            self._token = (torch.zeros(1, dtype=torch.float64, device=self._torch_device), torch.zeros(self.world, dtype=torch.float64, device=self._torch_device))
        self.allgather_into(self._token[1], self._token[0])
        torch.cuda.synchronize(self._torch_device)

    def max(self, value):
        """Maximum of a host scalar over the ranks."""
        import torch
        send = torch.tensor([float(value)], dtype=torch.float64, device=self._torch_device)
        recv = torch.empty(self.world, dtype=torch.float64, device=self._torch_device)
        self.allgather_into(recv, send)
        return float(recv.max().item())

    def close(self):
        if getattr(self, '_handle', None):
            self._lib.dl_comm_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TorchGroup(object):
    """``torch.distributed`` process group (``gloo``: CPU tests / smoke runs of the N > 1 path on a 1-GPU box); same surface as :class:`RcclGroup`."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = device   # tensors are moved there for the collective (None: where they are; gloo needs CPU tensors)
        self.handle = None

    def _place(self, tensor):
        if self.backend == 'gloo': return tensor.cpu()
        return tensor if self.device is None else tensor.to(self.device)

    def allgather_into(self, recv, send, stream=None):
        r, s = self._place(recv), self._place(send)
        self.dist.all_gather_into_tensor(r, s.contiguous(), group=self.group)
        if r.data_ptr() != recv.data_ptr(): recv.copy_(r)
        return recv

    def allgather(self, tensor, stream=None):
        import torch
        is_numpy = isinstance(tensor, np.ndarray)
        send = torch.as_tensor(np.ascontiguousarray(tensor) if is_numpy else tensor)
        home = send.device
        send = self._place(send).contiguous()
        recv = torch.empty((self.world * send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        return recv.numpy() if is_numpy else recv.to(home)

    def broadcast(self, tensor, src=0, stream=None):
        import torch
        is_numpy = isinstance(tensor, np.ndarray)
        buf = torch.as_tensor(np.ascontiguousarray(tensor) if is_numpy else tensor)
        placed = self._place(buf).contiguous()
        self.dist.broadcast(placed, src=src, group=self.group)
        if is_numpy: return placed.cpu().numpy()
        if placed.data_ptr() != tensor.data_ptr(): tensor.copy_(placed)
        return tensor

    def barrier(self):
        self.dist.barrier(group=self.group)

    def max(self, value):
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def close(self):
        pass


_default_group = [None]


def set_default_group(group):
    """Process group picked up by samplers created without an explicit ``sharding``."""
    _default_group[0] = group


def get_default_group():
    """The group set by :func:`set_default_group`, else the initialised ``torch.distributed`` default group, else ``None`` (single process)."""
    if _default_group[0] is not None:
        return _default_group[0]
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return TorchGroup()
    return None


def init_group(device=None, backend=None):
    """Process group of this job from the launcher's environment (``RANK`` / ``WORLD_SIZE`` / ``LOCAL_RANK`` / ``MASTER_ADDR`` / ``MASTER_PORT``, as set by
    ``torchrun`` or ``bench.py --gpus N``): RCCL through the C ABI on GPUs; ``backend='gloo'``: ``torch.distributed`` (CPU tests, 1-GPU smoke runs).
    Returns ``None`` for a single process."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world <= 1:
        return None
    backend = backend or os.environ.get('DL_COMM_BACKEND', 'rccl')
    if backend == 'rccl':
        if device is None: device = int(os.environ.get('LOCAL_RANK', 0))
        group = RcclGroup(device)
    else:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=int(os.environ['RANK']), world_size=world)
        group = TorchGroup(device=None if backend == 'gloo' else device)
    set_default_group(group)
    return group


class WalkerSharding(object):
    """Evaluate a batch function on the local share of the rows and all-gather the results on every rank."""

    def __init__(self, group=None, device=None, min_shard_rows=4096):
        """``group``: :class:`RcclGroup` / :class:`TorchGroup` (default: :func:`get_default_group`; a raw ``torch.distributed`` group is wrapped).
        ``min_shard_rows``: below this many rows every rank evaluates ALL rows and nothing is exchanged.  The evaluation is latency-bound at small batches
        (34 us for 32 as for 256 points of the two-tracer likelihood) while an all-gather costs tens of microseconds: sharding 512 walkers over 8 GPUs is not
        faster than evaluating them redundantly (the kernels are deterministic: every rank gets the same bits); the ranks then hold DUPLICATES of one chain
        (their random generators are synchronised, see ``BasePosteriorSampler``) -- run independent chains by giving each process its own seed and no group.
        0: always shard."""
        if group is None:
            group = get_default_group()
        elif group is False:   # explicitly local: no group even if a default one is set (chains of a chain-parallel sampler)
            group = None
        elif not hasattr(group, 'allgather'):
            group = TorchGroup(group=group, device=device)
        self.min_shard_rows = int(min_shard_rows)
        self.group = group
        self.active = group is not None
        self.rank = group.rank if self.active else 0
        self.world = group.world if self.active else 1
        self.device = device

    def slice(self, size):
        return local_slice(size, self.rank, self.world)

    def sharded(self, size):
        """True if a batch of ``size`` rows is split over the ranks (and exchanged)."""
        return self.active and self.world > 1 and size >= self.min_shard_rows

    def allgather_rows(self, local, size):
        """``local``: array / tensor with the rows of this rank's slice of ``size`` rows -> all ``size`` rows, on every rank."""
        import torch
        if not self.active or self.world == 1:
            return local
        is_numpy = isinstance(local, np.ndarray)
        tensor = torch.as_tensor(local) if is_numpy else local
        counts = [local_slice(size, rank, self.world) for rank in range(self.world)]
        counts = [sl.stop - sl.start for sl in counts]
        nmax = max(counts)
        trailing = tuple(tensor.shape[1:])
        padded = torch.zeros((nmax,) + trailing, dtype=tensor.dtype, device=tensor.device)
        padded[:tensor.shape[0]] = tensor
        gathered = self.group.allgather(padded)   # the single collective of the path
        gathered = gathered.reshape((self.world, nmax) + trailing)
        out = torch.cat([gathered[rank, :count] for rank, count in enumerate(counts)], dim=0)
        return out.cpu().numpy() if is_numpy else out

    def map(self, func, values):
        """``func(values_local) -> array[len(values_local), ...]`` applied to this rank's slice; returns the full result everywhere."""
        values = np.asarray(values)
        if not self.sharded(len(values)):
            return np.asarray(func(values))
        if os.environ.get('DL_CHECK_SHARDING', '0') == '1':
            # debugging aid: the rows must be the same on every rank (they are when the ranks' random generators are synchronised)
            digest = np.array([[np.nansum(values), np.nansum(values * np.arange(1, values.size + 1).reshape(values.shape))]])
            everyone = np.asarray(self.group.allgather(digest))
            if not (everyone == everyone[0]).all():
                raise RuntimeError('sharded evaluation: the ranks hold different rows (unsynchronised random generators?)')
        sl = self.slice(len(values))
        local = np.asarray(func(values[sl]), dtype='f8')
        return self.allgather_rows(np.ascontiguousarray(local), len(values))

    def map_logposterior(self, ctx, values, offset=0.):
        """Log-posteriors of ``values [B, P]`` (host array, identical on every rank) through the device context ``ctx``: rank r evaluates rows
        ``[r c, (r + 1) c)``, ``c = ceil(B / world)``, straight into its slice of the gathered device buffer; one in-place all-gather on the same stream;
        one copy back to the host.  Nothing else touches the host."""
        import torch
        values = np.ascontiguousarray(np.atleast_2d(values), dtype='f8')
        B = len(values)
        if not self.sharded(B) or not isinstance(self.group, RcclGroup):
            if not self.sharded(B):
                return ctx.eval_logposterior_host(values)[0] + offset
            return self.map(lambda rows: ctx.eval_logposterior_host(rows)[0] + offset if len(rows) else np.zeros(0), values)
        device = torch.device('cuda', ctx.device)
        count = chunk_rows(B, self.world)
        lo = min(self.rank * count, B)
        hi = min(lo + count, B)
        theta = torch.as_tensor(values[lo:hi], device=device)
        gathered = torch.zeros(self.world * count, dtype=torch.float64, device=device)
        mine = gathered[self.rank * count:(self.rank + 1) * count]
        if hi > lo:
            ctx.eval_logposterior(theta, mine[:hi - lo])
        self.group.allgather_into(gathered, mine)
        return gathered[:B].cpu().numpy() + offset

    def broadcast(self, array, src=0):
        """Broadcast a numpy array from ``src`` (walker positions must be identical on all ranks: samplers/base.py:45-54)."""
        if not self.active or self.world == 1:
            return array
        return np.asarray(self.group.broadcast(np.ascontiguousarray(array, dtype='f8'), src=src)).reshape(np.shape(array))


class PipelinedAllGather(object):
    """Double-buffered asynchronous all-gather of per-point results, for drivers that keep several independent walker ensembles (chains) in flight.

    ``submit(slot, local)`` starts the all-gather of ensemble ``slot``'s local results on the collective's own stream and returns at once: the caller goes
    on enqueueing the evaluation of the *other* ensemble, whose kernels overlap the (latency-bound, kilobyte-sized) collective.  ``result(slot)`` makes the
    current stream wait for that collective and returns the gathered rows.  One collective per ensemble step, never split (xGMI is point-to-point:
    a small all-gather costs a ring latency whatever its size).  A slot must be drained (``result``) before it is submitted again.

    With an :class:`RcclGroup` the collective runs on a side HIP stream ordered by events against the evaluation stream (no host synchronisation); with a
    :class:`TorchGroup` it is ``torch.distributed``'s asynchronous ``all_gather_into_tensor``.
    """

    def __init__(self, shape, dtype, device, nslots=2, group=None, force_collective=False):
        import torch
        if group is None:
            group = get_default_group()
        elif not hasattr(group, 'allgather'):
            group = TorchGroup(group=group)
        self.group = group
        self.active = group is not None
        self.world = group.world if self.active else 1
        self.collective = self.active and (self.world > 1 or force_collective)   # (forcing: exercises the backend with a single rank)
        self.rccl = isinstance(group, RcclGroup)
        shape = tuple(shape)
        self.gathered = [torch.empty((self.world * shape[0],) + shape[1:], dtype=dtype, device=device) for _ in range(nslots)]
        self.work = [None] * nslots
        self.local = [None] * nslots
        if self.rccl and self.collective:
            self.side = torch.cuda.Stream(device=device)
            self.done = [torch.cuda.Event() for _ in range(nslots)]

    def submit(self, slot, local):
        import torch
        if self.work[slot] is not None:
            raise RuntimeError('slot {:d} resubmitted before its result was taken'.format(slot))
        self.local[slot] = local   # keep the input alive until the collective has consumed it
        if not self.collective:
            self.work[slot] = True
        elif self.rccl:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(local.device))
            self.side.wait_event(ready)                                     # the side stream starts once the evaluation that filled `local` is done
            self.group.allgather_into(self.gathered[slot], local, stream=self.side.cuda_stream)
            self.done[slot].record(self.side)
            self.work[slot] = True
        else:
            placed_in, placed_out = self.group._place(local), self.group._place(self.gathered[slot])
            self.work[slot] = (self.group.dist.all_gather_into_tensor(placed_out, placed_in, group=self.group.group, async_op=True), placed_out)

    def pending(self, slot):
        return self.work[slot] is not None

    def result(self, slot):
        import torch
        work = self.work[slot]
        if work is None:
            raise RuntimeError('nothing submitted in slot {:d}'.format(slot))
        self.work[slot] = None
        if not self.collective:
            return self.local[slot]
        if self.rccl:
            torch.cuda.current_stream(self.gathered[slot].device).wait_event(self.done[slot])   # stream-ordered: the current stream waits for the collective
            return self.gathered[slot]
        handle, placed_out = work
        handle.wait()   # stream-ordered on GPUs (the current stream waits for the collective), blocking on CPU backends
        if placed_out.data_ptr() != self.gathered[slot].data_ptr(): self.gathered[slot].copy_(placed_out)
        return self.gathered[slot]


class BucketedAllGather(object):
    """All-gather of per-point results for drivers with many independent walker ensembles (chains) in flight: the results of ``steps_per_bucket`` consecutive
    ensemble steps travel in ONE asynchronous collective (double-buffered buckets on top of :class:`PipelinedAllGather`).

    A kilobyte-sized all-gather costs a fixed issue time plus a ring latency whatever its payload, so small exchanges are bucketed, exactly like gradient
    buckets in data-parallel training.
    ``slot()`` returns the [B] slice the next step must write its results into; ``advance()`` closes the step and launches the bucket's collective when
    it is full; ``results()`` drains everything still in flight and returns the list of gathered buckets, each [world, steps_per_bucket, B]
    (buckets recycled in the meantime are kept as copies unless ``keep=False``).
    """

    def __init__(self, npoints, dtype, device, steps_per_bucket=4, group=None, force_collective=False, keep=True):
        import torch
        self.keep = bool(keep)   # keep a copy of a bucket's gathered results when its buffers are recycled before results() was called
        self.ready = []
        self.k = int(steps_per_bucket)
        self.npoints = int(npoints)
        self.pipe = PipelinedAllGather((self.k * self.npoints,), dtype, device, nslots=2, group=group, force_collective=force_collective)
        self.buckets = [torch.zeros(self.k * self.npoints, dtype=dtype, device=device) for _ in range(2)]
        self.current, self.filled = 0, 0

    def slot(self):
        if self.filled == 0 and self.pipe.pending(self.current):
            gathered = self.pipe.result(self.current)   # the stream waits for the collective that last used this bucket's buffers
            if self.keep: self.ready.append(gathered.reshape(self.pipe.world, self.k, self.npoints).clone())
        return self.buckets[self.current][self.filled * self.npoints:(self.filled + 1) * self.npoints]

    def advance(self):
        self.filled += 1
        if self.filled == self.k:
            self.flush()

    def flush(self):
        if self.filled:
            self.pipe.submit(self.current, self.buckets[self.current])
            self.current, self.filled = 1 - self.current, 0

    def results(self):
        self.flush()
        out, self.ready = self.ready, []
        for slot in (self.current, 1 - self.current):
            if self.pipe.pending(slot):
                out.append(self.pipe.result(slot).reshape(self.pipe.world, self.k, self.npoints))
        return out
