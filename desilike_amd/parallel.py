"""Walker / parameter-point sharding across the GPUs of a node.

The reference's only parallelism is data-parallel over points (``vmap(..., backend='mpi')`` desilike/base.py:291-378:
``Scatterv`` of the points, local loop, gather; ``mpi.local_size`` desilike/mpi.py:145-149).  Here: one process per GPU,
rank r evaluates the contiguous slice ``[r B // G, (r + 1) B // G)`` and the only exchange is ONE all-gather of the per-point
results (log-posteriors: B / G doubles per rank) through ``torch.distributed`` -- backend "nccl" (= RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.  Payloads are kilobytes: the exchange is latency-bound, so it is never split or bucketed.
"""
import numpy as np


def local_slice(size, rank, world):
    """Contiguous share of ``size`` items for ``rank`` out of ``world`` (same rule as desilike/mpi.py:145-149)."""
    return slice(rank * size // world, (rank + 1) * size // world)


class WalkerSharding(object):
    """Evaluate a batch function on the local share of the rows and all-gather the results on every rank."""

    def __init__(self, group=None, device=None, min_shard_rows=4096):
        """``min_shard_rows``: below this many rows every rank evaluates ALL rows and nothing is exchanged.  The evaluation is latency-bound at small batches
        (34 us for 32 as for 256 points of the two-tracer likelihood) while a synchronous all-gather through the host costs ~100 us: sharding 512 walkers over 8 GPUs is
        slower than evaluating them redundantly (the kernels are deterministic: every rank gets the same bits); the break-even is a few thousand points.  Several
        GPUs then serve independent chains, as the reference's ``chains=N`` does over MPI communicators (utils.py:1090).  0: always shard."""
        import torch.distributed as dist
        self.min_shard_rows = int(min_shard_rows)
        self.dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.active else 0
        self.world = dist.get_world_size(group) if self.active else 1
        self.device = device

    def slice(self, size):
        return local_slice(size, self.rank, self.world)

    def allgather_rows(self, local, size):
        """``local``: array / tensor with the rows of this rank's slice of ``size`` rows -> all ``size`` rows, on every rank."""
        import torch
        if not self.active or self.world == 1:
            return local
        is_numpy = isinstance(local, np.ndarray)
        tensor = torch.as_tensor(local) if is_numpy else local
        if self.device is not None:
            tensor = tensor.to(self.device)
        counts = [local_slice(size, rank, self.world) for rank in range(self.world)]
        counts = [sl.stop - sl.start for sl in counts]
        nmax = max(counts)
        trailing = tuple(tensor.shape[1:])
        padded = torch.zeros((nmax,) + trailing, dtype=tensor.dtype, device=tensor.device)
        padded[:tensor.shape[0]] = tensor
        gathered = torch.empty((self.world * nmax,) + trailing, dtype=tensor.dtype, device=tensor.device)
        self.dist.all_gather_into_tensor(gathered, padded, group=self.group)   # the single collective of the path
        gathered = gathered.reshape((self.world, nmax) + trailing)
        out = torch.cat([gathered[rank, :count] for rank, count in enumerate(counts)], dim=0)
        return out.cpu().numpy() if is_numpy else out

    def map(self, func, values):
        """``func(values_local) -> array[len(values_local), ...]`` applied to this rank's slice; returns the full result everywhere."""
        values = np.asarray(values)
        if not self.active or self.world == 1 or len(values) < self.min_shard_rows:
            return np.asarray(func(values))
        sl = self.slice(len(values))
        local = np.asarray(func(values[sl]))
        return self.allgather_rows(np.ascontiguousarray(local), len(values))

    def broadcast(self, array, src=0):
        """Broadcast a numpy array from ``src`` (walker positions must be identical on all ranks: samplers/base.py:45-54)."""
        import torch
        if not self.active or self.world == 1:
            return array
        tensor = torch.as_tensor(np.ascontiguousarray(array))
        if self.device is not None:
            tensor = tensor.to(self.device)
        self.dist.broadcast(tensor, src=src, group=self.group)
        return tensor.cpu().numpy()


class PipelinedAllGather(object):
    """Double-buffered asynchronous all-gather of per-point results, for drivers that keep several independent walker ensembles (chains) in flight.

    ``submit(slot, local)`` starts the all-gather of ensemble ``slot``'s local results on the collective's own stream and returns at once: the caller goes
    on enqueueing the evaluation of the *other* ensemble, whose kernels overlap the (latency-bound, kilobyte-sized) collective.  ``result(slot)`` makes the
    current stream wait for that collective and returns the gathered rows.  One collective per ensemble step, never split (xGMI is point-to-point:
    a small all-gather costs a ring latency whatever its size).  A slot must be drained (``result``) before it is submitted again.
    """

    def __init__(self, shape, dtype, device, nslots=2, group=None, force_collective=False):
        import torch
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.collective = self.active and (self.world > 1 or force_collective)   # (forcing: exercises the backend with a single rank)
        shape = tuple(shape)
        self.gathered = [torch.empty((self.world * shape[0],) + shape[1:], dtype=dtype, device=device) for _ in range(nslots)]
        self.work = [None] * nslots
        self.local = [None] * nslots

    def submit(self, slot, local):
        if self.work[slot] is not None:
            raise RuntimeError('slot {:d} resubmitted before its result was taken'.format(slot))
        if not self.collective:
            self.local[slot] = local
            self.work[slot] = True
            return
        self.local[slot] = local   # keep the input alive until the collective has consumed it
        self.work[slot] = self.dist.all_gather_into_tensor(self.gathered[slot], local, group=self.group, async_op=True)

    def pending(self, slot):
        return self.work[slot] is not None

    def result(self, slot):
        work = self.work[slot]
        if work is None:
            raise RuntimeError('nothing submitted in slot {:d}'.format(slot))
        self.work[slot] = None
        if not self.collective:
            return self.local[slot]
        work.wait()   # stream-ordered on GPUs (the current stream waits for the collective), blocking on CPU backends
        return self.gathered[slot]


class BucketedAllGather(object):
    """All-gather of per-point results for drivers with many independent walker ensembles (chains) in flight: the results of ``steps_per_bucket`` consecutive
    ensemble steps travel in ONE asynchronous collective (double-buffered buckets on top of :class:`PipelinedAllGather`).

    A kilobyte-sized all-gather costs a fixed ~25 us of host-side issue time plus a ring latency whatever its payload (measured with a single-rank RCCL group:
    +26 us per 30 us step when issued every step), so small exchanges are bucketed, exactly like gradient buckets in data-parallel training.
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
