"""What one bucketed all-gather costs the evaluation stream (single-rank RCCL communicator): the real collective on its side stream, the same events with the
collective replaced by a device-to-device copy, and the events alone.  Probe: docs/EXPERIMENTS.md."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from desilike_amd.parallel import RcclGroup, BucketedAllGather

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29511')
device = torch.device('cuda', 0)
group = bench.init_rccl_group(sys.modules['desilike_amd.parallel'], 0, 0, 1)
like = bench.make_likelihood(0)
ctx = like._get_context()
B, K = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 8
ONLY = sys.argv[2] if len(sys.argv) > 2 else None   # one variant per process: the variants leave streams and events behind
theta = torch.as_tensor(bench.sample_theta(like, B, seed=42), dtype=torch.float64, device=device).contiguous()
status = torch.zeros(B, dtype=torch.int32, device=device)
real = group.allgather_into


def copy_only(recv, send, stream=None):
    with torch.cuda.stream(torch.cuda.ExternalStream(stream)): recv.copy_(send)


for name, fn in (('no exchange', None), ('RCCL all-gather', real), ('copy on the side stream', copy_only), ('events only', lambda recv, send, stream=None: None), ('RCCL all-gather', real)):
    if ONLY is not None and not name.startswith(ONLY): continue
    bucket = None
    if fn is not None:
        group.allgather_into = fn
        bucket = BucketedAllGather(B, torch.float64, device, steps_per_bucket=K, keep=False, group=group, force_collective=True)
    out = torch.empty(B, dtype=torch.float64, device=device)

    def run(n):
        for _ in range(n):
            if bucket is None: ctx.eval_logposterior(theta, out, status=status)
            else:
                ctx.eval_logposterior(theta, bucket.slot(), status=status)
                bucket.advance()
        if bucket is not None: bucket.results()
        torch.cuda.synchronize()

    run(12000)   # (clocks and the runtime's lazy set-up: the first few thousand steps of a process are slow)
    t0 = time.perf_counter(); run(2000); dt = (time.perf_counter() - t0) / 2000
    print('%-24s buckets of %2d: %.2f us per step' % (name, K, 1e6 * dt))
