"""Phase durations of dl_fullshape_kernel (DL_FS_STAMPS) with one workgroup per CU (256 points) and with four (1024 points): how much of a point's 8 us chain is
contention between the co-resident workgroups, which all start together and meet in the same phase.
    python tools/fs_contention.py <B> <stamp file>      (one process per B: the launch counter of the diagnostics is static)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B, path = int(sys.argv[1]), sys.argv[2]
os.environ['DL_FS_STAMPS'] = path
import torch
import bench
like = bench.make_likelihood(0)
theta = torch.as_tensor(bench.sample_theta(like, B, seed=42), dtype=torch.float64, device='cuda').contiguous()
out = torch.empty(B, dtype=torch.float64, device='cuda')
ctx = like._get_context()
for _ in range(40): ctx.eval_logposterior(theta, out)
torch.cuda.synchronize()
