"""Timing of BASELINE configs[2] on the emulator layout the reference ships (stacked engines, emulators/conversion.py:44-98): dl_stk_chain_kernel + dl_emulated_stacked_gemm_kernel
(DL_NO_STK_SPLIT=1: dl_emulated_stacked_kernel, one launch).
  python tools/time_stacked.py [B] [marg 0|1] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch


def main():
    from bench_configs import make_cfg3_stacked
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    marg = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    like, pt, theory, solved, networks = make_cfg3_stacked(marg=marg)
    ctx = like._get_context()
    rng = np.random.RandomState(3)
    theta = np.column_stack([np.clip(param.ref.sample(size=B, random_state=rng), *param.prior.limits) for param in like.varied_params])
    th = torch.as_tensor(theta, dtype=torch.float64, device='cuda').contiguous()
    out = torch.empty(B, dtype=torch.float64, device='cuda')
    st = torch.empty(B, dtype=torch.int32, device='cuda')
    for _ in range(250): ctx.eval_logposterior(th, out, status=st)     # (the first ~200 calls of a fresh process are slower: tools/time_gap_stacked.py)
    torch.cuda.synchronize()
    ctx.profile_enable(2)
    t0 = time.perf_counter()
    for _ in range(steps): ctx.eval_logposterior(th, out, status=st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ms = ctx.profile_read(); ctx.profile_enable(0)
    print('stacked cfg3 marg=%d B=%5d  %8.1f us/step  %7.2f M evals/s  kernels (us): network chains %.1f feature GEMMs %.1f finalize %.1f  [%d ok]' % (
        marg, B, 1e6 * dt, B / dt / 1e6, *(1e3 * ms[k] for k in ['theory', 'window_gemm', 'finalize']), int((st == 0).sum().item())))


if __name__ == '__main__':
    main()
