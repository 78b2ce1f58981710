import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from oracle import np_oracle as orc
from bench_configs import make_cfg2
from desilike_amd.emulators import emulate_power
g, like = make_cfg2(dense=False)
names = like.varied_params.names()
rng = np.random.RandomState(7)
center = np.array([param.value for param in like.varied_params])
half = 0.4 * np.array([param.proposal for param in like.varied_params])
theta = center + rng.uniform(-1., 1., size=(64, len(names))) * half
direct = like._get_context().eval_theory_host(theta, iobs=0)
scale = np.abs(direct[:, 0]).max()
for nsteps, lr, decay, batch, ns in [(4000, 3e-3, 0.2, 1024, 4096), (20000, 3e-3, 0.05, 1024, 4096), (20000, 5e-3, 0.02, 2048, 8192), (40000, 3e-3, 0.02, 1024, 8192)]:
    t0 = time.time()
    pt = emulate_power(like, engine='mlp', nsamples=ns, delta_scale=0.5, hidden=(64, 64, 64), nsteps=nsteps, batch=batch, lr=lr, lr_decay=decay, seed=1)
    dt = time.time() - t0
    engine = pt.engines['power']
    emulated = np.array([orc.mlp_predict(row, engine.xlimits, engine.layers, 'silu', engine.ylimits).reshape(3, -1) for row in theta])
    print(nsteps, lr, decay, batch, ns, 'max err / scale = %.2e' % (np.abs(emulated - direct).max() / scale), 'time %.1f s' % dt, flush=True)
