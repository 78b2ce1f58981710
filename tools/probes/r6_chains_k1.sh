cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from bench import make_likelihood_config5
from desilike_amd.samplers import EmceeSampler
from desilike_amd.parallel import WalkerSharding
like = make_likelihood_config5(0)
for K in (1, 2):
    sampler = EmceeSampler(like, nwalkers=512, chains=K, seed=42, sharding=WalkerSharding(group=False), device_resident=True)
    times = []
    for rep in range(12):
        t0 = time.perf_counter(); sampler.run(niterations=300); torch.cuda.synchronize(); times.append(1e3 * (time.perf_counter() - t0))
    print('K = %d: ms per batch of 300 updates: %s' % (K, ' '.join('%.1f' % t for t in times)))
    for runner in sampler._runners.values(): runner.ens.close()
PY
