"""GPU (-m gpu): MLP emulator training on the device (row f2; dl_mlp_*, csrc/dl_mlp.hip) against the oracle's restatement (gradient, Adam steps) and end to end:
an MLP trained on ``dl_eval_theory`` outputs reproduces the theory multipoles, and a likelihood built on it tracks the direct likelihood."""
import numpy as np
import pytest

from oracle import np_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('activation', ['silu', 'tanh', 'relu'])
def test_loss_gradient_and_adam_steps_vs_oracle(activation):
    import torch
    from desilike_amd._lib import MLPTrainer
    rng = np.random.RandomState(3)
    widths = [5, 33, 17, 70]          # ragged on purpose: no dimension is a multiple of the 16 x 16 x 4 MFMA tile
    layers = [(rng.standard_normal((a, b)) / a**0.5, 0.05 * rng.standard_normal(b)) for a, b in zip(widths[:-1], widths[1:])]
    S = 203
    x, y = rng.uniform(0., 1., (S, 5)), rng.standard_normal((S, 70))
    xt, yt = torch.as_tensor(x, device='cuda:0'), torch.as_tensor(y, device='cuda:0')
    trainer = MLPTrainer(layers, activation=activation, device=0)
    loss, grad = trainer.loss_and_grad(xt, yt)
    ref_loss, ref_grads = orc.mlp_loss_and_grad(layers, x, y, activation)
    ref_flat = np.concatenate([np.concatenate([gk.ravel(), gb.ravel()]) for gk, gb in ref_grads])
    assert abs(loss - ref_loss) <= 1e-13 * abs(ref_loss)
    assert np.allclose(grad, ref_flat, rtol=1e-11, atol=1e-14 * np.abs(ref_flat).max())
    out = trainer.forward(xt).cpu().numpy()
    a = x
    for il, (kernel, bias) in enumerate(layers):
        a = a.dot(kernel) + bias
        if il < 2: a = orc._mlp_act(a, activation)
    assert np.allclose(out, a, rtol=1e-12, atol=1e-13)
    # 25 Adam steps on chunks of 50 samples (4 chunks, the remainder of 3 samples unused): same weights as the oracle's Adam to rounding
    losses = trainer.train(xt, yt, batch=50, nsteps=25, lr=3e-3)
    ref_layers, ref_losses = orc.mlp_adam(layers, x, y, batch=50, nsteps=25, lr=3e-3, activation=activation)
    assert np.allclose(losses, ref_losses, rtol=1e-9, atol=0.)
    for (kernel, bias), (rk, rb) in zip(trainer.layers(), ref_layers):
        assert np.allclose(kernel, rk, rtol=1e-8, atol=1e-10) and np.allclose(bias, rb, rtol=1e-8, atol=1e-10)
    # determinism: a second trainer from the same start gives the same bits
    again = MLPTrainer(layers, activation=activation, device=0)
    assert np.array_equal(again.train(xt, yt, batch=50, nsteps=25, lr=3e-3), losses)
    trainer.close(); again.close()


def test_mlp_emulator_of_the_theory_trained_on_the_gpu():
    """emulate_power(engine='mlp'): samples from the R_d sequence, theory by dl_eval_theory (resident), Adam on the device; the emulated multipoles match the direct theory
    and the likelihood on the emulator tracks the direct likelihood."""
    from test_host_api import make_cfg2
    from desilike_amd.emulators import emulate_power
    from desilike_amd.theories.galaxy_clustering import EmulatedTracerPowerSpectrumMultipoles
    from desilike_amd.observables.galaxy_clustering import TracerPowerSpectrumMultipolesObservable
    from desilike_amd.likelihoods import ObservablesGaussianLikelihood
    g, like = make_cfg2(dense=False)
    pt = emulate_power(like, engine='mlp', nsamples=4096, delta_scale=0.5, hidden=(64, 64, 64), nsteps=20000, batch=1024, lr=3e-3, lr_decay=0.05, seed=1)   # ~18 s on an MI355X
    engine = pt.engines['power']
    assert engine.layers[0][0].shape == (6, 64) and engine.layers[-1][0].shape == (64, 3 * 400)
    names = like.varied_params.names()
    rng = np.random.RandomState(7)
    center = np.array([param.value for param in like.varied_params])
    half = 0.4 * np.array([param.proposal for param in like.varied_params])
    theta = center + rng.uniform(-1., 1., size=(64, len(names))) * half
    direct = like._get_context().eval_theory_host(theta, iobs=0)
    emulated = np.array([orc.mlp_predict(row, engine.xlimits, engine.layers, 'silu', engine.ylimits).reshape(3, -1) for row in theta])
    scale = np.abs(direct[:, 0]).max(axis=-1)[:, None, None]          # monopole amplitude of each point
    assert np.abs(emulated - direct).max() <= 4e-3 * scale.max(), np.abs(emulated - direct).max() / scale.max()     # measured 1.6e-3 (7e-3 after 4000 steps, 8e-4 after 40000)
    # the emulated theory through the device path (emulator forward + folded operator) = the same numbers
    theory = EmulatedTracerPowerSpectrumMultipoles(pt=pt)
    obs = TracerPowerSpectrumMultipolesObservable(data=g['obs0']['flatdata'], kedges=np.linspace(0., 0.2, 41), ells=(0, 2, 4), wmatrix={'resolution': 10}, theory=theory, shotnoise=1e4)
    like_emu = ObservablesGaussianLikelihood(observables=[obs], covariance=g['covariance'])
    assert like_emu.varied_params.names() == names
    ll_emu = like_emu._get_context().eval_batch_host(theta)[0]
    ll_dir = like._get_context().eval_batch_host(theta)[0]
    assert np.isfinite(ll_emu).all() and np.corrcoef(ll_emu, ll_dir)[0, 1] > 0.99
