"""Shared helpers for the emulator / velocileptors-table tests: the stand-in PT node of fixture cfg3_velocileptors_table as an exact
second-order Taylor emulator (layout of emulators/__init__.py:471-507)."""
import numpy as np

POWERS = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [0, 2, 0]])     # qpar - 1, qper - 1, dm
CENTER = np.array([1., 1., 0.])
EMU_PARAMS = ['qpar', 'qper', 'dm']


def taylor_state(g):
    """Engines of the three emulated arrays of the PT node: pktable [n_ell, n_kpt, 19], sigma8, fsigma8."""
    tables = g['obs0']['tables']
    sigma8 = np.zeros(7); powers8 = np.vstack([POWERS, [[2, 0, 0]]])
    sigma8[0], sigma8[3], sigma8[6] = 0.8, 0.8 * 0.2, 0.8 * 0.1          # 0.8 (1 + 0.2 dm + 0.1 (qpar - 1)^2)
    fsigma8 = np.zeros(6)
    fsigma8[0], fsigma8[2], fsigma8[3] = 0.45, 0.45 * 0.3, -0.45 * 0.1    # 0.45 (1 + 0.3 (qper - 1) - 0.1 dm)
    return {'pktable': dict(center=CENTER, powers=POWERS, derivatives=tables), 'sigma8': dict(center=CENTER, powers=powers8, derivatives=sigma8),
            'fsigma8': dict(center=CENTER, powers=POWERS, derivatives=fsigma8)}


# ---- BASELINE configs[2] at the size SURVEY.md section 8d states: MLP in = 6 -> 4 x 64 silu -> 3 * 128 * 19 = 7296 outputs, n_kin = 400, W 120 x 1200 ----------
CFG3_PARAMS = ['qpar', 'qper', 'dm', 'df', 'dn', 'lnA']        # qpar, qper, dm, df + 2 spare inputs
CFG3_XLIMITS = np.array([[0.9, 1.1], [0.9, 1.1], [-0.1, 0.1], [0.8, 1.2], [-0.1, 0.1], [-0.2, 0.2]])
CFG3_SPECS = {'qpar': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])), 'qper': dict(value=1., prior=dict(limits=[0.8, 1.2]), ref=dict(limits=[0.98, 1.02])),
              'dm': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05])), 'df': dict(value=1., prior=dict(limits=[0., 2.]), ref=dict(limits=[0.95, 1.05])),
              'dn': dict(value=0., prior=dict(limits=[-0.5, 0.5]), ref=dict(limits=[-0.02, 0.02])), 'lnA': dict(value=0., prior=dict(limits=[-1., 1.]), ref=dict(limits=[-0.05, 0.05]))}


def cfg3_full_kpt():
    """128 wavenumbers of the emulated perturbation-theory tables."""
    return np.concatenate([[0.0005], np.geomspace(0.0015, 0.025, 27), np.arange(0.03, 1.025, 0.01)])[:128]


def cfg3_full_engines(seed=1):
    """Weights ~ N(0, 1 / fan_in) from ``RandomState(seed)`` (nothing physical is trained here), min-max scalers as emulators/conversion.py:75-79.
    Returns {'pktable', 'sigma8', 'fsigma8'}: dict(xlimits, layers, ylimits) -- the same numbers feed the reference-side stand-in node (make_golden.cfg3_full)
    and the device engines (tests/test_gpu_emulator.py)."""
    rng = np.random.RandomState(seed)
    kpt = cfg3_full_kpt()
    nk = kpt.size
    base = 2e4 * (kpt / 0.05)**0.96 / (1. + (kpt / 0.02)**2.5)

    def mlp(nout, widths):
        layers, last = [], len(CFG3_PARAMS)
        for width in widths + [nout]:
            layers.append((rng.standard_normal((last, width)) / last**0.5, 0.1 * rng.standard_normal(width)))
            last = width
        return layers

    amp = np.concatenate([[1.], 0.2 * np.ones(11), 0.05 * np.ones(4), [0., 0., 0.]])
    ylim = np.stack([-(base[None, :, None] * amp) * np.ones((3, 1, 1)), (base[None, :, None] * amp) * np.ones((3, 1, 1))], axis=-1)     # [3, nk, 19, 2]
    for ill in range(3): ylim[ill, :, 16 + ill, :] = (kpt**(2 * ill))[:, None]      # stochastic monomials are exact constants: zero output range, offset lo = hi
    engines = {'pktable': dict(xlimits=CFG3_XLIMITS, layers=mlp(3 * nk * 19, [64, 64, 64, 64]), ylimits=ylim.reshape(-1, 2), yshape=(3, nk, 19))}
    engines['sigma8'] = dict(xlimits=CFG3_XLIMITS, layers=mlp(1, [16]), ylimits=np.array([[0.7, 0.9]]), yshape=(1,))
    engines['fsigma8'] = dict(xlimits=CFG3_XLIMITS, layers=mlp(1, [16]), ylimits=np.array([[0.4, 0.5]]), yshape=(1,))
    return engines
