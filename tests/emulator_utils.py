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


# ---- the jaxeffort layout (emulators/conversion.py:44-98): engines '11' / 'loop' / 'ct' / 'st', one network per (z, ell) stacked in each, amplitude rescale by the input logA ----------
STK_PARAMS = ['logA', 'n_s', 'h', 'omega_b', 'omega_cdm']                                     # conversion.py:15
STK_COMPONENTS = (('11', 3), ('loop', 9), ('ct', 4), ('st', 3))                               # monomial groups: jnp.split(pktable, [3, 12, 16], axis=2), conversion.py:50
STK_LIMITS = np.array([[2.5, 3.5], [0.9, 1.02], [55., 80.], [0.020, 0.024], [0.09, 0.15]])    # in_MinMax; the 'h' row in km / s / Mpc (conversion.py:73-74 divides it by 100)
STK_SPECS = {'logA': dict(value=3.04, prior=dict(limits=[2.5, 3.5]), ref=dict(limits=[3.02, 3.06])), 'n_s': dict(value=0.965, prior=dict(limits=[0.9, 1.02]), ref=dict(limits=[0.96, 0.97])),
             'h': dict(value=0.674, prior=dict(limits=[0.55, 0.8]), ref=dict(limits=[0.67, 0.68])), 'omega_b': dict(value=0.0224, prior=dict(dist='norm', loc=0.0224, scale=0.0004, limits=[0.020, 0.024]), ref=dict(limits=[0.0222, 0.0226])),
             'omega_cdm': dict(value=0.12, prior=dict(limits=[0.09, 0.15]), ref=dict(limits=[0.118, 0.122]))}


def stacked_kgrid(nk=30, kmax=0.32):
    """``emu.k_grid`` of the component emulators (conversion.py:69): wide enough for the tracer's cubic interpolation to its own wavenumbers (full_shape.py:1598)."""
    return np.concatenate([[0.0005], np.geomspace(0.0015, 0.025, nk // 3), np.linspace(0.03, kmax, nk - nk // 3 - 1)])


def stacked_networks(z, ells=(0, 2, 4), nk=30, hidden=(32, 32), activation='tanh', seed=3, kmax=0.32):
    """What ``jaxeffort.load_component_emulator`` hands to conversion.py:68-79 for every (component, iz, ell), as synthetic DATA (nothing physical is trained here):
    ``networks[component][iz][ill] = dict(k_grid, layers [(kernel [in, out], bias [out])], activations, in_MinMax [5, 2], out_MinMax [n_m * n_k, 2])``.
    The output ranges carry the amplitudes conversion.py:88-92 divides out: '11' and 'ct' by ``exp(logA) 1e-10``, 'loop' by its square."""
    rng = np.random.RandomState(seed)
    k = stacked_kgrid(nk, kmax=kmax)
    base = 2e4 * (k / 0.05)**0.96 / (1. + (k / 0.02)**2.5)
    a1 = np.exp(3.04) * 1e-10
    amplitude = {'11': 1. / a1, 'loop': 0.2 / a1**2, 'ct': 0.05 / a1, 'st': 1.}
    networks = {}
    for component, nm in STK_COMPONENTS:
        networks[component] = []
        for iz, zz in enumerate(z):
            growth = 1. / (1. + 0.4 * zz)
            row = []
            for ill, ell in enumerate(ells):
                layers, last = [], len(STK_PARAMS)
                for width in list(hidden) + [nm * nk]:
                    layers.append((rng.standard_normal((last, width)) / last**0.5, 0.1 * rng.standard_normal(width)))
                    last = width
                if component == 'st':   # stochastic tables: 1, k^2, k^4 on the monomial sn_{2 i} / nd, the same for every cosmology (zero output range, conversion.py:79: v * 0 + lo)
                    lo = np.array([[(0.3 + 0.1 * ill) * (k / 0.1)**(2 * i) * (1. if i >= ill else 0.) for i in range(nm)]]).reshape(nm, nk)
                    out = np.stack([lo, lo], axis=-1)
                else:
                    shape = base[None, :] * (1. + 0.3 * np.arange(nm)[:, None] / nm) * (k[None, :] / 0.1)**(0.5 * ill) * growth**2 * amplitude[component] * (0.5 if ell else 1.)
                    out = np.stack([-shape, 1.3 * shape], axis=-1)
                row.append(dict(k_grid=k, layers=layers, activations=[activation] * len(hidden), in_MinMax=STK_LIMITS.copy(), out_MinMax=out.reshape(nm * nk, 2)))
            networks[component].append(row)
    return networks
