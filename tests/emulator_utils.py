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
