"""Load golden fixtures (tests/golden/*.npz, written by tests/golden/make_golden.py from the reference)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    data = np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False)
    out = {}
    for key in data.files:
        value = data[key]
        if value.dtype.kind in 'US' and value.ndim == 0:
            value = str(value)
        elif value.ndim == 0:
            value = value[()]
        if '.' in key:
            group, sub = key.split('.', 1)
            out.setdefault(group, {})[sub] = value
        else:
            out[key] = value
    return out


def observable_constants(g, iobs=0):
    """Oracle-side constants dict of observable ``iobs`` from a golden fixture."""
    c = dict(g['obs{:d}'.format(iobs)])
    c['template'] = {'ShapeFitPowerSpectrumTemplate': 'shapefit'}.get(c['template'], 'fixed')
    c['ellsin'] = tuple(int(ell) for ell in c['ellsin'])
    c['ells'] = tuple(int(ell) for ell in c['ells'])
    return c


def prior_list(g):
    return [dict(dist=['uniform', 'norm'][int(row[0])], limits=(row[1], row[2]), loc=row[3], scale=row[4]) for row in g['priors']]
