"""CPU: full-shape correlation function multipoles -- (EFT-like) Kaiser P_ell -> xi_ell through get_corr (full_shape.py:336-364, 553-574, 664-687;
tgc/base.py:46-139) -- oracle restatement against fixtures captured from the reference (tests/golden/make_golden.py kaiser_xi; the Hankel step of the
reference is third-party: the fixtures ran on the oracle's FFTLog, "parity unpinned" for that step, see oracle/np_oracle.py)."""
import numpy as np
import pytest

from oracle import np_oracle as orc
from golden_utils import load_golden, prior_list


def kaiser_xi_point(g, row, interp_order=1):
    c = dict(g['obs0'])
    names = [str(n) for n in g['names']]
    p = dict(zip(names, row))
    p['b1'] = (p['b1'], p['b1'])
    c['template'] = 'shapefit'
    c['flatdata'] = np.zeros(len(c['ellsin']) * len(c['kin']))
    if 'ct_params' in c:
        p['ct'] = [2. * p.get(str(n), 0.) for n in c['ct_params']]   # auto-correlation: sum over the two (identical) tracers
        p['sn'] = [0.] * len(c['sn_params'])
    power = orc.fullshape_observable(c, p)['power']
    return power, orc.get_corr(power, c['kin'], c['s'], tuple(int(ell) for ell in c['ells']), interp_order=interp_order)


@pytest.mark.parametrize('name', ['kaiser_xi', 'kaiser_xi_eft', 'kaiser_xi_cubic'])
def test_kaiser_xi_chain_vs_reference(name):
    g = load_golden(name)
    priors = prior_list(g)
    assert len(g['obs0']['kin']) == (100 if name.endswith('cubic') else 300)     # tgc/base.py:66
    for i, row in enumerate(g['theta']):
        power, corr = kaiser_xi_point(g, row, interp_order=3 if name.endswith('cubic') else 1)
        assert np.allclose(power, g['power'][i], rtol=1e-11, atol=1e-12 * np.abs(g['power'][i]).max())
        assert np.allclose(corr, g['theory'][i], rtol=1e-10, atol=1e-12 * np.abs(g['theory'][i]).max())
        logl = orc.gaussian_loglikelihood(np.ravel(corr), g['obs0']['flatdata'], g['precision'])[0]
        assert abs(logl - g['loglikelihood'][i]) <= 1e-10 * max(1., abs(g['loglikelihood'][i]))
        assert np.isclose(orc.logprior(row, priors), g['logprior'][i], rtol=1e-13, atol=1e-13)
