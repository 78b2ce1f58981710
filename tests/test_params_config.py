"""``calculator.all_params = config`` / ``likelihood.all_params = config`` (desilike/base.py:1302-1310, 436-470; parameter.py:1472-1547, 1588-1620): the walk through
the parameter API of the reference's own test (desilike/tests/test_base.py:139-165), on the host mirror (no GPU needed: parameters only)."""
import numpy as np
import pytest

from test_host_api import make_cfg2


def test_all_params_configuration(tmp_path):
    from desilike_amd import ParameterCollection
    from desilike_amd.parameter import ParameterError
    g, like = make_cfg2(dense=True)
    theory = like.init['observables'][0].init['theory']
    # calculator level: a live view of template + own parameters
    assert theory.all_params.names() == ['dm', 'dn', 'qpar', 'qper', 'df', 'b1', 'sn0', 'sigmapar', 'sigmaper']
    theory.all_params['b1'].update(prior={'dist': 'norm', 'loc': 0., 'scale': 1.})
    assert theory.params['b1'].prior.dist == 'norm'
    theory.all_params = {'q*': {'prior': {'limits': [0.9, 1.1]}}, 'sn0': {'prior': {'dist': 'norm', 'loc': 0., 'scale': 1e4}}, '.fixed': 'df'}
    assert theory.template.params['qpar'].prior.limits == (0.9, 1.1) and theory.template.params['qper'].prior.limits == (0.9, 1.1)
    assert theory.params['sn0'].prior.scale == 1e4 and theory.template.params['df'].fixed
    theory.all_params = {'.varied': ['df']}
    assert theory.template.params['df'].varied
    # likelihood level: YAML file, item update, patterns (test_base.py:140-145)
    fn = tmp_path / 'test_params.yaml'
    fn.write_text("dm:\n  prior:\n    dist: norm\n    loc: 0.\n    scale: 2.\nqpar: 1.02\n")
    like.all_params = str(fn)
    assert like.varied_params['dm'].prior.scale == 2. and like.all_params['qpar'].value == 1.02
    like.all_params['dm'].update(prior={'dist': 'norm', 'loc': 0., 'scale': 100.})
    assert like.varied_params['dm'].prior.scale == 100.
    like.all_params = {'*': {'prior': {'dist': 'norm', 'loc': 0., 'scale': 1.}}}
    assert like.varied_params['dm'].prior.scale == 1. and like.all_params['sigmapar'].prior.scale == 1.
    # an explicit entry wins over a pattern of the same configuration; meta entries
    like.all_params = {'*': {'prior': {'dist': 'norm', 'loc': 0., 'scale': 3.}}, 'dm': {'prior': {'limits': [-1., 1.]}}, '.fixed': ['qp*']}
    assert like.all_params['dm'].prior.dist == 'uniform' and like.all_params['b1'].prior.scale == 3. and like.all_params['qpar'].fixed and like.all_params['qper'].fixed
    like.all_params = {'.varied': 'qp*', 'sn0': {'derived': '.marg'}}
    assert like.all_params['qper'].varied and like.all_params['sn0'].solved and 'sn0' not in like.varied_params
    # a ParameterCollection as configuration
    like.all_params = ParameterCollection({'b1': dict(value=1.7, prior=dict(limits=[0., 4.]))})
    assert like.all_params['b1'].value == 1.7 and like.all_params['b1'].prior.limits == (0., 4.)
    # new parameters only where others are derived from them (test_base.py:161-162)
    with pytest.raises(ParameterError):
        like.all_params = {'b': {'prior': {'limits': [0., 2.]}}}
    like.all_params = {'b1': {'derived': '{b}**2', 'prior': None}, 'b': {'prior': {'limits': [0., 2.]}}}
    assert 'b' in like.varied_params and like.dependent_params.names() == ['b1']
    like.all_params = {'.delete': 'b', 'b1': {'derived': False, 'prior': {'limits': [0., 4.]}}}
    assert 'b' not in like.all_params and 'b1' in like.varied_params
    spec = like._spec({}, like._flatdata_list(), like.precision)     # still compiles to a context specification
    assert int(spec['n_params'][0]) == len(like.varied_params)
