"""GPU (-m gpu): the end-to-end example runs and recovers the parameters its mock data were generated with."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'examples'))


def test_full_shape_fit(tmp_path):
    import full_shape_fit
    out = full_shape_fit.main(str(tmp_path), quick=True)
    for name, value in out['truth'].items():
        if name in out['mean']:
            assert abs(out['best'][name] - value) < 1e-3 * max(1., abs(value)), (name, out['best'][name], value)      # mock data = theory: the maximum is the truth
            assert abs(out['mean'][name] - value) < 1.5 * out['std'][name]
    assert out['eigen_gr'] < 0.3 and os.path.exists(os.path.join(str(tmp_path), 'chain_0.npy'))
