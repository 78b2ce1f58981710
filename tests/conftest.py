import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    # the tests build templates on the synthetic stand-in cosmology deliberately (tests/test_host_api.py::test_default_fiducial_warns checks the warning itself)
    config.addinivalue_line('filterwarnings', 'ignore::desilike_amd.fiducial.FiducialWarning')
