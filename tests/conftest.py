import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + '.npz')))
    return load


@pytest.fixture(scope='session')
def golden_full_size_inputs():
    """The seeded full-size draw of fixture G9 (tests/golden/make_golden.py: only its input generator is imported -- it needs neither the
    reference nor a GPU), checked against the fp64 checksums the fixture stores."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location('_afx_make_golden', os.path.join(GOLDEN, 'make_golden.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    inp = mod.golden_full_size_inputs()
    g = dict(np.load(os.path.join(GOLDEN, 'g9_step_full_size.npz')))
    for k, v in inp.items():
        assert abs(v.double().sum().item() - float(g['in_sum_' + k])) <= 1e-6 * max(1.0, abs(float(g['in_sum_' + k]))), \
            f'the seeded draw of {k} differs from the one fixture G9 was generated with'
    return inp, g
