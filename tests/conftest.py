import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir(os.environ.get('ARFLOW_REFERENCE', '/root/reference'))
    have_gpu = torch.cuda.is_available()
    for item in items:
        if 'reference' in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason='reference not mounted'))
        if 'gpu' in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason='no GPU visible'))


class Golden:
    """Lazy npz reader returning torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

    def __contains__(self, k):
        return k in self._z.files

    def raw(self, k):
        return self._z[k]

    def __getitem__(self, k):
        a = self._z[k]
        if a.dtype.kind in 'US':
            return [str(s) for s in a.tolist()] if a.ndim else str(a)
        return torch.from_numpy(np.array(a, copy=True))

    def names(self):
        return self['names']


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def assert_close(actual, expected, atol, rtol, msg=''):
    actual = actual.detach().cpu().double()
    expected = expected.detach().cpu().double()
    assert actual.shape == expected.shape, '%s shape %s vs %s' % (msg, tuple(actual.shape), tuple(expected.shape))
    err = (actual - expected).abs()
    tol = atol + rtol * expected.abs()
    # `~(err <= tol)` and not `err > tol`: a NaN/Inf in `actual` compares False either way round, and must count
    # as a mismatch (tests/test_bench_cpu.py::test_assert_close_rejects_nan pins this)
    bad = ~(err <= tol)
    if bad.any():
        i = int(torch.argmax(torch.where(bad, torch.nan_to_num(err - tol, nan=float('inf'), posinf=float('inf')),
                                         torch.full_like(err, -1.0))))
        raise AssertionError('%s: %d/%d elements out of tolerance (atol=%g rtol=%g); worst |err|=%.3e at flat %d '
                             '(actual %.8g expected %.8g)' % (msg, int(bad.sum()), bad.numel(), atol, rtol,
                                                             float(err.flatten()[i]), i,
                                                             float(actual.flatten()[i]), float(expected.flatten()[i])))
