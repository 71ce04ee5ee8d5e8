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


# Achieved parity margins: every assert_close() call records max(err / tol) under the running test's
# id; a GPU session leaves them in gpurun_out/parity_margins.json (summary committed as profiles/r03_parity_margins.json),
# so that a tolerance can be read against the error the kernels actually achieve.
MARGINS = []


def _record_margin(msg, err, tol, atol, rtol):
    test = os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]
    import inspect
    fr = inspect.stack()[2]  # the test's assert_close() call site
    site = '%s:%d' % (os.path.basename(fr.filename), fr.lineno)
    finite = torch.isfinite(err)
    if not bool(finite.all()) or err.numel() == 0:
        return
    ratio = torch.where(tol > 0, err / tol.clamp_min(1e-300), torch.where(err > 0, torch.full_like(err, float('inf')), torch.zeros_like(err)))
    MARGINS.append({'test': test, 'site': site, 'what': msg, 'max_err': float(err.max()), 'atol': atol, 'rtol': rtol,
                    'err_over_tol': float(ratio.max())})


def pytest_sessionfinish(session, exitstatus):
    if not MARGINS or not torch.cuda.is_available():
        return
    import json
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, 'parity_margins.json')
    old = []
    if os.path.exists(path) and os.environ.get('ARFLOW_MARGINS_APPEND') == '1':
        old = json.load(open(path))
    json.dump(old + MARGINS, open(path, 'w'), indent=0)


def assert_close(actual, expected, atol, rtol, msg=''):
    actual = actual.detach().cpu().double()
    expected = expected.detach().cpu().double()
    assert actual.shape == expected.shape, '%s shape %s vs %s' % (msg, tuple(actual.shape), tuple(expected.shape))
    err = (actual - expected).abs()
    tol = atol + rtol * expected.abs()
    _record_margin(msg, err, tol, atol, rtol)
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
