"""CPU: the C-ABI library loads and exports every symbol include/arflow_hip.h declares (no compute
calls without a GPU), argument validation works, and the product ops refuse CPU tensors."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'arflow_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(arflow_[a-z0-9_]+)\s*\(', text)))


@pytest.fixture(scope='module')
def lib():
    from arflow_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from arflow_amd import _lib
    names = _declared()
    assert len(names) >= 17
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'symbol %s missing from libarflow_hip.so' % n
        if n not in ('arflow_strerror',):
            assert n in _lib.PROTOTYPES, 'no ctypes prototype for %s' % n
    assert set(_lib.PROTOTYPES) <= set(names)


def test_prototype_arity_matches_header():
    from arflow_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'arflow_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for name, argtypes in _lib.PROTOTYPES.items():
        m = re.search(r'\b%s\s*\(([^)]*)\)' % name, text)
        assert m, name
        params = [p for p in m.group(1).split(',') if p.strip() and p.strip() != 'void']
        assert len(params) == len(argtypes), '%s: header has %d params, binding %d' % (name, len(params), len(argtypes))


def test_argument_errors_without_gpu(lib):
    # validation happens before any launch, so these are safe on a CPU-only host
    assert lib.arflow_abi_version() == 10
    assert lib.arflow_corr_fwd(None, None, None, None, 1, 1, 1, 1, 4, 1.0, None) == -1001
    one = ctypes.c_void_p(16)
    assert lib.arflow_corr_fwd(one, one, one, None, 0, 1, 1, 1, 4, 1.0, None) == -1002
    assert lib.arflow_corr_fwd(one, one, one, None, 1, 1, 1, 1, 0, 1.0, None) == -1003
    assert lib.arflow_corr_bwd(one, None, None, one, one, one, one, 1, 1, 1, 1, 4, 0.1, None) == -1001  # fused LeakyReLU needs `out` or sign_bits
    assert lib.arflow_corr_fwd(one, one, one, one, 1, 1, 1, 1, 4, 0.1, None) == -1003  # no sign_bits off the fast path
    assert lib.arflow_corr_sign_planes(32, 160, 4) == 3 and lib.arflow_corr_sign_planes(32, 10, 4) == 0
    assert lib.arflow_warp_fwd(one, one, one, None, 1, 1, 4, 4, 4, 4, 32, 7, 1, 0, None) == -1003
    assert lib.arflow_warp_fwd(one, one, one, None, 1, 1, 4, 4, 4, 4, 3, 0, 1, 0, None) == -1002
    assert lib.arflow_census_fwd(one, one, None, one, None, None, 1, 8, 8, 17, None) == -1003  # radius 1..16
    assert lib.arflow_down4(one, one, 1, 6, 8, None) == -1002
    assert b'NULL' in lib.arflow_strerror(-1001)


def test_ops_refuse_cpu_tensors_loudly():
    from arflow_amd import functional as AF, _lib
    from arflow_amd.warp_utils import flow_warp
    from arflow_amd.uflow_utils import census_loss
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(_lib.ArflowHipError):
        AF.correlation(x, x, 4)
    with pytest.raises(_lib.ArflowHipError):
        flow_warp(x, torch.zeros(1, 2, 8, 8))
    with pytest.raises(_lib.ArflowHipError):
        census_loss(x, x, torch.ones(1, 1, 8, 8))


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under arflow_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'arflow_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(dirpath, f)


def test_stale_error_slot_is_polled_and_cleared(lib):
    """the slot a pending foreign HIP error is parked in must be read (and so cleared) by the Python layer after
    every call, and surfaced -- checked here with a stand-in library object (no GPU needed)."""
    import warnings
    from arflow_amd import _lib
    lib.arflow_take_stale_error()  # (a GPU-less host may have parked hipErrorNoDevice from the calls of the tests above)
    assert lib.arflow_take_stale_error() == 0  # reading clears the slot

    class Fake:
        def __init__(self):
            self.code, self.reads = 719, 0

        def arflow_take_stale_error(self):
            self.reads += 1
            c, self.code = self.code, 0
            return c
    fake = Fake()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        _lib.poll_stale_error(fake, 'arflow_corr_fwd')
        _lib.poll_stale_error(fake, 'arflow_corr_fwd')
    assert fake.reads == 2 and len(w) == 1 and '719' in str(w[0].message)
    import inspect
    from arflow_amd import functional
    assert 'poll_stale_error' in inspect.getsource(functional._call)


def test_correlation_signature_follows_correlation_native():
    from arflow_amd.correlation import Correlation
    c = Correlation(3)  # models/correlation_native.py:7: max_displacement is the first positional argument
    assert c.max_displacement == 3 and c.output_dim == 7 and c.pad_size == 3 and c.general is None
    c = Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)  # models/pwclite.py:124
    assert c.general is None and c.output_dim == 9
    c = Correlation(pad_size=3, kernel_size=3, max_displacement=2, stride1=1, stride2=2)
    assert c.general == (3, 3, 2, 1, 2) and c.pad_size == 3 and c.output_dim == 3
