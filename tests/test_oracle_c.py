"""CPU: the plain-C restatement (oracle/corr_oracle.c) against the golden vectors and oracle/ops.py --
three independent implementations of the cost volume must agree."""
import ctypes
import os
import subprocess

import numpy as np
import torch

from oracle import ops
from tests.conftest import assert_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    so = os.path.join(ROOT, 'oracle', 'libcorr_oracle.so')
    if not os.path.exists(so):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.corr_oracle_fwd.argtypes = [fp, fp, fp] + [ctypes.c_int] * 5
    lib.corr_oracle_bwd.argtypes = [fp] * 5 + [ctypes.c_int] * 5
    return lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def test_c_restatement_matches_golden_and_torch_oracle(golden):
    lib = _lib()
    g = golden('corr')
    for name in g.names():
        x1, x2, go = (np.ascontiguousarray(g.raw(name + k)) for k in ('_x1', '_x2', '_g'))
        d = int(g[name + '_d'])
        B, C, H, W = x1.shape
        out = np.empty((B, (2 * d + 1) ** 2, H, W), np.float32)
        lib.corr_oracle_fwd(_p(x1), _p(x2), _p(out), B, C, H, W, d)
        assert_close(torch.from_numpy(out), g[name + '_y'], 1e-6, 1e-5, name + ' C fwd vs reference')
        assert_close(torch.from_numpy(out), ops.correlation(torch.from_numpy(x1), torch.from_numpy(x2), d), 1e-6, 1e-5,
                     name + ' C fwd vs torch oracle')
        g1, g2 = np.empty_like(x1), np.empty_like(x2)
        lib.corr_oracle_bwd(_p(go), _p(x1), _p(x2), _p(g1), _p(g2), B, C, H, W, d)
        assert_close(torch.from_numpy(g1), g[name + '_gx1'], 5e-6, 1e-5, name + ' C gx1')
        assert_close(torch.from_numpy(g2), g[name + '_gx2'], 5e-6, 1e-5, name + ' C gx2')
