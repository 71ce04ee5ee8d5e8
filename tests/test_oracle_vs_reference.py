"""Build container only (skipped wherever /root/reference is absent, i.e. on the GPU box): the CPU oracle
against the REFERENCE's own pure-PyTorch functions, run live on fresh seeded inputs -- a random sweep on top of
the frozen vectors of tests/golden (which tests/test_oracle_golden.py replays everywhere).  The reference is
imported read-only from /root/reference exactly as oracle/make_golden.py does; nothing of it is stored."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.reference
REF = os.environ.get('ARFLOW_REFERENCE', '/root/reference')


@pytest.fixture(scope='module')
def ref():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        from models.correlation_native import Correlation
        from utils import warp_utils, uflow_utils
        from losses import loss_blocks
        yield dict(Correlation=Correlation, W=warp_utils, U=uflow_utils, LB=loss_blocks)
    finally:
        sys.path.remove(REF)
        for m in [k for k, v in sys.modules.items() if getattr(v, '__file__', None) and str(v.__file__).startswith(REF)]:
            del sys.modules[m]


def _close(a, b, atol, rtol, what):
    err = (a.double() - b.double()).abs()
    assert bool((err <= atol + rtol * b.double().abs()).all()), '%s: max err %.3e' % (what, float(err.max()))


@pytest.mark.parametrize('shape', [(2, 32, 24, 40), (1, 7, 9, 33), (3, 16, 8, 14)])
def test_correlation_live(ref, shape):
    from oracle import ops as O
    g = torch.Generator().manual_seed(sum(shape))
    x1 = torch.randn(*shape, generator=g).requires_grad_(True)
    x2 = torch.randn(*shape, generator=g).requires_grad_(True)
    go = torch.randn(shape[0], 81, *shape[2:], generator=g)
    y_ref = ref['Correlation'](pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)(x1, x2)
    r1, r2 = torch.autograd.grad(y_ref, [x1, x2], go)
    _close(O.correlation(x1, x2, 4), y_ref, 1e-6, 1e-5, 'corr fwd')  # models/correlation_native.py:13-23
    o1, o2 = O.correlation_backward(go, x1.detach(), x2.detach(), 4)
    _close(o1, r1, 5e-6, 1e-5, 'corr gx1')
    _close(o2, r2, 5e-6, 1e-5, 'corr gx2')


@pytest.mark.parametrize('pad', ['zeros', 'border'])
@pytest.mark.parametrize('ac', [True, False])
def test_flow_warp_live(ref, pad, ac):
    from oracle import ops as O
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 19, 27, generator=g)
    fl = 3 * torch.randn(2, 2, 19, 27, generator=g)
    go = torch.randn(2, 6, 19, 27, generator=g)
    outs = []
    for fn in (ref['W'].flow_warp, O.flow_warp):  # utils/warp_utils.py:83-90
        a, f = x.clone().requires_grad_(True), fl.clone().requires_grad_(True)
        y = fn(a, f, pad=pad, align_corners=ac)
        outs.append((y,) + torch.autograd.grad(y, [a, f], go))
    for a, b, n in zip(outs[1], outs[0], ('fwd', 'gsrc', 'gflow')):
        _close(a, b, 2e-5, 1e-4, 'flow_warp ' + n)


def test_masks_and_losses_live(ref):
    from oracle import ops as O
    g = torch.Generator().manual_seed(11)
    fl = 2.5 * torch.randn(2, 2, 24, 31, generator=g)
    im1, im2 = torch.rand(2, 3, 24, 31, generator=g), torch.rand(2, 3, 24, 31, generator=g)
    mask = (torch.rand(2, 1, 24, 31, generator=g) > 0.3).float()
    _close(O.compute_range_map(fl), ref['U'].compute_range_map(fl), 1e-6, 1e-5, 'range map')  # uflow_utils.py:80-160
    _close(O.border_mask(fl), ref['W'].border_mask(fl), 0, 0, 'border mask')                    # warp_utils.py:119-134
    _close(O.mask_invalid(O.flow_to_warp(fl)), ref['U'].mask_invalid(ref['U'].flow_to_warp(fl)), 0, 0, 'mask_invalid')
    for ps in (7, 3):
        outs = []
        for fn in (ref['U'].census_loss, O.census_loss):  # utils/uflow_utils.py:282-293
            b = im2.clone().requires_grad_(True)
            y = fn(im1, b, mask, ps)
            outs.append((y,) + torch.autograd.grad(y, [b]))
        _close(outs[1][0], outs[0][0], 1e-6, 1e-5, 'census loss')
        _close(outs[1][1], outs[0][1], 1e-7, 1e-4, 'census grad')
    _close(O.ssim(im1, im2), ref['LB'].SSIM(im1, im2), 5e-6, 1e-5, 'ssim')                    # loss_blocks.py:65-84
    _close(O.smooth_grad_1st(fl, im1, 10.), ref['LB'].smooth_grad_1st(fl, im1, 10.), 1e-7, 1e-5, 'smooth 1st')
    _close(O.smooth_grad_2nd(fl, im1, 10.), ref['LB'].smooth_grad_2nd(fl, im1, 10.), 1e-7, 1e-5, 'smooth 2nd')
