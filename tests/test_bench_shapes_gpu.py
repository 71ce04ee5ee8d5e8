"""GPU parity AT THE BASELINE SHAPES: every kernel instantiation bench.py times is compared with the CPU
oracle on the same seeded inputs, at the exact launch shapes of BASELINE configs 2, 3 and 4 (VERDICT r1,
"what's weak" 1-2).  The oracle needs seconds per case on the GPU box's host cores.

  config 2  (384x640, batch 8, fw+bw stacked -> B=16): PWCLiteUflow levels C=32 @ 12x20 .. 96x160,
            PWCLite levels C=192@6x10 .. 32@96x160; full-resolution loss kernels at [8,3,384,640]
  config 3  (448x1024, batch 4 -> B=8): levels 7x16 .. 112x256, loss at [4,3,448,1024]
  config 4  (256x448, batch 8 per GPU -> B=16): levels 8x14 (W % 4 != 0: pad-and-crop path) .. 64x112

The launcher's choice of template instantiation depends on the tile count, so the shapes below reach
corr_v2::fwd_kernel<2,1> / <4,1> / <2,4>, bwd_kernel<2,ACT> / <4,ACT> for ACT in {0,2}, warp_fwd_kernel<2>/<3>,
the slab / list forms of the feature-warp gradient, featnorm's large-n path (n = 491 520) and the small one,
census4 / photo4 at full resolution.  Tolerances are the stated fp32 ones of test_hip_parity.py.

Tolerances (round 3): every assert_close() below was re-derived from the error MEASURED on MI355X -- tests/conftest.py
records max(err / tol) per call site, profiles/r03_parity_margins.json holds the summary -- and sites that had more than
20x headroom were divided down (the `/ N` factors and the small literals) so that each keeps about 10x over its measured
error (float atomics and summation order move the error by 2-3x from run to run).  Sites left as they were sit within
20x of their measured error already.
"""
import pytest
import torch

from tests.conftest import assert_close

pytestmark = pytest.mark.gpu


def cu(t):
    return t.cuda()


@pytest.fixture(scope='module')
def AF():
    from arflow_amd import functional
    return functional


@pytest.fixture(scope='module')
def O():
    from oracle import ops
    torch.set_num_threads(16)
    return ops


CORR_SHAPES = [
    # config 2, PWCLiteUflow / PWCFlow pyramid (fw and bw stacked on the batch axis)
    (16, 32, 96, 160), (16, 32, 48, 80), (16, 32, 24, 40), (16, 32, 12, 20),
    # config 2, ARFlow PWCLite pyramid
    (16, 64, 48, 80), (16, 96, 24, 40), (16, 128, 12, 20), (16, 192, 6, 10),
    # config 3 (448x1024, batch 4)
    (8, 32, 112, 256), (8, 32, 56, 128), (8, 32, 28, 64), (8, 32, 14, 32),
    # config 4 (256x448, batch 8 per GPU); 8x14 and 16x28 ... widths 14 -> padded to 16
    (16, 32, 64, 112), (16, 32, 32, 56), (16, 32, 16, 28), (16, 32, 8, 14),
]


@pytest.mark.parametrize('slope', [1.0, 0.1], ids=['plain', 'leaky'])
@pytest.mark.parametrize('shape', CORR_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_correlation_at_bench_shapes(AF, O, shape, slope):
    """models/correlation_native.py:13-23 (+ the LeakyReLU(0.1) of models/pwclite.py:184 fused): forward and
    both gradients.  Output gradients are zeroed where the pre-activation is within 1e-6 of 0 -- there the
    LeakyReLU branch is decided by summation order and either derivative is right."""
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(B * 7 + C * 3 + H)
    x1 = torch.randn(B, C, H, W, generator=gen)
    x2 = torch.randn(B, C, H, W, generator=gen)
    go = torch.randn(B, 81, H, W, generator=gen)
    pre = O.correlation(x1, x2, 4)
    if slope != 1.0:
        go = go * ((pre.abs() > 1e-6).float())
        ref = torch.nn.functional.leaky_relu(pre, slope)
        gpre = go * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope))
    else:
        ref, gpre = pre, go
    r1, r2 = O.correlation_backward(gpre, x1, x2, 4)
    a, b = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
    y = AF.correlation(a, b, 4, negative_slope=slope)
    assert_close(y, ref, 1e-6, 1e-5, 'corr fwd %s' % (shape,))
    g1, g2 = torch.autograd.grad(y, [a, b], cu(go))
    assert_close(g1, r1, 5e-6, 1e-5, 'corr gx1 %s' % (shape,))
    assert_close(g2, r2, 5e-6, 1e-5, 'corr gx2 %s' % (shape,))


WARP_SHAPES = [
    (16, 32, 96, 160, 'zeros', True), (16, 32, 48, 80, 'zeros', True), (16, 32, 24, 40, 'zeros', True),
    (8, 32, 112, 256, 'zeros', True), (16, 32, 64, 112, 'zeros', True), (16, 32, 8, 14, 'zeros', True),
    (16, 64, 48, 80, 'border', True),  # ARFlow PWCLite level (flow_warp default pad in the loss is border)
]


@pytest.mark.parametrize('cfg', WARP_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_feature_warp_at_bench_shapes(AF, O, cfg):
    """utils/warp_utils.py:83-90 on feature maps with a ~3 px flow: forward, d/d src (the scatter), d/d flow."""
    B, C, H, W, pad, ac = cfg
    gen = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(B, C, H, W, generator=gen)
    # a smooth field plus 0.3 px of noise, 3 px rms: what a trained / randomly initialised estimator produces
    coarse = 3.0 * torch.randn(B, 2, max(2, H // 8), max(2, W // 8), generator=gen)
    fl = torch.nn.functional.interpolate(coarse, (H, W), mode='bilinear', align_corners=True) + \
        0.3 * torch.randn(B, 2, H, W, generator=gen)
    go = torch.randn(B, C, H, W, generator=gen)
    xr, fr = x.clone().requires_grad_(True), fl.clone().requires_grad_(True)
    ref = O.flow_warp(xr, fr, pad=pad, align_corners=ac)
    rgx, rgf = torch.autograd.grad(ref, [xr, fr], go)
    a, f = cu(x).requires_grad_(True), cu(fl).requires_grad_(True)
    y = AF.warp(a, f, pad=pad, align_corners=ac)
    ulp = 2.0 ** -23 * max(H, W)
    assert_close(y, ref, ((2e-6 + 4 * ulp) * float(x.abs().max())) / 20, 5e-7, 'warp fwd')
    gx, gf = torch.autograd.grad(y, [a, f], cu(go))
    assert_close(gx, rgx, (1e-5 * max(1.0, float(rgx.abs().max()))) / 10, 1e-5, 'warp gsrc')
    assert_close(gf, rgf, (1e-5 * (C ** 0.5) * float(x.abs().max()) * 4) / 20, 1e-5, 'warp gflow')


def test_feature_warp_noisy_flow_at_bench_shape(AF, O):
    """The same at [16,32,96,160] with 3 px WHITE-NOISE flow (what kbench times): every tile's tap box is large."""
    B, C, H, W = 16, 32, 96, 160
    gen = torch.Generator().manual_seed(17)
    x = torch.randn(B, C, H, W, generator=gen)
    fl = 3.0 * torch.randn(B, 2, H, W, generator=gen)
    go = torch.randn(B, C, H, W, generator=gen)
    xr, fr = x.clone().requires_grad_(True), fl.clone().requires_grad_(True)
    rgx, rgf = torch.autograd.grad(O.flow_warp(xr, fr), [xr, fr], go)
    a, f = cu(x).requires_grad_(True), cu(fl).requires_grad_(True)
    gx, gf = torch.autograd.grad(AF.warp(a, f), [a, f], cu(go))
    assert_close(gx, rgx, (1e-5 * max(1.0, float(rgx.abs().max()))) / 10, 1e-5, 'warp gsrc (noise)')
    assert_close(gf, rgf, (1e-5 * (C ** 0.5) * float(x.abs().max()) * 4) / 20, 1e-5, 'warp gflow (noise)')


@pytest.mark.parametrize('mode', ['joint', 'avg'])
@pytest.mark.parametrize('shape', [(16, 32, 96, 160), (16, 32, 12, 20), (8, 32, 112, 256), (16, 32, 8, 14)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_feature_normalisation_at_bench_shapes(AF, O, mode, shape):
    """normalize_features (models/pwclite_uflow.py:30-38 'joint', models/uflow_model.py:8-50 'avg') at
    n = C*h*w = 491 520 (two-pass path) and at the coarse levels (one-workgroup path)."""
    gen = torch.Generator().manual_seed(sum(shape))
    x1 = torch.randn(*shape, generator=gen) * 0.7 + 0.2
    x2 = torch.randn(*shape, generator=gen) * 1.3 - 0.1
    g1, g2 = torch.randn(*shape, generator=gen), torch.randn(*shape, generator=gen)
    fn = O.normalize_features_joint if mode == 'joint' else O.normalize_features_uflow
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    r1, r2 = fn([a, b])
    ra, rb = torch.autograd.grad([r1, r2], [a, b], [g1, g2])
    ac, bc = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
    y1, y2 = AF.normalize_pair(ac, bc, mode)
    assert_close(y1, r1, 2e-6, 2e-6, 'y1')
    assert_close(y2, r2, 5e-6, 5e-6, 'y2')
    da, db = torch.autograd.grad([y1, y2], [ac, bc], [cu(g1), cu(g2)])
    gt = max(float(ra.abs().max()), float(rb.abs().max()))
    assert_close(da, ra, (2e-5 * gt) / 20, 5e-6, 'gx1')
    assert_close(db, rb, (2e-5 * gt) / 20, 5e-6, 'gx2')


def _pair(B, H, W, gen):
    from oracle.fixture_common import synth_pair
    return synth_pair(B, H, W, gen)[0]


@pytest.mark.parametrize('size', [(8, 384, 640), (4, 448, 1024), (8, 256, 448)], ids=lambda s: 'x'.join(map(str, s)))
def test_census_loss_at_bench_shapes(O, size):
    """utils/uflow_utils.py:282-293 at full resolution (census4::fwd/bwd_kernel<3>), loss and d/d image_b."""
    from arflow_amd import uflow_utils as U
    B, H, W = size
    gen = torch.Generator().manual_seed(H + W)
    img = _pair(B, H, W, gen)
    im1, im2 = img[:, :3].contiguous(), img[:, 3:].contiguous()
    mask = (torch.rand(B, 1, H, W, generator=gen) > 0.15).float()
    b = im2.clone().requires_grad_(True)
    ref = O.census_loss(im1, b, mask)
    rg, = torch.autograd.grad(ref, [b])
    bc = cu(im2).requires_grad_(True)
    y = U.census_loss(cu(im1), bc, cu(mask))
    assert_close(y, ref, 2e-7, 2e-6, 'census loss')
    gb, = torch.autograd.grad(y, [bc])
    assert_close(gb, rg, (1e-4 * float(rg.abs().max())) / 20, 5e-6, 'census grad')


@pytest.mark.parametrize('size', [(8, 384, 640), (4, 448, 1024)], ids=lambda s: 'x'.join(map(str, s)))
def test_photometric_sums_at_bench_shapes(AF, O, size):
    """losses/flow_loss.py:13-27 at full resolution (photo4::fwd/bwd_kernel): the three sums and d/d recons."""
    B, H, W = size
    gen = torch.Generator().manual_seed(H * 3 + W)
    img = _pair(B, H, W, gen)
    im, rec0 = img[:, :3].contiguous(), img[:, 3:].contiguous()
    mask = (torch.rand(B, 1, H, W, generator=gen) > 0.2).float()
    rec = rec0.clone().requires_grad_(True)
    l1 = ((im - rec).abs() * mask).sum()
    ss = O.ssim(rec * mask, im * mask).sum()
    rg, = torch.autograd.grad(0.3 * l1 + 0.7 * ss, [rec])
    rc = cu(rec0).requires_grad_(True)
    s = AF.PhotoSumsFunction.apply(cu(im), rc, cu(mask))
    assert_close(s[0], l1, 5e-5, 1e-6, 'sum |im - rec| mask')
    assert_close(s[1], ss, 0.0001, 5e-6, 'sum SSIM distance')
    assert_close(s[2], mask.sum(), 0, 1e-7, 'sum mask')
    gg, = torch.autograd.grad(0.3 * s[0] + 0.7 * s[1], [rc])
    assert_close(gg, rg, (1e-3 * float(rg.abs().max())) / 20, 5e-5, 'd / d recons')


@pytest.mark.parametrize('size,order', [((2, 448, 1024), 1), ((2, 384, 640), 2), ((2, 256, 448), 1)],
                         ids=['config3', 'config2-order2', 'config4'])
def test_uflow_loss_end_to_end_at_bench_resolution(size, order):
    """losses/uflow_loss.py:8-109 end to end at the BASELINE resolutions against oracle/losses.py: all four loss
    terms, the mask and the gradients w.r.t. both flow levels it reads."""
    from arflow_amd.config import AttrDict
    from arflow_amd.losses import UFlowLoss
    from oracle import losses as OL
    B, H, W = size
    gen = torch.Generator().manual_seed(H)
    img = _pair(B, H, W, gen)
    sizes = [(H, W), (H // 2, W // 2), (H // 4, W // 4)]
    flows = []
    for i, (h, w) in enumerate(sizes):
        coarse = (8.0 / 2 ** i) * torch.randn(B, 4, 6, 10, generator=gen)
        flows.append(torch.nn.functional.interpolate(coarse, (h, w), mode='bilinear', align_corners=True) +
                     0.2 * torch.randn(B, 4, h, w, generator=gen))
    cfg = AttrDict(edge_constant=150, w_smooth=4.0, w_census=1.0, with_bk=True, smooth_order=order)
    fr = [f.clone().requires_grad_(True) for f in flows]
    ref = OL.UFlowLoss(cfg)(fr, img)
    rg = torch.autograd.grad(ref[0], [fr[0], fr[2]])
    fc = [cu(f).requires_grad_(True) for f in flows]
    got = UFlowLoss(cfg)(fc, cu(img))
    names = ['total', 'census', 'smooth', '|flow|']
    for k in range(4):
        assert_close(got[k], ref[k], 5e-8, 3e-6, 'uflow loss ' + names[k])
    # the occlusion mask is a bilinear upsample of a clamped splat map times a validity mask: continuous
    assert_close(got[4], ref[4], 2e-6, 1e-5, 'mask1')
    gg = torch.autograd.grad(got[0], [fc[0], fc[2]])
    for a, b, n in zip(gg, rg, ('flow0', 'flow2')):
        assert_close(a, b, 2e-7 + 2e-4 * float(b.abs().max()), 2e-3, 'd loss / d ' + n)


def test_unflow_loss_end_to_end_at_bench_resolution():
    """losses/flow_loss.py:38-114 (the ARFlow pyramid loss of 'pwclite+unflow_loss') at 2 x 384x640, 6 levels."""
    from arflow_amd.config import AttrDict
    from arflow_amd.losses import unFlowLoss
    from oracle import losses as OL
    B, H, W = 2, 384, 640
    gen = torch.Generator().manual_seed(5)
    img = _pair(B, H, W, gen)
    sizes = [(H, W)] + [(H // s, W // s) for s in (4, 8, 16, 32, 64)]
    flows = []
    for i, (h, w) in enumerate(sizes):
        coarse = (6.0 * h / H) * torch.randn(B, 4, 3, 5, generator=gen)
        flows.append(torch.nn.functional.interpolate(coarse, (h, w), mode='bilinear', align_corners=True) +
                     0.05 * torch.randn(B, 4, h, w, generator=gen))
    cfg = AttrDict(w_l1=0.15, w_ssim=0.85, w_ternary=0.0, warp_pad='border', alpha=10, occ_from_back=True, with_bk=True,
                   w_smooth=75.0, w_scales=[1.0, 1.0, 1.0, 1.0, 1.0, 0.0], w_sm_scales=[1.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    fr = [f.clone().requires_grad_(True) for f in flows]
    ref = OL.unFlowLoss(cfg)(fr, img)
    rg = torch.autograd.grad(ref[0], fr[:5])
    fc = [cu(f).requires_grad_(True) for f in flows]
    got = unFlowLoss(cfg)(fc, cu(img))
    for k in range(4):
        assert_close(got[k], ref[k], 2e-8, 1e-6, 'unflow loss term %d' % k)
    gg = torch.autograd.grad(got[0], fc[:5])
    for i, (a, b) in enumerate(zip(gg, rg)):
        assert_close(a, b, 2e-7 + 2e-4 * float(b.abs().max()), 2e-3, 'd loss / d flow%d' % i)
