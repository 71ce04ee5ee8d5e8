"""CPU: the oracle (oracle/ops.py, oracle/losses.py) against the golden vectors frozen from the
reference by oracle/make_golden.py.  fp32, same inputs; tolerances are rounding-level because the
oracle restates the same arithmetic."""
import pytest
import torch

from oracle import ops, losses as olosses
from arflow_amd.config import AttrDict
from tests.conftest import assert_close

A, R = 2e-6, 1e-5


def test_correlation_forward_backward(golden):
    g = golden('corr')
    for name in g.names():
        x1 = g[name + '_x1'].requires_grad_(True)
        x2 = g[name + '_x2'].requires_grad_(True)
        d = int(g[name + '_d'])
        y = ops.correlation(x1, x2, d)
        assert_close(y, g[name + '_y'], 1e-6, 1e-5, name + ' fwd')
        gx1, gx2 = torch.autograd.grad(y, [x1, x2], g[name + '_g'])
        assert_close(gx1, g[name + '_gx1'], 1e-6, 1e-5, name + ' gx1')
        assert_close(gx2, g[name + '_gx2'], 1e-6, 1e-5, name + ' gx2')
        e1, e2 = ops.correlation_backward(g[name + '_g'], x1.detach(), x2.detach(), d)
        # the closed form sums 81 O(1) products before dividing by C (the reference's autograd divides
        # each product first): a different fp32 summation order, hence the wider absolute tolerance
        assert_close(e1, g[name + '_gx1'], 5e-6, 1e-5, name + ' closed-form gx1')
        assert_close(e2, g[name + '_gx2'], 5e-6, 1e-5, name + ' closed-form gx2')


def test_appendix_a_anchors(golden):
    """SURVEY Appendix A known answers for the analytic inputs."""
    g = golden('corr')
    y = ops.correlation(g['analytic_x1'], g['analytic_x2'], 4)
    assert tuple(y.shape) == (2, 81, 12, 20)
    assert abs(float(y.sum()) - (-6.518948)) < 2e-3
    assert abs(float(y.abs().sum()) - 4227.336914) < 5e-2
    assert abs(float(y[0, 40, 5, 7]) - (-0.1410838)) < 1e-6
    assert float(y[1, 0, 0, 0]) == 0.0
    w = golden('warp')
    out = ops.flow_warp(w['analytic_x'], w['analytic_flow'], 'zeros', align_corners=True)
    assert abs(float(out.sum()) - 62.051033) < 1e-3
    m = golden('masks')
    fl = m['analytic_flow']
    assert abs(float(ops.compute_range_map(fl).sum()) - 288.760803) < 1e-3
    assert float(ops.get_occu_mask_backward(fl, .2).sum()) == 33
    assert float(ops.get_occu_mask_bidirection(fl, -fl).sum()) == 416
    assert float(ops.border_mask(fl).sum()) == 259
    assert float(ops.mask_invalid(ops.flow_to_warp(fl)).sum()) == 259
    p = golden('photo')
    im1, im2 = p['analytic_im1'], p['analytic_im2']
    s = ops.ssim(im1, im2)
    assert tuple(s.shape) == (2, 3, 10, 18) and abs(float(s.mean()) - 0.39956456) < 1e-6
    assert abs(float(ops.census_loss(im1, im2, torch.ones(2, 1, 12, 20))) - 3.30243874) < 1e-5
    assert abs(float(ops.smooth_grad_1st(fl, im1, 10)) - 0.08836684) < 1e-7
    assert abs(float(ops.smooth_grad_2nd(fl, im1, 10)) - 0.03248186) < 1e-7


def test_flow_warp_all_modes(golden):
    g = golden('warp')
    for name in g.names():
        for pad in ('zeros', 'border'):
            for ac in (True, False):
                x = g[name + '_x'].requires_grad_(True)
                fl = g[name + '_flow'].requires_grad_(True)
                tag = '%s_%s_%d' % (name, pad, int(ac))
                y = ops.flow_warp(x, fl, pad=pad, align_corners=ac)
                scale = float(x.abs().max())
                assert_close(y, g[tag + '_y'], 2e-6 * scale, 1e-5, tag + ' fwd')
                gx, gf = torch.autograd.grad(y, [x, fl], g[name + '_g'])
                assert_close(gx, g[tag + '_gx'], 1e-5, 1e-4, tag + ' gx')
                assert_close(gf, g[tag + '_gf'], 2e-5, 1e-4, tag + ' gflow')


def test_resample_family(golden):
    g = golden('warp')
    for name in g.names():
        x = g[name + '_x'].requires_grad_(True)
        fl = g[name + '_flow'].requires_grad_(True)
        coords = ops.flow_to_warp(fl)
        assert_close(coords, g[name + '_coords'], 0, 0, name + ' flow_to_warp')
        assert_close(ops.mask_invalid(coords), g[name + '_mask_invalid'], 0, 0, name + ' mask_invalid')
        y = ops.resample(x, coords)
        assert_close(y, g[name + '_resample_y'], 2e-6 * float(x.abs().max()), 1e-5, name + ' resample')
        gx, gf = torch.autograd.grad(y, [x, fl], g[name + '_g'])
        assert_close(gx, g[name + '_resample_gx'], 1e-5, 1e-4, name + ' resample gx')
        assert_close(gf, g[name + '_resample_gf'], 2e-5, 1e-4, name + ' resample gflow')
        nhwc = ops.resampler_nhwc(x.detach().permute(0, 2, 3, 1).contiguous(),
                                  coords.detach().permute(0, 2, 3, 1).contiguous())
        assert_close(nhwc, g[name + '_resampler_nhwc'], 1e-6, 1e-5, name + ' resampler nhwc')


def test_splat_maps_and_masks(golden):
    g = golden('masks')
    for name in g.names():
        fl = g[name + '_flow']
        rm = ops.compute_range_map(fl)
        assert_close(rm, g[name + '_range_map'], 1e-6, 1e-5, name + ' range map')
        assert_close(rm, g[name + '_range_map_wu'], 1e-6, 1e-5, name + ' range map (warp_utils dup)')
        cm = ops.get_corresponding_map(ops.flow_to_warp(fl))
        assert_close(cm, g[name + '_corr_map'], 1e-6, 1e-5, name + ' corresponding map')
        assert_close(ops.get_occu_mask_backward(fl, .2), g[name + '_occ_back_02'], 0, 0, name + ' occ back')
        assert_close(ops.get_occu_mask_backward(fl, 0.), g[name + '_occ_back_0'], 1e-6, 1e-5, name + ' occ back soft')
        assert_close(ops.get_occu_mask_bidirection(fl, -0.7 * fl.flip(-1)), g[name + '_occ_bidir'], 0, 0, name)
        assert_close(ops.get_occu_mask_bidirection(fl, -fl), g[name + '_occ_bidir_neg'], 0, 0, name)
        assert_close(ops.border_mask(fl), g[name + '_border_mask'], 0, 0, name + ' border mask')


def test_photometric_blocks(golden):
    g = golden('photo')
    for name in g.names():
        im1, im2, fl, mask = g[name + '_im1'], g[name + '_im2'], g[name + '_flow'], g[name + '_mask']
        a = im1.clone().requires_grad_(True)
        b = im2.clone().requires_grad_(True)
        y = ops.ssim(a, b)
        assert_close(y, g[name + '_ssim'], A, R, name + ' ssim')
        ga, gb = torch.autograd.grad(y, [a, b], g[name + '_ssim_g'])
        assert_close(ga, g[name + '_ssim_ga'], 2e-5, 1e-4, name + ' ssim ga')
        assert_close(gb, g[name + '_ssim_gb'], 2e-5, 1e-4, name + ' ssim gb')
        for md, sd in ((1, False), (3, True)):
            tag = '%s_ternary_%d_%d' % (name, md, int(sd))
            a = im1.clone().requires_grad_(True)
            b = im2.clone().requires_grad_(True)
            dist, tm = ops.ternary_loss(a, b, md, sd)
            assert_close(dist, g[tag + '_dist'], 1e-5, 1e-5, tag)
            assert_close(tm, g[tag + '_mask'], 0, 0, tag + ' mask')
            ga, gb = torch.autograd.grad(dist, [a, b], g[tag + '_g'])
            assert_close(ga, g[tag + '_ga'], 1e-3, 1e-4, tag + ' ga')
            assert_close(gb, g[tag + '_gb'], 1e-3, 1e-4, tag + ' gb')
        for ps in (7, 3):
            b = im2.clone().requires_grad_(True)
            y = ops.census_loss(im1, b, mask, ps)
            assert_close(y, g['%s_census_%d' % (name, ps)], 1e-6, 1e-5, name + ' census')
            gb, = torch.autograd.grad(y, [b])
            assert_close(gb, g['%s_census_%d_gb' % (name, ps)], 1e-6, 1e-4, name + ' census gb')
        assert_close(ops.census_loss(im1, im2, torch.ones_like(mask)), g[name + '_census_ones'], 1e-6, 1e-5, name)
        for fn, key in ((lambda f: ops.smooth_grad_1st(f, im1, 10.), 'sm1_abs'),
                        (lambda f: ops.smooth_grad_1st(f, im1, 10., penalty='uflow'), 'sm1_uflow'),
                        (lambda f: ops.smooth_grad_2nd(f, im1, 10.), 'sm2')):
            f = fl.clone().requires_grad_(True)
            y = fn(f)
            assert_close(y, g['%s_%s' % (name, key)], 1e-7, 1e-5, name + ' ' + key)
            gf, = torch.autograd.grad(y, [f])
            assert_close(gf, g['%s_%s_gf' % (name, key)], 1e-8, 1e-4, name + ' ' + key + ' grad')


def test_resize_and_normalisation(golden):
    g = golden('aux')
    img, m = g['img'], g['m']
    assert_close(ops.downsample(img, False, 4.0), g['down4'], 1e-6, 1e-5, 'down4')
    assert_close(ops.downsample4_explicit(img), g['down4'], 1e-6, 1e-5, 'down4 explicit 2x2-centre mean')
    assert_close(ops.upsample(m, False, 4.0), g['up4'], 1e-6, 1e-5, 'up4')
    assert_close(ops.upsample(m, True), g['up2_flow'], 1e-6, 1e-5, 'up2 flow')
    assert_close(ops.downsample(img, True), g['down2_flow'], 1e-6, 1e-5, 'down2 flow')
    a, b = ops.normalize_features_joint([g['f1'], g['f2']])
    assert_close(a, g['nj_1'], 1e-6, 1e-5, 'joint norm 1')
    assert_close(b, g['nj_2'], 1e-6, 1e-5, 'joint norm 2')
    c, d = ops.normalize_features_uflow([g['f1'], g['f2']])
    assert_close(c, g['nu_1'], 1e-6, 1e-5, 'uflow norm 1')
    assert_close(d, g['nu_2'], 1e-6, 1e-5, 'uflow norm 2')


def _loss_cases():
    from oracle.fixture_common import loss_cfgs
    return loss_cfgs()


@pytest.mark.parametrize('group', [0, 1, 2])
def test_loss_modules(golden, group):
    g = golden('losses')
    cases = _loss_cases()[group]
    cls = (olosses.UFlowLoss, olosses.unFlowLoss, olosses.FullResLoss)[group]
    img = g['img']
    for name, cfg in cases:
        flows = [g['flow%d' % i].clone().requires_grad_(True) for i in range(5)]
        res = cls(AttrDict(cfg))(flows, img)
        assert_close(res[0], g[name + '_total'], 1e-6, 2e-5, name + ' total')
        assert_close(res[1], g[name + '_warp'], 1e-6, 2e-5, name + ' warp')
        assert_close(res[2], g[name + '_smooth'], 1e-6, 2e-5, name + ' smooth')
        assert_close(res[3], g[name + '_absflow'], 1e-6, 1e-5, name + ' |flow|')
        if len(res) > 4:
            assert_close(res[4], g[name + '_mask1'], 1e-6, 1e-5, name + ' mask1')
        grads = torch.autograd.grad(res[0], flows, allow_unused=True)
        for i, gi in enumerate(grads):
            ref = g['%s_g%d' % (name, i)]
            gi = gi if gi is not None else torch.zeros_like(ref)
            assert_close(gi, ref, 2e-7 + 1e-4 * float(ref.abs().max()), 1e-3, '%s dflow%d' % (name, i))


def test_unflow_ternary_raises_like_reference():
    cfg = AttrDict(w_l1=0.1, w_ssim=0.1, w_ternary=0.5, warp_pad='border', alpha=10, occ_from_back=True,
                   with_bk=False, w_smooth=1.0, w_scales=[1.0], w_sm_scales=[1.0])
    with pytest.raises(TypeError):
        olosses.unFlowLoss(cfg)([torch.zeros(1, 4, 8, 8)], torch.rand(1, 6, 8, 8))


def test_sampler_restatement_matches_torch_grid_sample():
    """The explicit bilinear sampler must agree with the installed torch kernel the reference calls."""
    torch.manual_seed(0)
    src = torch.randn(2, 3, 9, 7, dtype=torch.float64, requires_grad=True)
    grid = (torch.rand(2, 5, 6, 2, dtype=torch.float64) * 2.6 - 1.3).requires_grad_(True)
    for pad in ('zeros', 'border'):
        for ac in (True, False):
            ref = torch.nn.functional.grid_sample(src, grid, mode='bilinear', padding_mode=pad, align_corners=ac)
            ix = ops._unnormalize(grid[..., 0], 7, ac)
            iy = ops._unnormalize(grid[..., 1], 9, ac)
            out = ops.sample_bilinear(src, ix, iy, pad)
            assert_close(out, ref, 1e-12, 1e-12, 'sampler %s %s' % (pad, ac))
            go = torch.randn_like(ref)
            g1 = torch.autograd.grad(ref, [src, grid], go)
            g2 = torch.autograd.grad(out, [src, grid], go)
            assert_close(g2[0], g1[0], 1e-12, 1e-12, 'sampler dsrc')
            assert_close(g2[1], g1[1], 1e-11, 1e-11, 'sampler dgrid')


def test_correlation_gradcheck_float64():
    torch.manual_seed(1)
    x1 = torch.randn(1, 3, 5, 6, dtype=torch.float64, requires_grad=True)
    x2 = torch.randn(1, 3, 5, 6, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda a, b: ops.correlation(a, b, 2), (x1, x2), eps=1e-6, atol=1e-8)


def test_general_parameter_space_oracle_vs_reference(golden):
    """Parameter values no shipped config uses (tests/golden/general.npz, frozen from the reference):
    flow_warp(mode='nearest' | 'bicubic'), SSIM(md=2,3), TernaryLoss(max_distance=4,5)."""
    from oracle import ops as O
    g = golden('general')
    for name in g['wnames']:
        for pad in ('zeros', 'border'):
            for ac in (True, False):
                tag = '%s_%s_%d' % (name, pad, int(ac))
                x = g[name + '_x'].requires_grad_(True)
                y = O.flow_warp(x, g[name + '_flow'], pad=pad, mode='nearest', align_corners=ac)
                assert_close(y, g[tag + '_y'], 0, 0, tag)
                gx, = torch.autograd.grad(y, [x], g[name + '_g'])
                assert_close(gx, g[tag + '_gx'], 1e-6, 1e-6, tag + ' gx')
                # mode='bicubic': the oracle's restatement of ATen's bicubic sampler against the reference's outputs
                xb, fb = g[name + '_x'].clone().requires_grad_(True), g[name + '_flow'].clone().requires_grad_(True)
                yb = O.flow_warp(xb, fb, pad=pad, mode='bicubic', align_corners=ac)
                mx = float(g[name + '_x'].abs().max())
                assert_close(yb, g[tag + '_cub_y'], 4e-6 * mx, 1e-5, tag + ' bicubic')
                gxb, gfb = torch.autograd.grad(yb, [xb, fb], g[name + '_g'])
                assert_close(gxb, g[tag + '_cub_gx'], 1e-5, 1e-5, tag + ' bicubic gx')
                assert_close(gfb, g[tag + '_cub_gf'], 1e-5 * (1 + float(g[tag + '_cub_gf'].abs().max())), 1e-4, tag + ' bicubic gflow')
    for md in (2, 3):
        assert_close(O.ssim(g['im1'], g['im2'], md), g['ssim%d' % md], 1e-6, 1e-6, 'ssim md')
    for md, sd in ((4, True), (5, False)):
        tag = 'tern%d_%d' % (md, int(sd))
        d, m = O.ternary_loss(g['im1'], g['im2'], md, sd)
        assert_close(d, g[tag + '_dist'], 1e-6, 1e-6, tag)
        assert_close(m, g[tag + '_mask'], 0, 0, tag + ' mask')


def test_general_correlation_oracle_reduces_to_the_pinned_default():
    """oracle.correlation_general (restated from correlation_cuda_kernel.cu:41-114; parity unpinned for non-default
    parameters) with (pad=d, kernel=1, strides 1) equals the pinned correlation_native restatement."""
    from oracle import ops as O
    gen = torch.Generator().manual_seed(3)
    x1, x2 = torch.randn(2, 5, 9, 11, generator=gen), torch.randn(2, 5, 9, 11, generator=gen)
    assert_close(O.correlation_general(x1, x2, 4, 1, 4, 1, 1), O.correlation(x1, x2, 4), 1e-7, 1e-6, 'default')
    y = O.correlation_general(x1, x2, 3, 3, 2, 2, 2)  # FlowNetC-like: pad 3, 3x3 kernel, d 2, strides 2
    assert y.shape == (2, 9, 5, 6)  # ceil((9+6-6)/2), ceil((11+6-6)/2); (2*(2//2)+1)^2 channels
