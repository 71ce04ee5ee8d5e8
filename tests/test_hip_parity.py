"""GPU parity: the gfx950 kernels (through the C ABI / autograd Functions) against
 (1) the golden vectors frozen from the reference (tests/golden/*.npz) and
 (2) the CPU oracle on seeded inputs at sizes the oracle finishes in seconds.

Stated fp32 tolerances: correlation fwd/bwd atol 1e-6 (5e-6 with N(0,1) output
gradients: 81 O(1) terms) rtol 1e-5; warp fwd atol (2e-6 + 4 ulp(coordinate)) * max|x|; warp bwd and
splat maps (fp32 atomics, order-dependent) atol 1e-5 rtol 1e-4; scalar losses rtol 1e-5.
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


@pytest.fixture(scope='module')
def AF():
    from arflow_amd import functional
    return functional


@pytest.fixture(scope='module')
def oracle():
    from oracle import ops
    return ops


def cu(t):
    return t.cuda()


# ------------------------------------------------------------------------------------------------
def test_library_loaded_and_no_fallback():
    from arflow_amd import _lib, functional
    lib = _lib.load()
    assert lib.arflow_abi_version() == _lib.ABI_VERSION
    with pytest.raises(_lib.ArflowHipError):
        functional.correlation(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4))  # CPU tensor must raise


def test_correlation_golden(golden, AF):
    g = golden('corr')
    for name in g.names():
        x1 = cu(g[name + '_x1']).requires_grad_(True)
        x2 = cu(g[name + '_x2']).requires_grad_(True)
        d = int(g[name + '_d'])
        y = AF.correlation(x1, x2, d)
        assert_close(y, g[name + '_y'], 1e-6, 1e-5, name + ' fwd')
        gx1, gx2 = torch.autograd.grad(y, [x1, x2], cu(g[name + '_g']))
        assert_close(gx1, g[name + '_gx1'], 5e-6, 1e-5, name + ' gx1')
        assert_close(gx2, g[name + '_gx2'], 5e-6, 1e-5, name + ' gx2')


@pytest.mark.parametrize('shape', [(2, 32, 24, 40, 4), (1, 192, 6, 10, 4), (2, 96, 12, 20, 4), (1, 7, 33, 65, 4),
                                   (1, 64, 48, 80, 4), (3, 5, 17, 31, 2), (1, 3, 9, 9, 6), (2, 16, 96, 160, 4)])
def test_correlation_vs_oracle(AF, oracle, shape):
    B, C, H, W, d = shape
    gen = torch.Generator().manual_seed(B * 1000 + C)
    x1 = torch.randn(B, C, H, W, generator=gen)
    x2 = torch.randn(B, C, H, W, generator=gen)
    go = torch.randn(B, (2 * d + 1) ** 2, H, W, generator=gen)
    ref = oracle.correlation(x1, x2, d)
    r1, r2 = oracle.correlation_backward(go, x1, x2, d)
    a = cu(x1).requires_grad_(True)
    b = cu(x2).requires_grad_(True)
    y = AF.correlation(a, b, d)
    assert_close(y, ref, 1e-6, 1e-5, 'corr fwd %s' % (shape,))
    g1, g2 = torch.autograd.grad(y, [a, b], cu(go))
    assert_close(g1, r1, 5e-6, 1e-5, 'corr gx1 %s' % (shape,))
    assert_close(g2, r2, 5e-6, 1e-5, 'corr gx2 %s' % (shape,))
    # only one gradient requested
    a2 = cu(x1).requires_grad_(True)
    y2 = AF.correlation(a2, cu(x2), d)
    g1b, = torch.autograd.grad(y2, [a2], cu(go))
    assert_close(g1b, r1, 5e-6, 1e-5, 'corr gx1-only')


def test_correlation_fused_leaky_relu(AF, oracle):
    """corr + LeakyReLU(0.1) fused in the kernel == leaky_relu(oracle corr), forward and both gradients."""
    gen = torch.Generator().manual_seed(77)
    # fast path with sign words (aligned, ragged tiles, several tile columns), padded width, generic path
    for B, C, H, W in ((2, 32, 24, 40), (1, 8, 20, 36), (2, 4, 9, 68), (1, 8, 6, 10), (1, 5, 9, 11)):
        x1 = torch.randn(B, C, H, W, generator=gen)
        x2 = torch.randn(B, C, H, W, generator=gen)
        go = torch.randn(B, 81, H, W, generator=gen)
        a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        ref = torch.nn.functional.leaky_relu(oracle.correlation(a, b, 4), 0.1)
        r1, r2 = torch.autograd.grad(ref, [a, b], go)
        ac, bc = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
        y = AF.correlation(ac, bc, 4, negative_slope=0.1)
        assert_close(y, ref, 1e-6, 1e-5, 'fused leaky fwd')
        g1, g2 = torch.autograd.grad(y, [ac, bc], cu(go))
        # a pre-activation within rounding of 0 may pick the other branch: compare away from it
        assert_close(g1, r1, 1e-5, 1e-4, 'fused leaky gx1')
        assert_close(g2, r2, 5e-6, 5e-5, 'fused leaky gx2')


def test_correlation_bwd_derivative_from_output(AF, oracle):
    """arflow_corr_bwd on the fast path WITHOUT sign words: the LeakyReLU derivative is taken from the
    sign of the forward output (`out` argument), like torch's in-place leaky_relu backward."""
    from arflow_amd import _lib
    lib = _lib.load()
    gen = torch.Generator().manual_seed(78)
    B, C, H, W = 2, 8, 20, 36
    x1, x2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, H, W, generator=gen)
    go = torch.randn(B, 81, H, W, generator=gen)
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    ref = torch.nn.functional.leaky_relu(oracle.correlation(a, b, 4), 0.1)
    r1, r2 = torch.autograd.grad(ref, [a, b], go)
    x1c, x2c, goc = cu(x1), cu(x2), cu(go)
    out = torch.empty(B, 81, H, W, device='cuda')
    g1, g2 = torch.empty_like(x1c), torch.empty_like(x2c)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.arflow_corr_sign_planes(C, W, 4) == 3
    _lib.check(lib.arflow_corr_fwd(x1c.data_ptr(), x2c.data_ptr(), out.data_ptr(), None, B, C, H, W, 4, 0.1, st), 'fwd')
    _lib.check(lib.arflow_corr_bwd(goc.data_ptr(), out.data_ptr(), None, x1c.data_ptr(), x2c.data_ptr(), g1.data_ptr(),
                                   g2.data_ptr(), B, C, H, W, 4, 0.1, st), 'bwd')
    assert_close(out, ref, 1e-6, 1e-5, 'fwd')
    assert_close(g1, r1, 5e-6, 5e-5, 'gx1 via out')
    assert_close(g2, r2, 5e-6, 5e-5, 'gx2 via out')


def test_feature_normalisation_golden(golden, AF):
    """Both normalize_features variants against the reference's outputs (tests/golden/aux.npz)."""
    g = golden('aux')
    a, b = AF.normalize_pair(cu(g['f1']), cu(g['f2']), 'joint')
    assert_close(a, g['nj_1'], 4e-7, 2e-6, 'joint norm 1')
    assert_close(b, g['nj_2'], 4e-7, 2e-6, 'joint norm 2')
    c, d = AF.normalize_pair(cu(g['f1']), cu(g['f2']), 'avg')
    assert_close(c, g['nu_1'], 4e-7, 2e-6, 'uflow norm 1')
    assert_close(d, g['nu_2'], 4e-7, 2e-6, 'uflow norm 2')


@pytest.mark.parametrize('mode', ['joint', 'avg'])
@pytest.mark.parametrize('shape,offset', [((3, 32, 24, 40), 0.0), ((2, 8, 9, 11), 0.3), ((2, 32, 48, 80), 25.0),
                                          ((1, 3, 1, 2), -1.0)])
def test_feature_normalisation_vs_oracle(AF, oracle, mode, shape, offset):
    """Forward and both gradients against the oracle (torch autograd of the reference expression), on
    aligned / unaligned sizes, and with a mean 25x the spread (the moment sums must not cancel)."""
    gen = torch.Generator().manual_seed(21)
    x1 = torch.randn(*shape, generator=gen) + offset
    x2 = 0.5 * torch.randn(*shape, generator=gen) + 1.5 * offset
    g1, g2 = torch.randn(*shape, generator=gen), torch.randn(*shape, generator=gen)
    fn = oracle.normalize_features_joint if mode == 'joint' else oracle.normalize_features_uflow
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    r1, r2 = fn([a, b])
    ra, rb = torch.autograd.grad([r1, r2], [a, b], [g1, g2])
    ac, bc = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
    y1, y2 = AF.normalize_pair(ac, bc, mode)
    # (x - mu) keeps the input's absolute rounding: atol scales with |x| / std
    tol = 2e-6 * (1.0 + abs(offset) * 4)
    assert_close(y1, r1, (tol * 4) / 2, 5e-6, 'y1')
    assert_close(y2, r2, (tol * 4) / 2, 5e-6, 'y2')
    da, db = torch.autograd.grad([y1, y2], [ac, bc], [cu(g1), cu(g2)])
    gt = max(float(ra.abs().max()), float(rb.abs().max()))
    assert_close(da, ra, (2e-5 * gt * (1.0 + abs(offset))) / 20, 5e-6, 'gx1')
    assert_close(db, rb, (2e-5 * gt * (1.0 + abs(offset))) / 20, 5e-6, 'gx2')
    # one-sided gradient request
    ac2 = cu(x1).requires_grad_(True)
    y1b, y2b = AF.normalize_pair(ac2, cu(x2), mode)
    da2, = torch.autograd.grad([y1b, y2b], [ac2], [cu(g1), cu(g2)])
    assert_close(da2, ra, (2e-5 * gt * (1.0 + abs(offset))) / 20, 5e-6, 'gx1 only')


@pytest.mark.parametrize('shape', [(2, 16, 24, 40), (3, 5, 3, 5), (1, 128, 6, 10), (2, 3, 96, 160)])
def test_bias_leaky_relu(AF, shape):
    """Fused conv epilogue: leaky_relu(x + bias) in place, gradient w.r.t. x and the bias."""
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(*shape, generator=gen)
    bias = torch.randn(shape[1], generator=gen)
    go = torch.randn(*shape, generator=gen)
    a, bb = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = torch.nn.functional.leaky_relu(a + bb.view(1, -1, 1, 1), 0.1)
    ra, rb = torch.autograd.grad(ref, [a, bb], go)
    xc, bc = cu(x).requires_grad_(True), cu(bias).requires_grad_(True)
    y = AF.bias_leaky_relu(xc * 1.0, bc, 0.1)  # * 1.0: the op works in place on a non-leaf
    assert_close(y, ref, 1e-7, 1e-7, 'fwd')
    ga, gb = torch.autograd.grad(y, [xc, bc], cu(go))
    assert_close(ga, ra, 1e-7, 1e-7, 'gx')
    assert_close(gb, rb, (1e-5 * max(1.0, float(rb.abs().max()))) / 2, 5e-6, 'gbias')


def test_correlation_module_signature(AF):
    from arflow_amd.correlation import Correlation, compute_cost_volume
    m = Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)
    x = torch.randn(1, 8, 12, 20, device='cuda')
    assert m(x, x).shape == (1, 81, 12, 20)
    # the CUDA extension's other parameters (FlowNetC's set) run the general kernels with its output geometry
    # (correlation_cuda.cc:31-34): ((20/2)*2+1)^2 channels, ceil((12 + 40 - 40) / 1) x ceil((20 + 40 - 40) / 1)
    g = Correlation(pad_size=20, kernel_size=1, max_displacement=20, stride1=1, stride2=2)
    assert g(x, x).shape == (1, 441, 12, 20)
    with pytest.raises(ValueError):
        Correlation(pad_size=3, kernel_size=2, max_displacement=4)  # even kernel sizes do not exist in the extension
    with pytest.raises(ValueError):
        compute_cost_volume(x[:, :, :4], x[:, :, :4], 4)


def _warp_tol(x, H, W):
    ulp = 2.0 ** -23 * max(H, W)
    return (2e-6 + 4 * ulp) * float(x.abs().max())


@pytest.mark.parametrize('pad', ['zeros', 'border'])
def test_warp_backward_adversarial_fields(oracle, pad):
    """d/d src and d/d flow of the warp on fields that stress the scatter -> per-cell-list conversion:
    smooth, violently noisy (windows beyond the LDS budget: direct-atomic fallback), all pixels converging
    onto a few source cells (1000-entry lists), everything leaving the image (empty tiles)."""
    from arflow_amd.warp_utils import flow_warp
    gen = torch.Generator().manual_seed(5)
    B, C, H, W = 2, 5, 21, 70
    src = torch.randn(B, C, H, W, generator=gen)
    go = torch.randn(B, C, H, W, generator=gen)
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    fields = {
        'smooth': torch.nn.functional.interpolate(3 * torch.randn(B, 2, 3, 5, generator=gen), (H, W), mode='bilinear',
                                                  align_corners=True),
        'noise': 9 * torch.randn(B, 2, H, W, generator=gen),
        'huge noise': 60 * torch.randn(B, 2, H, W, generator=gen),
        'converging': (torch.stack([W / 2 - xs, H / 2 - ys]) * 0.96).expand(B, 2, H, W).clone(),
        'leaving': torch.full((B, 2, H, W), 500.0),
    }
    for name, flow in fields.items():
        s_ref, f_ref = src.clone().requires_grad_(True), flow.clone().requires_grad_(True)
        r_s, r_f = torch.autograd.grad(oracle.flow_warp(s_ref, f_ref, pad=pad, align_corners=True), [s_ref, f_ref], go)
        sc, fc = cu(src).requires_grad_(True), cu(flow).requires_grad_(True)
        gs, gf = torch.autograd.grad(flow_warp(sc, fc, pad=pad, align_corners=True), [sc, fc], cu(go))
        assert_close(gs, r_s, 1e-5 * max(1.0, float(r_s.abs().max())), 1e-4, '%s %s gsrc' % (name, pad))
        assert_close(gf, r_f, (1e-5 * max(1.0, float(r_f.abs().max()))) / 10, 1e-5, '%s %s gflow' % (name, pad))


def test_flow_warp_golden(golden):
    from arflow_amd.warp_utils import flow_warp
    g = golden('warp')
    for name in g.names():
        for pad in ('zeros', 'border'):
            for ac in (True, False):
                x = cu(g[name + '_x']).requires_grad_(True)
                fl = cu(g[name + '_flow']).requires_grad_(True)
                tag = '%s_%s_%d' % (name, pad, int(ac))
                y = flow_warp(x, fl, pad=pad, align_corners=ac)
                assert_close(y, g[tag + '_y'], _warp_tol(x.detach(), *x.shape[2:]), 1e-5, tag + ' fwd')
                gx, gf = torch.autograd.grad(y, [x, fl], cu(g[name + '_g']))
                assert_close(gx, g[tag + '_gx'], 1e-5, 1e-4, tag + ' gx')
                if name != 'integer':  # at exactly-integer coordinates d/dflow is one-sided; see next test
                    assert_close(gf, g[tag + '_gf'], 2e-5, 5e-5, tag + ' gflow')


def test_resample_family_golden(golden):
    from arflow_amd import uflow_utils as U, uflow_resampler as R
    g = golden('warp')
    for name in g.names():
        x = cu(g[name + '_x']).requires_grad_(True)
        fl = cu(g[name + '_flow']).requires_grad_(True)
        coords = U.flow_to_warp(fl)
        assert_close(coords, g[name + '_coords'], 0, 0, name + ' flow_to_warp')
        assert_close(U.mask_invalid(coords), g[name + '_mask_invalid'], 0, 0, name + ' mask_invalid')
        assert_close(U.mask_invalid_flow(fl), g[name + '_mask_invalid'], 0, 0, name + ' mask_invalid_flow')
        tol = _warp_tol(x.detach(), *x.shape[2:])
        y = U.resample(x, coords)
        assert_close(y, g[name + '_resample_y'], (tol) / 10, 1e-6, name + ' resample')
        gx, gf = torch.autograd.grad(y, [x, fl], cu(g[name + '_g']))
        assert_close(gx, g[name + '_resample_gx'], 5e-7, 5e-6, name + ' resample gx')
        if name != 'integer':
            assert_close(gf, g[name + '_resample_gf'], 3e-6, 1e-5, name + ' resample gflow')
        y2 = U.resample_flow(x, fl)
        assert_close(y2, g[name + '_resample_y'], (tol) / 10, 1e-6, name + ' resample_flow')
        nhwc = R.resampler(x.detach().permute(0, 2, 3, 1).contiguous(), coords.detach().permute(0, 2, 3, 1).contiguous())
        assert_close(nhwc, g[name + '_resampler_nhwc'], tol + 2e-6, 1e-5, name + ' resampler nhwc')


@pytest.mark.parametrize('cfg', [(2, 32, 48, 80, 'zeros', True), (1, 64, 24, 40, 'border', True),
                                 (2, 3, 96, 160, 'border', False), (1, 16, 31, 57, 'zeros', False)])
def test_flow_warp_vs_oracle(AF, oracle, cfg):
    B, C, H, W, pad, ac = cfg
    gen = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, C, H, W, generator=gen)
    fl = 3.0 * torch.randn(B, 2, H, W, generator=gen)
    go = torch.randn(B, C, H, W, generator=gen)
    xr, fr = x.clone().requires_grad_(True), fl.clone().requires_grad_(True)
    ref = oracle.flow_warp(xr, fr, pad=pad, align_corners=ac)
    rgx, rgf = torch.autograd.grad(ref, [xr, fr], go)
    a, f = cu(x).requires_grad_(True), cu(fl).requires_grad_(True)
    y = AF.warp(a, f, pad=pad, align_corners=ac)
    assert_close(y, ref, (_warp_tol(x, H, W)) / 50, 2e-7, 'warp fwd')
    gx, gf = torch.autograd.grad(y, [a, f], cu(go))
    assert_close(gx, rgx, 4e-6, 2e-5, 'warp gx')
    # d/dflow multiplies sums over C channels by up to W/2: scale the absolute tolerance with it
    assert_close(gf, rgf, (1e-5 * (C ** 0.5) * float(x.abs().max()) * 4) / 20, 1e-5, 'warp gflow')


def test_warp_strided_flow_and_flow_only_grad(AF, oracle):
    """A [B,4,H,W] (fw,bw) tensor is consumed in place through the batch stride, and a detached
    source yields only d/dflow (the loss-side call, losses/uflow_loss.py:31)."""
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 20, 28, generator=gen)
    f4 = 2 * torch.randn(2, 4, 20, 28, generator=gen)
    for sl in (slice(0, 2), slice(2, 4)):
        fr = f4[:, sl].clone().requires_grad_(True)
        ref = oracle.resample(x, oracle.flow_to_warp(fr))
        rg, = torch.autograd.grad(ref, [fr], torch.ones_like(ref))
        f4c = cu(f4).requires_grad_(True)
        y = AF.warp(cu(x), f4c[:, sl], pad='zeros', align_corners=True, norm=AF.NORM_UFLOW)
        assert_close(y, ref, 1e-6, 1e-6, 'strided warp')
        g, = torch.autograd.grad(y, [f4c], torch.ones_like(y))
        assert_close(g[:, sl], rg, 5e-6, 1e-5, 'strided gflow')
        other = slice(2, 4) if sl.start == 0 else slice(0, 2)
        assert float(g[:, other].abs().max()) == 0.0


def test_masks_golden(golden):
    from arflow_amd import warp_utils as WU, uflow_utils as U
    g = golden('masks')
    for name in g.names():
        fl = cu(g[name + '_flow'])
        assert_close(U.compute_range_map(fl), g[name + '_range_map'], 2e-6, 2e-5, name + ' range map')
        assert_close(WU.compute_range_map(fl), g[name + '_range_map_wu'], 2e-6, 2e-5, name + ' range map wu')
        coords = U.flow_to_warp(fl)
        assert_close(WU.get_corresponding_map(coords), g[name + '_corr_map'], 2e-6, 2e-5, name + ' corr map')
        assert_close(WU.get_occu_mask_backward(fl, th=0.0), g[name + '_occ_back_0'], 5e-6, 5e-5, name + ' occ soft')
        # thresholded masks: compare where the reference's soft value is not within 1e-5 of the threshold
        soft = 1.0 - g[name + '_occ_back_0']
        safe = (soft - 0.2).abs() > 1e-5
        got = WU.get_occu_mask_backward(fl, th=0.2).cpu()
        assert torch.equal(got[safe], g[name + '_occ_back_02'][safe]), name + ' occ back'
        assert_close(WU.border_mask(fl), g[name + '_border_mask'], 0, 0, name + ' border mask')
        for key, other in (('_occ_bidir', -0.7 * fl.flip(-1)), ('_occ_bidir_neg', -fl)):
            other = other.contiguous()
            got = WU.get_occu_mask_bidirection(fl, other).cpu()
            ref = g[name + key]
            # thresholded: |f12 + f21w|^2 > 0.01 (|f12|^2 + |f21w|^2) + 0.5 (utils/warp_utils.py:93-100).  Compare
            # exactly wherever the oracle's margin to the threshold exceeds the fp32 noise of the warp (1e-4
            # relative), as for the range-map masks above; inside that band either side is right.
            from oracle import ops as O_
            f12, f21 = fl.cpu(), other.cpu()
            f21w = O_.flow_warp(f21, f12, pad='zeros')
            lhs = ((f12 + f21w) ** 2).sum(1, keepdim=True)
            rhs = 0.01 * ((f12 ** 2).sum(1, keepdim=True) + (f21w ** 2).sum(1, keepdim=True)) + 0.5
            safe = (lhs - rhs).abs() > 1e-4 * (lhs + rhs)
            assert float(safe.float().mean()) > 0.97, name + key
            assert torch.equal(got[safe], ref[safe]), name + key


def test_photo_blocks_golden(golden):
    from arflow_amd import loss_blocks as LB, uflow_utils as U
    g = golden('photo')
    for name in g.names():
        im1, im2, fl, mask = (cu(g[name + k]) for k in ('_im1', '_im2', '_flow', '_mask'))
        a, b = im1.clone().requires_grad_(True), im2.clone().requires_grad_(True)
        y = LB.SSIM(a, b)
        # sigma = E[x^2]-mu^2 cancels against C2 = 9e-4: fp32 summation-order noise is amplified ~1e3x
        assert_close(y, g[name + '_ssim'], 5e-6, 1e-6, name + ' ssim')
        ga, gb = torch.autograd.grad(y, [a, b], cu(g[name + '_ssim_g']))
        # relative to the gradient's scale: 2e-4 of max|ref| (the window variances cancel against C2 = 9e-4, which
        # amplifies fp32 summation-order noise ~1e2 on the flattest windows) + 1e-3 of the element
        for got_, key in ((ga, '_ssim_ga'), (gb, '_ssim_gb')):
            ref_ = g[name + key]
            assert_close(got_, ref_, (2e-4 * float(ref_.abs().max())) / 2, 0.0005, name + key)
        for md, sd in ((1, False), (3, True)):
            tag = '%s_ternary_%d_%d' % (name, md, int(sd))
            a, b = im1.clone().requires_grad_(True), im2.clone().requires_grad_(True)
            dist, tm = LB.TernaryLoss(a, b, md, sd)
            assert_close(dist, g[tag + '_dist'], 2e-5, 2e-5, tag)
            assert_close(tm, g[tag + '_mask'], 0, 0, tag + ' mask')
            ga, gb = torch.autograd.grad(dist, [a, b], cu(g[tag + '_g']))
            for got_, key in ((ga, '_ga'), (gb, '_gb')):  # relative to the gradient's scale, not an absolute number
                ref_ = g[tag + key]
                assert_close(got_, ref_, (1e-4 * float(ref_.abs().max())) / 20, 1e-5, tag + key)
        for ps in (7, 3):
            b = im2.clone().requires_grad_(True)
            y = U.census_loss(im1, b, mask, ps)
            assert_close(y, g['%s_census_%d' % (name, ps)], 5e-7, 5e-6, name + ' census')
            gb, = torch.autograd.grad(y, [b])
            ref = g['%s_census_%d_gb' % (name, ps)]
            assert_close(gb, ref, (1e-6 + 1e-4 * float(ref.abs().max())) / 20, 5e-6, name + ' census gb')
        for fn, key in ((lambda f: LB.smooth_grad_1st(f, im1, 10.), 'sm1_abs'),
                        (lambda f: LB.smooth_grad_1st(f, im1, 10., penalty='uflow'), 'sm1_uflow'),
                        (lambda f: LB.smooth_grad_2nd(f, im1, 10.), 'sm2')):
            f = fl.clone().requires_grad_(True)
            y = fn(f)
            assert_close(y, g['%s_%s' % (name, key)], 2e-8, 2e-6, name + ' ' + key)
            gf, = torch.autograd.grad(y, [f])
            ref = g['%s_%s_gf' % (name, key)]
            assert_close(gf, ref, (1e-8 + 1e-5 * float(ref.abs().max())) / 10, 1e-5, name + ' ' + key + ' grad')


def test_resize_helpers_golden(golden, AF):
    g = golden('aux')
    assert_close(AF.down4(cu(g['img'])), g['down4'], 1e-7, 1e-6, 'down4')
    up = AF.up4_clamp_mul(cu(g['m']))
    assert_close(up, g['up4'], 2e-7, 2e-6, 'up4 (inputs already in [0,1])')


def _loss_cases():
    from tests.test_oracle_golden import _loss_cases as lc
    return lc()


@pytest.mark.parametrize('group', [0, 1, 2])
def test_loss_modules_golden(golden, group):
    from arflow_amd import losses as L
    from arflow_amd.config import AttrDict
    g = golden('losses')
    cases = _loss_cases()[group]
    cls = (L.UFlowLoss, L.unFlowLoss, L.FullResLoss)[group]
    img = cu(g['img'])
    for name, cfg in cases:
        flows = [cu(g['flow%d' % i]).requires_grad_(True) for i in range(5)]
        res = cls(AttrDict(cfg))(flows, img)
        assert_close(res[0], g[name + '_total'], 5e-8, 3e-6, name + ' total')
        assert_close(res[1], g[name + '_warp'], 5e-8, 3e-6, name + ' warp')
        assert_close(res[2], g[name + '_smooth'], 2e-8, 1e-6, name + ' smooth')
        assert_close(res[3], g[name + '_absflow'], 1e-7, 1e-6, name + ' |flow|')
        if len(res) > 4:
            assert_close(res[4], g[name + '_mask1'], 1e-6, 1e-5, name + ' mask1')
        grads = torch.autograd.grad(res[0], flows, allow_unused=True)
        for i, gi in enumerate(grads):
            ref = g['%s_g%d' % (name, i)]
            gi = gi if gi is not None else torch.zeros_like(ref).cuda()
            assert_close(gi, ref, 2e-7 + 2e-4 * float(ref.abs().max()), 2e-3, '%s dflow%d' % (name, i))


def test_census_vs_oracle_midsize(oracle):
    from arflow_amd import uflow_utils as U
    gen = torch.Generator().manual_seed(11)
    B, H, W = 2, 96, 160
    im1 = torch.rand(B, 3, H, W, generator=gen)
    im2 = (im1 + 0.1 * torch.randn(B, 3, H, W, generator=gen)).clamp(0, 1)
    mask = (torch.rand(B, 1, H, W, generator=gen) > 0.2).float()
    b = im2.clone().requires_grad_(True)
    ref = oracle.census_loss(im1, b, mask)
    rg, = torch.autograd.grad(ref, [b])
    bc = cu(im2).requires_grad_(True)
    y = U.census_loss(cu(im1), bc, cu(mask))
    assert_close(y, ref, 2e-7, 2e-6, 'census loss')
    gb, = torch.autograd.grad(y, [bc])
    assert_close(gb, rg, (1e-4 * float(rg.abs().max())) / 20, 5e-6, 'census grad')


def test_full_size_properties(AF):
    """BASELINE config-2 sizes (B=8, 96x160 C=32 and 384x640 images): size-independent properties."""
    gen = torch.Generator(device='cuda').manual_seed(3)
    x1 = torch.randn(8, 32, 96, 160, device='cuda', generator=gen)
    x2 = torch.randn(8, 32, 96, 160, device='cuda', generator=gen)
    y = AF.correlation(x1, x2, 4)
    # centre channel is the per-pixel mean product
    assert_close(y[:, 40], (x1 * x2).mean(1), 1e-6, 1e-5, 'centre channel')
    # shifting x2 by one pixel moves the volume one displacement channel
    x2s = torch.roll(x2, shifts=1, dims=3)
    ys = AF.correlation(x1, x2s, 4)
    assert_close(ys[:, 41, :, 8:-8], y[:, 40, :, 8:-8], 1e-7, 1e-6, 'shift equivariance')
    # linearity in x1
    y2 = AF.correlation(2.5 * x1, x2, 4)
    assert_close(y2, 2.5 * y, 1e-6, 1e-5, 'linearity')
    # <corr(x1,x2), g> == <x1, gx1> == <x2, gx2>  (adjoint identity)
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    go = torch.randn(8, 81, 96, 160, device='cuda', generator=gen)
    out = AF.correlation(a, b, 4)
    g1, g2 = torch.autograd.grad(out, [a, b], go)
    lhs = float((out.double() * go.double()).sum())
    assert abs(float((g1.double() * x1.double()).sum()) - lhs) <= 1e-4 * abs(lhs) + 1e-2
    assert abs(float((g2.double() * x2.double()).sum()) - lhs) <= 1e-4 * abs(lhs) + 1e-2
    # warp with zero flow is the identity (align_corners=True), full-resolution 3-channel image
    img = torch.rand(8, 3, 384, 640, device='cuda', generator=gen)
    z = torch.zeros(8, 2, 384, 640, device='cuda')
    # not exact in the reference either: x -> 2x/(W-1)-1 -> ((g+1)/2)(W-1) costs ~ulp(W) = 7.6e-5 at W=640
    # bound: (2 ulp(640) in x + 2 ulp(384) in y = 1.8e-4 px) x (image slope <= 1 per px for U[0,1) noise)
    assert_close(AF.warp(img, z, 'zeros', True, AF.NORM_UFLOW), img, 2e-4, 0, 'identity warp')
    # census(a, a) = 0.01^0.4 on the valid interior
    from arflow_amd import uflow_utils as U
    ones = torch.ones(8, 1, 384, 640, device='cuda')
    assert abs(float(U.census_loss(img, img, ones)) - 0.01 ** 0.4) < 1e-5
    # range map of zero flow is all ones
    assert_close(AF.splat_map(z, 0), ones, 0, 0, 'range map identity')


def test_determinism_of_atomic_free_kernels(AF):
    gen = torch.Generator(device='cuda').manual_seed(9)
    x1 = torch.randn(2, 32, 48, 80, device='cuda', generator=gen, requires_grad=True)
    x2 = torch.randn(2, 32, 48, 80, device='cuda', generator=gen, requires_grad=True)
    go = torch.randn(2, 81, 48, 80, device='cuda', generator=gen)
    outs = []
    for _ in range(2):
        y = AF.correlation(x1, x2, 4)
        outs.append((y.detach().clone(),) + tuple(t.clone() for t in torch.autograd.grad(y, [x1, x2], go)))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_mv_loss_vs_oracle():
    """BASELINE config 5 wiring (build-defined): product MvLoss on the HIP kernels vs the
    same composition of oracle functions, values and gradients w.r.t. every flow level."""
    from arflow_amd.config import AttrDict
    from arflow_amd.losses import MvLoss
    from oracle import losses as OL
    from oracle.fixture_common import synth_pair
    gen = torch.Generator().manual_seed(31)
    img6, _ = synth_pair(2, 64, 96, gen)
    img = torch.cat([img6, synth_pair(2, 64, 96, gen)[0][:, :3]], 1)
    sizes = [(64, 96), (32, 48), (16, 24), (8, 12)]
    f12 = [1.5 * torch.randn(2, 2, h, w, generator=gen) for h, w in sizes]
    f10 = [1.5 * torch.randn(2, 2, h, w, generator=gen) for h, w in sizes]
    cfg = AttrDict(w_l1=0.15, w_ssim=0.85, alpha=10, w_smooth=75.0, w_scales=[1.0, 1.0, 0.5, 1.0],
                   w_sm_scales=[1.0, 0.5, 0.0, 0.0])
    a12 = [f.clone().requires_grad_(True) for f in f12]
    a10 = [f.clone().requires_grad_(True) for f in f10]
    ref = OL.MvLoss(cfg)(a12, a10, img)
    rg = torch.autograd.grad(ref[0], a12 + a10)
    c12 = [cu(f).requires_grad_(True) for f in f12]
    c10 = [cu(f).requires_grad_(True) for f in f10]
    got = MvLoss(cfg)(c12, c10, cu(img))
    for k in range(4):
        assert_close(got[k], ref[k], 1e-8, 1e-6, 'mv loss term %d' % k)
    gg = torch.autograd.grad(got[0], c12 + c10)
    for a, b in zip(gg, rg):
        assert_close(a, b, (2e-7 + 2e-4 * float(b.abs().max())) / 10, 0.0002, 'mv dflow')


def test_random_shapes_fuzz(AF, oracle):
    """Seeded sweep over ragged shapes (odd sizes, one-row / one-column maps, sizes around the 8 x 32 and
    16 x 64 tile edges, widths with and without 16-byte alignment): forward and gradients of the main ops
    against the oracle, with the tolerances of the dedicated tests."""
    import random
    from arflow_amd.warp_utils import flow_warp
    from arflow_amd.uflow_utils import census_loss, compute_range_map
    from arflow_amd.loss_blocks import smooth_grad_1st
    rnd = random.Random(1234)
    gen = torch.Generator().manual_seed(99)
    sizes = [(1, 1), (1, 9), (9, 1), (2, 3), (7, 31), (8, 32), (9, 33), (16, 64), (17, 65), (23, 40), (5, 100)]
    for trial in range(14):
        H, W = sizes[trial % len(sizes)] if trial < len(sizes) else (rnd.randint(1, 40), rnd.randint(1, 90))
        B, C = rnd.randint(1, 3), rnd.choice([1, 3, 4, 8, 12, 32])
        d = 4 if H > 4 else max(1, H - 1)
        x1, x2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, H, W, generator=gen)
        tag = 'trial %d B%d C%d %dx%d' % (trial, B, C, H, W)
        # correlation (+ fused LeakyReLU) with gradients
        n = (2 * d + 1) ** 2
        go = torch.randn(B, n, H, W, generator=gen)
        a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        ref = torch.nn.functional.leaky_relu(oracle.correlation(a, b, d), 0.1)
        r1, r2 = torch.autograd.grad(ref, [a, b], go)
        ac, bc = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
        y = AF.correlation(ac, bc, d, negative_slope=0.1)
        assert_close(y, ref, 1e-6, 1e-5, tag + ' corr')
        g1, g2 = torch.autograd.grad(y, [ac, bc], cu(go))
        assert_close(g1, r1, 1e-5, 1e-4, tag + ' corr gx1')
        assert_close(g2, r2, 1e-5, 1e-4, tag + ' corr gx2')
        # warp with both gradients (a 1-pixel-wide map divides by W - 1 = 0 in the reference's norm_grid,
        # utils/warp_utils.py:16-23: NaN there, skipped here)
        flow = 2.5 * torch.randn(B, 2, H, W, generator=gen)
        gw = torch.randn(B, C, H, W, generator=gen)
        pad, ac_ = rnd.choice(['zeros', 'border']), rnd.choice([True, False])
        if min(H, W) == 1:
            continue
        s_ref, f_ref = x2.clone().requires_grad_(True), flow.clone().requires_grad_(True)
        wr = oracle.flow_warp(s_ref, f_ref, pad=pad, align_corners=ac_)
        rs, rf = torch.autograd.grad(wr, [s_ref, f_ref], gw)
        sc, fc = cu(x2).requires_grad_(True), cu(flow).requires_grad_(True)
        wy = flow_warp(sc, fc, pad=pad, align_corners=ac_)
        tol = (2e-6 + 4 * 2.0 ** -23 * max(H, W)) * max(1.0, float(x2.abs().max()))
        assert_close(wy, wr, (tol) / 20, 5e-7, tag + ' warp')
        gs, gf = torch.autograd.grad(wy, [sc, fc], cu(gw))
        assert_close(gs, rs, (1e-5 * max(1.0, float(rs.abs().max()))) / 20, 5e-6, tag + ' warp gsrc')
        assert_close(gf, rf, (2e-4 * max(1.0, float(rf.abs().max()))) / 200, 5e-6, tag + ' warp gflow')
        # splat map, feature normalisation
        assert_close(compute_range_map(cu(flow)), oracle.compute_range_map(flow), 2e-6, 2e-5, tag + ' range map')
        if C * H * W >= 2:
            ya, yb = AF.normalize_pair(cu(x1), cu(x2), 'joint')
            ra, rb = oracle.normalize_features_joint([x1, x2])
            assert_close(ya, ra, 2e-6, 2e-6, tag + ' norm')
            assert_close(yb, rb, 2e-6, 2e-6, tag + ' norm')
        # census loss and smoothness on 3-channel images (the reference's zero_mask_border,
        # utils/uflow_utils.py:234-238, breaks on maps smaller than its 3-pixel border: H, W >= 7 only)
        im1, im2 = torch.rand(B, 3, H, W, generator=gen), torch.rand(B, 3, H, W, generator=gen)
        mask = (torch.rand(B, 1, H, W, generator=gen) > 0.2).float()
        if H >= 7 and W >= 7:
            i2 = im2.clone().requires_grad_(True)
            lr = oracle.census_loss(im1, i2, mask)
            i2c = cu(im2).requires_grad_(True)
            lc = census_loss(cu(im1), i2c, cu(mask))
            assert abs(float(lc) - float(lr)) <= 5e-5 * abs(float(lr)) + 1e-6, (tag, float(lc), float(lr))
            if float(mask[:, :, 3:-3, 3:-3].sum()) > 0:
                gr, = torch.autograd.grad(lr, i2)
                gc, = torch.autograd.grad(lc, i2c)
                assert_close(gc, gr, (2e-5 * max(1e-3, float(gr.abs().max()))) / 5, 4e-5, tag + ' census grad')
        if H >= 2 and W >= 2:
            fl = flow.clone().requires_grad_(True)
            sr = oracle.smooth_grad_1st(fl, im1, 10.0)
            flc = cu(flow).requires_grad_(True)
            sc_ = smooth_grad_1st(flc, cu(im1), 10.0)
            assert abs(float(sc_) - float(sr)) <= 2e-5 * abs(float(sr)) + 1e-7, (tag, float(sc_), float(sr))
            gsr, = torch.autograd.grad(sr, fl)
            gsc, = torch.autograd.grad(sc_, flc)
            assert_close(gsc, gsr, (1e-6 + 1e-5 * float(gsr.abs().max())) / 100, 1e-6, tag + ' smooth grad')


@pytest.mark.parametrize('family', ['column', 'ordered', 'pair-symmetric'])
@pytest.mark.parametrize('size', [(2, 48, 64), (1, 16, 64), (3, 40, 132), (2, 96, 160), (1, 100, 236)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_fused_census_warp_vs_unfused_path_and_oracle(AF, oracle, size, family, monkeypatch):
    """arflow_census_warp_fwd/bwd (warp + validity + x4 mask upsample + census loss in one launch each way, on the
    grey planes of arflow_down4_gray) against the three separate launches it replaces and against the oracle's
    composition of the reference functions (losses/uflow_loss.py:30-54): mask bit-identical, loss and flow
    gradient at the census tolerances (grey-then-sample re-associates fp32 sums)."""
    from arflow_amd import uflow_utils as U
    # read by the library at every call: all three kernel families are tested (column = the default, census_col.hip; ordered
    # = the 4-pixels-per-lane kernels of census_warp.hip; pair-symmetric = census_sym.hip)
    monkeypatch.setenv('ARFLOW_CENSUS_SYM', '1' if family == 'pair-symmetric' else '0')
    monkeypatch.setenv('ARFLOW_CENSUS_COL', '0' if family == 'ordered' else '1')
    B, H, W = size
    gen = torch.Generator().manual_seed(H + W)
    im1, im2 = torch.rand(B, 3, H, W, generator=gen), torch.rand(B, 3, H, W, generator=gen)
    flow = 4.0 * torch.randn(B, 2, H, W, generator=gen)
    flow[:, :, :2] += 30.0  # some samples leave the image: validity mask, zero padding
    occ = 1.6 * torch.rand(B, 1, H // 4, W // 4, generator=gen) - 0.2  # exercises the clamp on both sides
    # grey planes + x1/4 copies
    small1, gray1 = AF.down4_gray(cu(im1))
    _, gray2 = AF.down4_gray(cu(im2), want_small=False)
    assert torch.equal(small1, AF.down4(cu(im1)))
    assert_close(gray1, oracle.rgb_to_grayscale(im1) * 255, 0, 0, 'grey plane')
    # unfused product path
    f1 = cu(flow).requires_grad_(True)
    rec, valid = AF.warp_with_valid(cu(im2), f1, pad='zeros', align_corners=True, norm=AF.NORM_UFLOW)
    mask1 = AF.up4_clamp_mul(cu(occ), valid)
    l1 = U.census_loss(cu(im1), rec, mask1)
    g1, = torch.autograd.grad(l1, [f1])
    # fused
    f2 = cu(flow).requires_grad_(True)
    l2, mask2 = AF.census_warp_loss(gray1, gray2, f2, cu(occ), 7)
    g2, = torch.autograd.grad(l2, [f2])
    assert torch.equal(mask1, mask2), 'mask differs from the unfused path'
    assert_close(l2, l1, 5e-7, 5e-6, 'loss vs the unfused path')
    assert_close(g2, g1, 1e-6 + 1e-4 * float(g1.abs().max()), 1e-3, 'flow gradient vs the unfused path')
    # oracle
    fr = flow.clone().requires_grad_(True)
    coords = oracle.flow_to_warp(fr)
    rmask = torch.nn.functional.interpolate(occ.clamp(0, 1), scale_factor=4, mode='bilinear', align_corners=False) * \
        oracle.mask_invalid(coords)
    lr = oracle.census_loss(im1, oracle.resample(im2, coords), rmask.detach())
    gr, = torch.autograd.grad(lr, [fr])
    assert_close(mask2, rmask, 1e-6, 1e-6, 'mask vs oracle')
    assert_close(l2, lr, 5e-7, 5e-6, 'loss vs oracle')
    assert_close(g2, gr, 1e-6 + 1e-4 * float(gr.abs().max()), 1e-3, 'flow gradient vs oracle')
    # without the occlusion term (occ_small = NULL): mask = validity only
    l3, mask3 = AF.census_warp_loss(gray1, gray2, cu(flow), None, 7)
    assert torch.equal(mask3, valid)


@pytest.mark.parametrize('occ_from_back', [True, False])
def test_unflow_loss_both_directions_stacked_equals_sequential(occ_from_back):
    """unFlowLoss with_bk: the pass over 2B stacked (pair, direction) samples (losses/flow_loss.py here, _forward_stacked)
    against the per-direction form (the reference's order, losses/flow_loss.py:60-114): same losses, masks and flow gradients."""
    from arflow_amd.config import AttrDict
    from arflow_amd.losses import unFlowLoss
    from oracle.fixture_common import synth_pair
    gen = torch.Generator().manual_seed(33)
    B, H, W = 3, 64, 96
    img = synth_pair(B, H, W, gen)[0].cuda()
    flows = [(3.0 / s) * torch.randn(B, 4, H // s, W // s, generator=gen).cuda() for s in (1, 4, 8)]
    res = {}
    for pair in (True, False):
        loss = unFlowLoss(AttrDict(w_l1=0.15, w_ssim=0.85, w_ternary=0.0, warp_pad='border', with_bk=True, smooth_2nd=True,
                                   occ_from_back=occ_from_back, alpha=10, w_smooth=75.0, w_scales=[1.0, 1.0, 1.0],
                                   w_sm_scales=[1.0, 0.0, 0.0]))
        loss.pair = pair
        f = [t.clone().requires_grad_(True) for t in flows]
        out = loss(f, img)
        g = torch.autograd.grad(out[0], f)
        res[pair] = ([o.detach() for o in out], g, [m.clone() for m in loss.pyramid_occu_mask1 + loss.pyramid_occu_mask2])
    for k, n in enumerate(['total', 'warp', 'smooth', '|flow|']):
        assert_close(res[True][0][k], res[False][0][k], 1e-7, 5e-6, 'stacked vs sequential ' + n)
    for a, b in zip(res[True][2], res[False][2]):
        assert torch.equal(a, b), 'occlusion masks'
    for a, b, n in zip(res[True][1], res[False][1], ('d flow0', 'd flow1', 'd flow2')):
        assert_close(a, b, 1e-6 * float(b.abs().max()) + 1e-12, 1e-4, n)


@pytest.mark.parametrize('size', [(1, 8, 8), (2, 12, 20), (1, 36, 68), (1, 8, 200), (1, 200, 8), (3, 64, 64), (1, 60, 124), (2, 32, 244)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_census_kernel_families_agree_on_odd_sizes(AF, size, monkeypatch):
    """The pair-shared column kernels (census_col.hip: 61-column x 32-row tiles, halo lanes, rolled pair loop) against the
    ordered-pair kernels (census_warp.hip) on sizes that do not divide into either tiling -- one tile, one column of tiles,
    images narrower than a wave -- for the three patch sizes: identical mask, loss and flow gradient up to the order of the
    48 additions per pixel."""
    B, H, W = size
    gen = torch.Generator().manual_seed(H * W)
    im1, im2 = cu(torch.rand(B, 3, H, W, generator=gen)), cu(torch.rand(B, 3, H, W, generator=gen))
    flow = cu(2.5 * torch.randn(B, 2, H, W, generator=gen))
    occ = cu(1.6 * torch.rand(B, 1, H // 4, W // 4, generator=gen) - 0.2)
    _, g1 = AF.down4_gray(im1)
    _, g2 = AF.down4_gray(im2, want_small=False)
    for patch in (3, 5, 7):
        res = {}
        for fam in ('1', '0'):
            monkeypatch.setenv('ARFLOW_CENSUS_COL', fam)
            f = flow.clone().requires_grad_(True)
            loss, mask = AF.census_warp_loss(g1, g2, f, occ, patch)
            g, = torch.autograd.grad(loss, [f])
            res[fam] = (loss.detach(), mask, g)
        assert torch.equal(res['1'][1], res['0'][1]), 'mask'
        assert_close(res['1'][0], res['0'][0], 1e-7, 1.5e-6, 'loss, patch %d' % patch)  # measured 2e-7 .. 7e-7 absolute on losses of 2 .. 3
        # measured: <= 8e-7 of max|g| (the order of the 48 additions per pixel); 10x over it
        assert_close(res['1'][2], res['0'][2], 3e-6 * float(res['0'][2].abs().max()) + 1e-12, 5e-6, 'flow gradient, patch %d' % patch)


@pytest.mark.parametrize('up_align', [True, False])
@pytest.mark.parametrize('shape,pad', [((2, 8, 12, 20), 'zeros'), ((3, 32, 24, 40), 'border'), ((2, 96, 12, 20), 'zeros'), ((1, 5, 6, 10), 'zeros')],
                         ids=lambda v: str(v))
def test_warp_with_fused_flow_upsample_vs_oracle(AF, oracle, shape, pad, up_align):
    """warp_up2 (arflow_level_warp_fwd + arflow_warp_bwd + arflow_up2_bwd: the x2 flow upsample of models/pwclite.py:178-179
    folded into the warp launch) against interpolate(flow * 2) + the oracle's flow_warp: warped map, upsampled flow and the
    gradients w.r.t. the source and the COARSE flow (with a second consumer of the upsampled flow, like the estimator)."""
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=gen)
    fc = 1.5 * torch.randn(B, 2, H // 2, W // 2, generator=gen)
    go, gf = torch.randn(B, C, H, W, generator=gen), torch.randn(B, 2, H, W, generator=gen)
    xr, fr = x.clone().requires_grad_(True), fc.clone().requires_grad_(True)
    up = torch.nn.functional.interpolate(fr * 2, scale_factor=2, mode='bilinear', align_corners=up_align)
    ref = oracle.flow_warp(xr, up, pad=pad)
    gxr, gfr = torch.autograd.grad([ref, up], [xr, fr], [go, gf])
    xc, fcu = cu(x).requires_grad_(True), cu(fc).requires_grad_(True)
    out, upc = AF.warp_up2(xc, fcu, pad=pad, align_corners=True, up_align=up_align)
    gx, gfl = torch.autograd.grad([out, upc], [xc, fcu], [cu(go), cu(gf)])
    mx = float(x.abs().max())
    assert_close(upc, up.detach(), 2e-6, 2e-6, 'upsampled flow')  # ATen's CPU kernel associates the 4 products differently
    assert_close(out, ref.detach(), (2e-6 + 4 * 1.2e-7 * max(H, W)) * mx, 1e-5, 'warped map')  # 2 ulp of the upsampled flow x the map's slope
    assert_close(gx, gxr, 1e-5, 1e-4, 'd src')
    assert_close(gfl, gfr, 3e-6 * (1 + float(gfr.abs().max())), 3e-5, 'd coarse flow')


@pytest.mark.parametrize('family', ['column', 'ordered'])
@pytest.mark.parametrize('patch', [3, 5])
def test_fused_census_warp_small_patches_vs_oracle(AF, oracle, patch, family, monkeypatch):
    """The fused photometric direction with the census patch sizes no shipped config uses (radius 1 and 2: the other
    instantiations of census_warp / census_col) against the oracle's composition of the reference functions
    (utils/uflow_utils.py:241-293 with patch_size)."""
    monkeypatch.setenv('ARFLOW_CENSUS_SYM', '0')
    monkeypatch.setenv('ARFLOW_CENSUS_COL', '0' if family == 'ordered' else '1')
    B, H, W = 2, 40, 132
    gen = torch.Generator().manual_seed(patch)
    im1, im2 = torch.rand(B, 3, H, W, generator=gen), torch.rand(B, 3, H, W, generator=gen)
    flow = 3.0 * torch.randn(B, 2, H, W, generator=gen)
    flow[:, :, :2] += 20.0
    occ = 1.6 * torch.rand(B, 1, H // 4, W // 4, generator=gen) - 0.2
    _, gray1 = AF.down4_gray(cu(im1))
    _, gray2 = AF.down4_gray(cu(im2), want_small=False)
    f2 = cu(flow).requires_grad_(True)
    l2, mask2 = AF.census_warp_loss(gray1, gray2, f2, cu(occ), patch)
    g2, = torch.autograd.grad(l2, [f2])
    fr = flow.clone().requires_grad_(True)
    coords = oracle.flow_to_warp(fr)
    rmask = torch.nn.functional.interpolate(occ.clamp(0, 1), scale_factor=4, mode='bilinear', align_corners=False) * \
        oracle.mask_invalid(coords)
    lr = oracle.census_loss(im1, oracle.resample(im2, coords), rmask.detach(), patch)
    gr, = torch.autograd.grad(lr, [fr])
    assert_close(mask2, rmask, 1e-6, 1e-6, 'mask vs oracle')
    assert_close(l2, lr, 5e-7, 5e-6, 'loss vs oracle')
    assert_close(g2, gr, 1e-6 + 1e-4 * float(gr.abs().max()), 1e-3, 'flow gradient vs oracle')


@pytest.mark.parametrize('shape,slope', [((2, 32, 24, 40), 0.1), ((2, 8, 20, 36), 1.0), ((3, 32, 12, 20), 0.1), ((1, 5, 9, 11), 0.1)],
                         ids=lambda v: str(v))
def test_correlation_concat_equals_cat_of_plain_op(AF, oracle, shape, slope):
    """correlation_concat (arflow_corr_fwd/bwd_strided: the volume written straight into / its gradient read straight
    from the decoder's concatenated tensor, models/pwclite_uflow.py:218-222) == torch.cat([before, corr, after]) of
    the plain op, bit for bit, values and every gradient; the last shape has no strided path (falls back to cat)."""
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x1, x2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, H, W, generator=gen)
    bf = torch.randn(B, 3, H, W, generator=gen)
    a1, a2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, 2, H, W, generator=gen)
    go = torch.randn(B, 3 + 81 + C + 2, H, W, generator=gen)

    def run(fused):
        t = [cu(v).requires_grad_(True) for v in (x1, x2, bf, a1, a2)]
        if fused:
            y = AF.correlation_concat(t[0], t[1], (t[2],), (t[3], t[4]), 4, slope)
        else:
            y = torch.cat([t[2], AF.correlation(t[0], t[1], 4, slope), t[3], t[4]], 1)
        return (y,) + torch.autograd.grad(y, t, cu(go))
    got, ref = run(True), run(False)
    for k, (a, b) in enumerate(zip(got, ref)):
        assert torch.equal(a, b), 'tensor %d differs from cat(plain op): max %g' % (k, float((a - b).abs().max()))
    # x1 is usually BOTH a correlation input and a member of the concatenation (models/pwclite_uflow.py:220)
    t1, t2 = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
    y = AF.correlation_concat(t1, t2, (), (t1,), 4, slope)
    g1, g2 = torch.autograd.grad(y, [t1, t2], cu(go[:, :81 + C]))
    r1, r2 = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
    yr = torch.cat([torch.nn.functional.leaky_relu(oracle.correlation(r1.cpu(), r2.cpu(), 4), slope).cuda(), r1], 1)
    assert_close(y, yr, 1e-6, 1e-5, 'concat with shared x1')
    q1, q2 = torch.autograd.grad(yr, [r1, r2], cu(go[:, :81 + C]))
    assert_close(g1, q1, 1e-5, 1e-4, 'gx1 (corr + identity paths)')
    assert_close(g2, q2, 5e-6, 5e-5, 'gx2')


@pytest.mark.parametrize('shape', [(2, 32, 24, 40), (1, 7, 17, 33), (16, 32, 96, 160)], ids=lambda s: 'x'.join(map(str, s)))
@pytest.mark.parametrize('slope', [1.0, 0.1])
def test_bf16_storage_correlation(AF, oracle, shape, slope):
    """Opt-in bf16 STORAGE of the correlation inputs (the reference's native path dispatches
    half too, correlation_cuda_kernel.cu:352,369): fp32 accumulation and outputs, so against the oracle run on the
    bf16-ROUNDED inputs the fp32 tolerances of the fp32 path apply -- the only difference is what is stored."""
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(sum(shape))
    x1, x2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, H, W, generator=gen)
    go = torch.randn(B, 81, H, W, generator=gen)
    r1, r2 = x1.bfloat16().float(), x2.bfloat16().float()  # what the kernels see
    pre = oracle.correlation(r1, r2, 4)
    if slope != 1.0:
        go = go * (pre.abs() > 1e-6).float()  # LeakyReLU branch undecided within rounding of 0
        ref = torch.nn.functional.leaky_relu(pre, slope)
        gpre = go * torch.where(pre > 0, torch.ones_like(pre), torch.full_like(pre, slope))
    else:
        ref, gpre = pre, go
    q1, q2 = oracle.correlation_backward(gpre, r1, r2, 4)
    a, b = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
    y = AF.correlation(a, b, 4, negative_slope=slope, storage='bf16')
    assert y.dtype == torch.float32
    assert_close(y, ref, 5e-7, 5e-6, 'bf16-storage corr fwd')
    g1, g2 = torch.autograd.grad(y, [a, b], cu(go))
    assert g1.dtype == torch.float32
    assert_close(g1, q1, 5e-6, 1e-5, 'bf16-storage corr gx1')
    assert_close(g2, q2, 5e-6, 1e-5, 'bf16-storage corr gx2')
    # and it is NOT the default: the plain call is the fp32 path
    assert not torch.equal(AF.correlation(cu(x1), cu(x2), 4, slope), y)


@pytest.mark.parametrize('cfg', [(2, 32, 24, 40, 'zeros', True), (1, 5, 17, 33, 'border', False), (16, 32, 96, 160, 'zeros', True)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_bf16_storage_warp(AF, oracle, cfg):
    """Opt-in bf16 storage of the warped source: fp32 sampling, output and gradients; oracle on the rounded source."""
    from arflow_amd.warp_utils import flow_warp
    B, C, H, W, pad, ac = cfg
    gen = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, C, H, W, generator=gen)
    fl = 2.5 * torch.randn(B, 2, H, W, generator=gen)
    go = torch.randn(B, C, H, W, generator=gen)
    xr, fr = x.bfloat16().float().requires_grad_(True), fl.clone().requires_grad_(True)
    ref = oracle.flow_warp(xr, fr, pad=pad, align_corners=ac)
    rgx, rgf = torch.autograd.grad(ref, [xr, fr], go)
    a, f = cu(x).requires_grad_(True), cu(fl).requires_grad_(True)
    y = flow_warp(a, f, pad=pad, align_corners=ac, storage_dtype=torch.bfloat16)
    ulp = 2.0 ** -23 * max(H, W)
    assert_close(y, ref, ((2e-6 + 4 * ulp) * float(x.abs().max())) / 20, 5e-7, 'bf16-storage warp fwd')
    gx, gf = torch.autograd.grad(y, [a, f], cu(go))
    assert gx.dtype == torch.float32
    assert_close(gx, rgx, (1e-5 * max(1.0, float(rgx.abs().max()))) / 10, 1e-5, 'bf16-storage warp gsrc')
    assert_close(gf, rgf, (1e-5 * (C ** 0.5) * float(x.abs().max()) * 4) / 20, 1e-5, 'bf16-storage warp gflow')


def test_general_parameter_space_golden(golden):
    """The parameter values no shipped config uses, against vectors frozen from the REFERENCE
    (tests/golden/general.npz): flow_warp(mode='nearest') for both paddings / align_corners with d/d source;
    SSIM(md=2,3) and TernaryLoss(max_distance=4,5) with both gradients."""
    from arflow_amd import loss_blocks as LB
    from arflow_amd.warp_utils import flow_warp
    g = golden('general')
    for name in g['wnames']:
        for pad in ('zeros', 'border'):
            for ac in (True, False):
                tag = '%s_%s_%d' % (name, pad, int(ac))
                x = cu(g[name + '_x']).requires_grad_(True)
                fl = cu(g[name + '_flow']).requires_grad_(True)
                y = flow_warp(x, fl, pad=pad, mode='nearest', align_corners=ac)
                # a coordinate within rounding of x.5 may pick the other neighbour: compare where it is not
                assert float((y.cpu() != g[tag + '_y']).float().mean()) <= 0.01, tag
                gx, gf = torch.autograd.grad(y, [x, fl], cu(g[name + '_g']))
                assert float((gx.cpu() - g[tag + '_gx']).abs().gt(1e-5).float().mean()) <= 0.02, tag + ' gx'
                assert float(gf.abs().max()) == 0.0  # grid_sample's nearest mode has no grid gradient
    im1, im2 = cu(g['im1']), cu(g['im2'])
    for md in (2, 3):
        a, b = im1.clone().requires_grad_(True), im2.clone().requires_grad_(True)
        y = LB.SSIM(a, b, md=md)
        assert_close(y, g['ssim%d' % md], 5e-6, 1e-6, 'ssim md=%d' % md)
        ga, gb = torch.autograd.grad(y, [a, b], cu(g['ssim%d_g' % md]))
        for got_, key in ((ga, 'ssim%d_ga' % md), (gb, 'ssim%d_gb' % md)):
            ref_ = g[key]
            assert_close(got_, ref_, (2e-4 * float(ref_.abs().max())) / 50, 2e-5, key)
    for md, sd in ((4, True), (5, False)):
        tag = 'tern%d_%d' % (md, int(sd))
        a, b = im1.clone().requires_grad_(True), im2.clone().requires_grad_(True)
        dist, tm = LB.TernaryLoss(a, b, md, sd)
        assert_close(dist, g[tag + '_dist'], 2e-5, 2e-5, tag)
        assert_close(tm, g[tag + '_mask'], 0, 0, tag + ' mask')
        ga, gb = torch.autograd.grad(dist, [a, b], cu(g[tag + '_g']))
        for got_, key in ((ga, '_ga'), (gb, '_gb')):
            ref_ = g[tag + key]
            assert_close(got_, ref_, (1e-4 * float(ref_.abs().max())) / 20, 1e-5, tag + key)


def test_flow_warp_bicubic_vs_reference_vectors(golden):
    """flow_warp(mode='bicubic') (utils/warp_utils.py:83-90 -> grid_sample bicubic) on the generic kernels against the
    reference's outputs and autograd gradients (tests/golden/general.npz), both paddings and align_corners flags."""
    from arflow_amd.warp_utils import flow_warp
    g = golden('general')
    for name in g['wnames']:
        for pad in ('zeros', 'border'):
            for ac in (True, False):
                tag = '%s_%s_%d' % (name, pad, int(ac))
                x, fl = cu(g[name + '_x']).requires_grad_(True), cu(g[name + '_flow']).requires_grad_(True)
                y = flow_warp(x, fl, pad=pad, mode='bicubic', align_corners=ac)
                mx = float(g[name + '_x'].abs().max())
                assert_close(y, g[tag + '_cub_y'], 4e-6 * mx, 1e-5, tag + ' bicubic')
                gx, gf = torch.autograd.grad(y, [x, fl], cu(g[name + '_g']))
                assert_close(gx, g[tag + '_cub_gx'], 1e-5, 1e-5, tag + ' bicubic gx')
                assert_close(gf, g[tag + '_cub_gf'], 2e-6 * (1 + float(g[tag + '_cub_gf'].abs().max())), 2e-5, tag + ' bicubic gflow')


@pytest.mark.parametrize('cfg', [(4, 1, 4, 1, 1), (3, 3, 2, 2, 2), (20, 1, 20, 1, 2), (2, 1, 4, 1, 1), (5, 3, 4, 2, 1)],
                         ids=lambda c: 'pad%d_k%d_d%d_s%d_%d' % c)
def test_general_correlation_parameters(oracle, cfg):
    """Correlation(pad_size, kernel_size, max_displacement, stride1, stride2) with the CUDA extension's semantics
    (correlation_cuda.cc:10-34, correlation_cuda_kernel.cu:41-114) against oracle.correlation_general and ITS autograd
    gradient.  Parity unpinned beyond the default parameter set (the extension cannot be built here); the default set
    must equal the tuned kernels and the pinned correlation_native restatement."""
    from arflow_amd.correlation import Correlation
    pad, k, d, s1, s2 = cfg
    gen = torch.Generator().manual_seed(sum(cfg))
    B, C, H, W = 2, 6, 26, 31
    x1, x2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, H, W, generator=gen)
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    ref = oracle.correlation_general(a, b, pad, k, d, s1, s2)
    go = torch.randn(ref.shape, generator=gen)
    r1, r2 = torch.autograd.grad(ref, [a, b], go)
    m = Correlation(pad_size=pad, kernel_size=k, max_displacement=d, stride1=s1, stride2=s2, corr_multiply=1)
    ac, bc = cu(x1).requires_grad_(True), cu(x2).requires_grad_(True)
    y = m(ac, bc)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert_close(y, ref, 1e-6, 1e-5, 'general corr fwd')
    g1, g2 = torch.autograd.grad(y, [ac, bc], cu(go))
    assert_close(g1, r1, 5e-6, 1e-5, 'general corr gx1')
    assert_close(g2, r2, 5e-6, 1e-5, 'general corr gx2')
    if cfg == (4, 1, 4, 1, 1):
        assert m.general is None  # the default set runs the tuned kernels
        assert_close(y, oracle.correlation(x1, x2, 4), 1e-6, 1e-5, 'default == correlation_native')


@pytest.mark.parametrize('family', ['column', 'ordered', 'column-forward-only'])
def test_uflow_loss_both_directions_in_one_pass_equals_sequential(family, monkeypatch):
    """UFlowLoss with_bk: the one-pass form over 2B (image pair, direction) samples (arflow_census_warp_pair_*) against the
    per-direction form (the reference's order, losses/uflow_loss.py:30-54): same losses, mask and flow gradients; with the
    pair-shared column kernels (census_col.hip, the default) and with the ordered-pair kernels (census_warp.hip)."""
    monkeypatch.setenv('ARFLOW_CENSUS_COL', {'ordered': '0', 'column': '1', 'column-forward-only': 'f'}[family])
    from arflow_amd.config import AttrDict
    from arflow_amd.losses import UFlowLoss
    from oracle.fixture_common import synth_pair
    gen = torch.Generator().manual_seed(21)
    B, H, W = 3, 64, 96
    img = synth_pair(B, H, W, gen)[0].cuda()
    flows = [(3.0 / s) * torch.randn(B, 4, H // s, W // s, generator=gen).cuda() for s in (1, 2, 4)]
    res = {}
    for pair in (True, False):
        loss = UFlowLoss(AttrDict(edge_constant=150, w_smooth=4.0, w_census=1.0, with_bk=True, smooth_order=1))
        loss.pair = pair
        f = [t.clone().requires_grad_(True) for t in flows]
        out = loss(f, img)
        g = torch.autograd.grad(out[0], [f[0], f[2]])
        res[pair] = ([o.detach() for o in out], g)
    for k, n in enumerate(['total', 'census', 'smooth', '|flow|']):
        assert_close(res[True][0][k], res[False][0][k], 1e-7, 2e-6, 'pair vs sequential ' + n)
    assert_close(res[True][0][4], res[False][0][4], 0, 0, 'mask1')
    for a, b, n in zip(res[True][1], res[False][1], ('d flow0', 'd flow2')):
        assert_close(a, b, 1e-7 * float(b.abs().max()) + 1e-12, 1e-5, n)
