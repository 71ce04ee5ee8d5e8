"""GPU parity of the FUSED pyramid level (arflow_level_warp_fwd / _moments / _corr_fwd / _corr_bwd
behind arflow_amd.functional.level) against the CPU oracle chain the reference runs per level
(models/pwclite_uflow.py:203-222, models/uflow_model.py:160-198):

    flow = interpolate(flow_c * 2, x2, bilinear); x2w = flow_warp(x2, flow); x1n, x2n = normalize_features([x1, x2w]);
    buf = cat([leaky_relu(corr(x1n, x2n), 0.1), x1n, flow, member], 1)

at the launch shapes of BASELINE configs 2, 3 and 4.  Tolerances (fp32, written per assertion): the upsampled flow
differs from ATen's CPU kernel by <= 2 ulp of |flow|; through the warp that moves a sampled feature by (slope of the
feature map) x 2 ulp(coordinate), which is why the quantities behind the warp carry a tolerance relative to max|x|.

Tolerances (round 3): every assert_close() below was re-derived from the error MEASURED on MI355X -- tests/conftest.py
records max(err / tol) per call site, profiles/r03_parity_margins.json holds the summary -- and sites that had more than
20x headroom were divided down (the `/ N` factors and the small literals) so that each keeps about 10x over its measured
error (float atomics and summation order move the error by 2-3x from run to run).  Sites left as they were sit within
20x of their measured error already.
"""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def AF():
    from arflow_amd import functional
    return functional


@pytest.fixture(scope='module')
def O():
    from oracle import ops
    torch.set_num_threads(16)
    return ops


def smooth(t):
    """3x3 box blur: features with a bounded slope (the conv features the op sees are smooth, white noise is not)."""
    return F.avg_pool2d(F.pad(t, (1, 1, 1, 1), mode='replicate'), 3, 1)


def oracle_level(O, x1, x2, flow_c, mode, up_align, pad, member, coord='arflow'):
    if flow_c is not None:
        flow = F.interpolate(flow_c * 2, scale_factor=2, mode='bilinear', align_corners=up_align)
        if coord == 'arflow':
            x2w = O.flow_warp(x2, flow, pad=pad, align_corners=up_align)
        else:
            x2w = O.resample(x2, O.flow_to_warp(flow))
    else:
        flow, x2w = None, x2
    if mode == 'joint':
        y1, y2 = O.normalize_features_joint([x1, x2w])
    else:
        y1, y2 = O.normalize_features_uflow([x1, x2w], normalize=True, center=True, moments_across_channels=True,
                                            moments_across_images=True)
    pre = O.correlation(y1, y2, 4)
    vol = F.leaky_relu(pre, 0.1)
    parts = [vol, y1] + ([flow] if flow is not None else []) + [member]
    return torch.cat(parts, 1), flow, pre, x2w


LEVEL_SHAPES = [
    # (B, C, H, W, has_flow): config 2 pyramid of PWCLiteUflow (fw + bw stacked)
    (16, 32, 12, 20, False), (16, 32, 24, 40, True), (16, 32, 48, 80, True), (16, 32, 96, 160, True),
    # config 3 (448x1024, batch 4) and config 4 (256x448)
    (8, 32, 14, 32, False), (8, 32, 28, 64, True), (8, 32, 112, 256, True), (16, 32, 32, 56, True),
    # ragged: H, W not multiples of the 8 x 32 tile, a single sample
    (1, 32, 10, 12, True), (3, 8, 6, 44, True),
]


@pytest.mark.parametrize('mode', ['joint', 'avg'])
@pytest.mark.parametrize('shape', LEVEL_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_level_forward_backward_vs_oracle(AF, O, shape, mode):
    B, C, H, W, has_flow = shape
    if mode == 'avg' and B * H * W > 16 * 48 * 80:
        pytest.skip('avg statistics are covered at the smaller shapes')
    gen = torch.Generator().manual_seed(B * 11 + C * 5 + H + (7 if mode == 'avg' else 0))
    x1 = smooth(torch.randn(B, C, H, W, generator=gen)) + 0.3
    x2 = smooth(torch.randn(B, C, H, W, generator=gen)) + 0.1
    flow_c = smooth(1.5 * torch.randn(B, 2, H // 2, W // 2, generator=gen)) if has_flow else None
    member = torch.randn(B, 5, H, W, generator=gen)
    for t in (x1, x2, member) + ((flow_c,) if has_flow else ()):
        t.requires_grad_(True)
    ref_buf, ref_flow, pre, ref_x2w = oracle_level(O, x1, x2, flow_c, mode, True, 'zeros', member)
    gbuf = torch.randn(ref_buf.shape, generator=gen)
    gbuf[:, :81] *= (pre.detach().abs() > 1e-6).float()  # at a LeakyReLU kink either derivative is right
    gflow = torch.randn(B, 2, H, W, generator=gen) if has_flow else None
    loss = (ref_buf * gbuf).sum() + ((ref_flow * gflow).sum() if has_flow else 0.0)
    inputs = [x1, x2, member] + ([flow_c] if has_flow else [])
    refs = torch.autograd.grad(loss, inputs)

    a, b, m = [t.detach().cuda().requires_grad_(True) for t in (x1, x2, member)]
    fc = flow_c.detach().cuda().requires_grad_(True) if has_flow else None
    if has_flow:
        cfg = AF.LevelCfg(['vol', 'x1n', 'flow', 0], mode, 0.1, 4, True, True, 'zeros', True)
        buf, flow = AF.level(a, b, fc, cfg, m)
    else:
        cfg = AF.LevelCfg(['vol', 'x1n', 0], mode, 0.1, 4)
        buf, flow = AF.level(a, b, None, cfg, m), None
    assert buf.shape == ref_buf.shape
    fmax = float(ref_flow.detach().abs().max()) if has_flow else 0.0
    xmax = float(x2.detach().abs().max())
    tag = '%s %s' % (shape, mode)
    if has_flow:
        # ATen's CPU kernel and this one round the same four products in a different order: <= 2 ulp of |flow|
        assert_close(flow, ref_flow, 4e-7 * max(fmax, 1.0), 0, 'flow_up ' + tag)
        assert_close(buf[:, 81 + C:81 + C + 2], ref_flow, 4e-7 * max(fmax, 1.0), 0, 'flow slot ' + tag)
    assert_close(buf[:, 81:81 + C], ref_buf[:, 81:81 + C], 1e-6, 5e-6, 'x1n ' + tag)
    # volume: O(1) values; behind the warp (see the module docstring)
    vol_tol = 2e-6 + (2e-5 * xmax if has_flow else 0.0)
    assert_close(buf[:, :81], ref_buf[:, :81], vol_tol, 1e-5, 'volume ' + tag)
    assert_close(buf[:, -5:], member, 0, 0, 'member copy ' + tag)
    loss_g = (buf * gbuf.cuda()).sum() + ((flow * gflow.cuda()).sum() if has_flow else 0.0)
    gin = [a, b, m] + ([fc] if has_flow else [])
    got = torch.autograd.grad(loss_g, gin)
    names = ['d x1', 'd x2', 'd member'] + (['d flow_c'] if has_flow else [])
    for n, g, r in zip(names, got, refs):
        scale = float(r.abs().max())
        if n == 'd member':
            assert_close(g, r, 0, 0, n + ' ' + tag)
        else:
            # float atomics (d/d x2 behind the warp), 81-term sums; relative to the gradient's magnitude
            assert_close(g, r, 2e-5 * scale + 1e-6, 1e-4, n + ' ' + tag)


def test_level_matches_unfused_ops_same_device(AF):
    """The fused launches against the stand-alone HIP ops on the SAME upsampled flow (bit-identical warp inputs):
    the volume of the normalised pair computed from the raw maps equals corr(normalize_pair(...)) to summation order."""
    gen = torch.Generator().manual_seed(5)
    B, C, H, W = 4, 32, 48, 80
    x1 = (smooth(torch.randn(B, C, H, W, generator=gen)) + 0.3).cuda()
    x2 = (smooth(torch.randn(B, C, H, W, generator=gen)) + 0.1).cuda()
    fc = smooth(1.5 * torch.randn(B, 2, H // 2, W // 2, generator=gen)).cuda()
    member = torch.randn(B, 3, H, W, generator=gen).cuda()
    cfg = AF.LevelCfg(['vol', 'x1n', 'flow', 0], 'joint', 0.1, 4, True, True, 'zeros', True)
    buf, flow = AF.level(x1, x2, fc, cfg, member)
    x2w = AF.warp(x2, flow, 'zeros', True, AF.NORM_ARFLOW)
    y1, y2 = AF.normalize_pair(x1, x2w, 'joint')
    vol = AF.correlation(y1, y2, 4, 0.1)
    assert_close(buf[:, 81:81 + C], y1, 1e-6, 1e-6, 'x1n')
    assert_close(buf[:, :81], vol, 2e-6, 1e-5, 'volume')


def test_level_mean_far_from_zero(AF, O):
    """mean = 25 sigma: the epilogue form sum(x1n x2w) - mu sum(x1n) cancels 25 x larger terms; stays within 2e-5."""
    gen = torch.Generator().manual_seed(9)
    B, C, H, W = 2, 32, 24, 40
    x1 = smooth(torch.randn(B, C, H, W, generator=gen)) * 0.2 + 5.0
    x2 = smooth(torch.randn(B, C, H, W, generator=gen)) * 0.2 + 5.0
    member = torch.zeros(B, 1, H, W)
    ref_buf, _, _, _ = oracle_level(O, x1, x2, None, 'joint', True, 'zeros', member)
    cfg = AF.LevelCfg(['vol', 'x1n', 0], 'joint', 0.1, 4)
    buf = AF.level(x1.cuda(), x2.cuda(), None, cfg, member.cuda())
    assert_close(buf[:, 81:81 + C], ref_buf[:, 81:81 + C], 2e-5, 1e-5, 'x1n at mean = 25 sigma')
    assert_close(buf[:, :81], ref_buf[:, :81], 2e-5, 1e-5, 'volume at mean = 25 sigma')


@pytest.mark.parametrize('shape', [(4, 32, 16, 28), (2, 32, 32, 56)], ids=lambda s: 'x'.join(map(str, s)))
def test_level_uflow_layout(AF, O, shape):
    """The level as PWCFlow runs it (models/uflow_model.py:160-198): upsample(flow, is_flow=True) (align_corners=False),
    resample(f2, flow_to_warp(flow_up)), 'avg' statistics, cat([context_up, flow_up, volume, features1]) -- the
    normalised first map is NOT concatenated (kept on the side for the backward), the raw one is a member."""
    B, C, H, W = shape
    gen = torch.Generator().manual_seed(H)
    x1 = (smooth(torch.randn(B, C, H, W, generator=gen)) + 0.2).requires_grad_(True)
    x2 = (smooth(torch.randn(B, C, H, W, generator=gen)) - 0.1).requires_grad_(True)
    flow_c = smooth(1.2 * torch.randn(B, 2, H // 2, W // 2, generator=gen)).requires_grad_(True)
    ctx = torch.randn(B, 6, H, W, generator=gen).requires_grad_(True)
    flow = F.interpolate(flow_c, scale_factor=2, mode='bilinear', align_corners=False) * 2
    x2w = O.resample(x2, O.flow_to_warp(flow))
    y1, y2 = O.normalize_features_uflow([x1, x2w], normalize=True, center=True, moments_across_channels=True,
                                        moments_across_images=True)
    pre = O.correlation(y1, y2, 4)
    ref = torch.cat([ctx, flow, F.leaky_relu(pre, 0.1), x1], 1)
    gbuf = torch.randn(ref.shape, generator=gen)
    gbuf[:, 8:8 + 81] *= (pre.detach().abs() > 1e-6).float()
    gflow = torch.randn(B, 2, H, W, generator=gen)
    refs = torch.autograd.grad((ref * gbuf).sum() + (flow * gflow).sum(), [x1, x2, flow_c, ctx])

    a, b, fc, cx = [t.detach().cuda().requires_grad_(True) for t in (x1, x2, flow_c, ctx)]
    cfg = AF.LevelCfg([0, 'flow', 'vol', 1], 'avg', 0.1, 4, True, False, 'zeros', True, AF.NORM_UFLOW)
    buf, fu = AF.level(a, b, fc, cfg, cx, a)
    fmax = float(flow.detach().abs().max())
    assert_close(fu, flow, 4e-7 * max(fmax, 1.0), 0, 'flow_up')
    assert_close(buf[:, 8:8 + 81], ref[:, 8:8 + 81], (2e-6 + 2e-5 * float(x2.detach().abs().max())) / 2, 5e-6, 'volume')
    assert_close(buf[:, :6], ctx, 0, 0, 'context copy')
    assert_close(buf[:, -C:], x1, 0, 0, 'features1 copy')
    got = torch.autograd.grad((buf * gbuf.cuda()).sum() + (fu * gflow.cuda()).sum(), [a, b, fc, cx])
    for n, g, r in zip(['d x1', 'd x2', 'd flow', 'd context'], got, refs):
        assert_close(g, r, 2e-5 * float(r.abs().max()) + 1e-6, 1e-4, n)


@pytest.mark.parametrize('env', [{'ARFLOW_WARP_SLAB': '1'}, {'ARFLOW_LEVEL_SMALL': '0'}, {'ARFLOW_WARP_GATHER': '1'}],
                         ids=['two-pass-slab', 'no-per-sample-kernel', 'inverse-window-gather'])
def test_opt_in_kernel_variants_in_a_fresh_process(env):
    """The library reads ARFLOW_WARP_SLAB / ARFLOW_LEVEL_SMALL once per process (they size workspaces), so the variants they
    select -- the atomics-free two-pass form of the warp's source gradient, its gather form over the inverse-flow window
    (DESIGN.md 4.1), the tiled kernels at the coarsest level -- are exercised by re-running the oracle comparison of the shapes they affect in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = {'ARFLOW_WARP_SLAB': '16x32x96x160 and joint', 'ARFLOW_LEVEL_SMALL': '16x32x12x20',
           'ARFLOW_WARP_GATHER': '(16x32x96x160 or 8x32x112x256)'}[next(iter(env))]
    cmd = [sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_level_gpu.py'), '-q', '-x', '-m', 'gpu', '-k',
           'test_level_forward_backward_vs_oracle and ' + sel]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and ' passed' in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
