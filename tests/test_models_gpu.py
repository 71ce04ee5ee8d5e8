"""GPU: the product host models running on the gfx950 kernels against the REFERENCE's flows,
losses and parameter-gradient fingerprints for identical deterministic weights
(tests/golden/models.npz).  Gate: flow EPE <= 1e-3 px (BASELINE north star)."""
import numpy as np
import pytest
import torch

from oracle.fixture_common import fill_deterministic, loss_cfgs, pool_to_quarter
from tests.helpers import epe, model_cases

pytestmark = pytest.mark.gpu
# The golden fingerprints (|g|_1 and sum(g) per parameter, from the REFERENCE with its losses) pass through the
# thresholded occlusion masks of the losses (occluded iff range map < 0.2 etc.): a flow difference of 1e-6 px flips
# a few mask pixels, each worth O(1/Sigma mask) of the loss, so they are a coarse 2 % net.  The TIGHT check of the
# whole-model backward is test_model_parameter_gradients_elementwise below (element-wise, 1e-3 of max|g| per
# parameter, on an objective without thresholds); op-level gradients are pinned tightly in test_hip_parity.py.
TOL = 2e-2


def _loss_for(tag):
    from arflow_amd import losses as L
    from arflow_amd.config import AttrDict
    uflow, unflow, _ = loss_cfgs()
    if tag == 'pwclite2':
        cfg = AttrDict(dict(unflow[0][1]))
        cfg['w_scales'] = [1.0, 1.0, 1.0, 1.0, 1.0, 0.0]
        cfg['w_sm_scales'] = [1.0, 0.0, 0.0, 0.0, 0.0, 0.0]
        return L.unFlowLoss(cfg)
    if tag == 'pwclite_uflow_1':
        return L.UFlowLoss(AttrDict(uflow[0][1]))
    if tag == 'pwcflow':
        return L.UFlowLoss(AttrDict(uflow[1][1]))
    return None


@pytest.mark.parametrize('case', model_cases(), ids=lambda c: c[0])
def test_model_on_hip_ops_matches_reference(golden, case):
    import arflow_amd.models as M
    tag, cls, cfg, frames, with_bk = case
    g = golden('models')
    x3 = g['x3'].float() / 255
    x = (x3 if frames == 3 else x3[:, :6].contiguous()).cuda()
    model = fill_deterministic(getattr(M, cls)(cfg)).cuda().eval()
    loss_fn = _loss_for(tag)
    with torch.set_grad_enabled(loss_fn is not None):
        res = model(x, with_bk=with_bk)
    for k in ('flows_fw', 'flows_bw'):
        if (tag + '_%s_0' % k) not in g:
            continue
        for i, f in enumerate(res[k]):
            ref = g['%s_%s_%d' % (tag, k, i)]
            e = epe(pool_to_quarter(f.detach(), x.shape[2]), ref)
            assert e <= 1e-3, '%s %s level %d: EPE %.3e px vs the reference' % (tag, k, i, e)
    if loss_fn is None:
        return
    flows = [torch.cat([a, b], 1) for a, b in zip(res['flows_fw'], res['flows_bw'])]
    lres = loss_fn(flows, x)
    ref_loss = float(g[tag + '_loss'])
    assert abs(float(lres[0].detach()) - ref_loss) <= 2e-4 * abs(ref_loss) + 1e-5, (float(lres[0].detach()), ref_loss)
    lres[0].backward()
    names = g[tag + '_gnames']
    gsum, gabs = g.raw(tag + '_gsum'), g.raw(tag + '_gabs')
    params = dict(model.named_parameters())
    assert list(params) == names
    def fp(n, fn):
        gr = params[n].grad  # None where the reference's gradient is None as well (recorded as 0)
        return 0.0 if gr is None else float(fn(gr.double()))
    got_abs = np.array([fp(n, lambda t: t.abs().sum()) for n in names])
    got_sum = np.array([fp(n, lambda t: t.sum()) for n in names])
    # fingerprints of every parameter gradient: |g|_1 within TOL, signed sum within TOL of |g|_1
    rel = np.abs(got_abs - gabs) / (gabs + 1e-7)
    w = int(np.argmax(rel))
    assert rel[w] <= TOL, 'worst |g|_1 mismatch %.3e at %s (got %.6g ref %.6g)' % (rel[w], names[w], got_abs[w], gabs[w])
    rel = np.abs(got_sum - gsum) / (gabs + 1e-7)
    w = int(np.argmax(rel))
    assert rel[w] <= TOL, 'worst sum(g) mismatch %.3e at %s (got %.6g ref %.6g)' % (rel[w], names[w], got_sum[w], gsum[w])


@pytest.mark.gpu
def test_conv_block_with_fused_epilogue_matches_torch():
    """blocks.conv(...) (bias-free MIOpen conv + fused bias/LeakyReLU pass) == Conv2d(bias) + LeakyReLU(0.1):
    output, input gradient, weight and bias gradients; same parameter names as the reference's Sequential."""
    import torch.nn as nn
    from arflow_amd.models.blocks import conv
    torch.manual_seed(3)
    blk = conv(12, 20, kernel_size=3, stride=2, dilation=1).cuda()
    assert sorted(blk.state_dict().keys()) == ['0.bias', '0.weight']
    ref = nn.Sequential(nn.Conv2d(12, 20, 3, stride=2, padding=1), nn.LeakyReLU(0.1)).cuda()
    ref.load_state_dict(blk.state_dict())
    x = torch.randn(3, 12, 21, 34, device='cuda')
    go = torch.randn(3, 20, 11, 17, device='cuda')
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = blk(xa), ref(xb)
    assert float((ya - yb).abs().max()) <= 1e-5 * float(yb.abs().max())
    ga = torch.autograd.grad(ya, [xa] + list(blk.parameters()), go)
    gb = torch.autograd.grad(yb, [xb] + list(ref.parameters()), go)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def _smooth_objective(flows_fw, flows_bw, x, warp, smooth):
    """A threshold-free objective over every output level of both directions: L1 photometric error of the
    border-padded warp + first-order edge-aware smoothness (losses/flow_loss.py:13-36 without the occlusion
    masks).  Continuous in the flows, so the HIP and the oracle gradients agree to fp32 rounding."""
    import torch.nn.functional as F
    im1, im2 = x[:, :3], x[:, 3:6]
    total = 0.
    for lv, (fw, bw) in enumerate(zip(flows_fw, flows_bw)):
        h, w = fw.shape[2:]
        a, b = F.interpolate(im1, (h, w), mode='area'), F.interpolate(im2, (h, w), mode='area')
        for f, src, dst in ((fw, b, a), (bw, a, b)):
            total = total + (dst - warp(src, f, pad='border')).abs().mean() / (lv + 1) + 0.1 * smooth(f / 20., dst, 10.)
    return total


def _param_grads(cls, cfg, x, dev, hip, linear):
    """Parameter gradients of the product model `cls` under the threshold-free objective, with the HIP ops
    (hip=True, GPU only) or with the oracle ops patched in (oracle.host_models.oracle_ops, any device).
    linear=True replaces the LeakyReLU behind every conv by the identity (on the HIP side: the same fused
    bias/activation kernel with slope 1), which removes the host model's own kinks."""
    import arflow_amd.models as M
    import arflow_amd.models.blocks as mb
    from arflow_amd import loss_blocks as LB
    from arflow_amd.warp_utils import flow_warp
    from oracle import ops as O
    from oracle.host_models import oracle_ops
    m = fill_deterministic(getattr(M, cls)(cfg)).to(dev).train()
    xx = x.to(dev)
    if hip:
        old = mb.bias_act
        if linear:
            mb.bias_act = lambda y, b, s: old(y, b, 1.0)
        try:
            res = m(xx, with_bk=True)
        finally:
            mb.bias_act = old
        loss = _smooth_objective(res['flows_fw'], res['flows_bw'], xx, flow_warp, LB.smooth_grad_1st)
    else:
        with oracle_ops(m):
            if linear:
                mb.bias_act = lambda y, b, s: y + b.view(1, -1, 1, 1)  # restored by oracle_ops on exit
            res = m(xx, with_bk=True)
            loss = _smooth_objective(res['flows_fw'], res['flows_bw'], xx, O.flow_warp, O.smooth_grad_1st)
    g = torch.autograd.grad(loss, list(m.parameters()), allow_unused=True)
    return float(loss.detach()), [None if t is None else t.detach().cpu().double() for t in g], [n for n, _ in m.named_parameters()]


def _worst_rel(ga, gb, names):
    worst = (0.0, None)
    for n, a, b in zip(names, ga, gb):
        assert (a is None) == (b is None), n
        if b is None:
            continue
        assert bool(torch.isfinite(a).all()), n
        rel = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)
        if rel > worst[0]:
            worst = (rel, n)
    return worst


@pytest.mark.parametrize('tag,linear', [('pwclite2', True), ('pwclite2', False), ('pwclite_uflow_0', False),
                                        ('pwclite_uflow_1', False), ('pwcflow', False), ('pwcflow', True)])
# (ADVICE r2 asked for a fixed-tolerance `linear` variant per architecture.  PWCFlow has one (1e-3, like pwclite2).  For
#  PWCLiteUflow it cannot be held: with the conv activations removed some bias gradients nearly cancel (max|g| 2e-5 against
#  3e-2 elsewhere) and the ORACLE twin on the GPU differs from the oracle twin on the CPU by 0.74 of that maximum -- more
#  than the HIP path does (0.38): measured with tools-style probing in round 3, nothing a tolerance can pin.)
def test_model_parameter_gradients_elementwise(tag, linear):
    """Whole-model backward, ELEMENT-WISE per parameter tensor: the product model on the HIP kernels against
    the same host model with the oracle ops patched in, same deterministic weights and inputs.

    What the tolerance can be was MEASURED, not assumed (profiles/r02_grad_noise_{leaky,linear}.log,
    tools/grad_noise_probe.py): these networks are piecewise linear (LeakyReLU behind ~60 convs, the fused
    LeakyReLU of the cost volume, floor() in the sampler), and with fixed weights a 1e-7 perturbation of the
    forward pass -- MIOpen vs the CPU's convolution, nothing else -- flips a few kinks: the ORACLE-patched
    model on the GPU differs from the ORACLE-patched model on the CPU by 4e-3 .. 2e-2 of max|g| in its worst
    parameter, while two GPU runs agree to 2e-6.  So:
      * ('pwclite2', linear): activations replaced by the identity -> no flip on this input; HIP vs the CPU
        oracle twin must agree to 1e-3 (measured 8e-5), strictly;
      * the LeakyReLU models: HIP vs the oracle twin ON THE SAME DEVICE (same MIOpen convs: isolates the
        hot-path kernels, measured 2e-5 .. 6e-3 where a kink flips) must lie within max(1e-3, 2 x the oracle's
        own GPU-vs-CPU spread); a wrong tap, sign or channel in a hot-path gradient is O(1) of max|g| in every
        parameter upstream of it and fails either bound (and the op-level tests at the bench shapes,
        tests/test_bench_shapes_gpu.py, pin each kernel's gradient to 1e-5 .. 1e-4 on their own)."""
    from oracle.fixture_common import synth_pair
    case = [c for c in model_cases() if c[0] == tag][0]
    _, cls, cfg, frames, _ = case
    x = synth_pair(2, 192, 256, torch.Generator().manual_seed(5))[0]
    torch.set_num_threads(16)
    l_hip, g_hip, names = _param_grads(cls, cfg, x, 'cuda', True, linear)
    l_cpu, g_cpu, _ = _param_grads(cls, cfg, x, 'cpu', False, linear)
    assert abs(l_hip - l_cpu) <= 2e-6 * abs(l_cpu), (l_hip, l_cpu)
    if linear:
        w = _worst_rel(g_hip, g_cpu, names)
        assert w[0] <= 1e-3, 'worst parameter gradient vs the CPU oracle twin: %s differs by %.3e of its max' % (w[1], w[0])
        return
    l_gor, g_gor, _ = _param_grads(cls, cfg, x, 'cuda', False, linear)
    spread = _worst_rel(g_gor, g_cpu, names)[0]  # the oracle against itself across conv implementations
    w = _worst_rel(g_hip, g_gor, names)
    # the run-time spread may widen the bound only up to a FIXED ceiling: a spread beyond what was measured for
    # these networks is itself a failure, and the bound never exceeds 4e-2 of max|g|.  Measured: 4e-3 .. 2e-2
    # (profiles/r02_grad_noise_leaky.log) and 2.02e-2 for PWCFlow once the GPU twin ran the op-by-op wiring with the oracle
    # ops (round 3: oracle_ops switches the fused level launches off) -> ceiling 2.5e-2
    assert spread <= 2.5e-2, 'oracle GPU-vs-CPU spread %.3e exceeds the measured range (profiles/r02_grad_noise_leaky.log)' % spread
    bound = min(max(1e-3, 2.0 * spread), 4e-2)
    assert w[0] <= bound, 'worst parameter gradient vs the oracle twin: %s differs by %.3e of its max (bound %.3e, ' \
                          'oracle GPU-vs-CPU spread %.3e)' % (w[1], w[0], bound, spread)


def test_pwclite_five_frames_on_hip_ops(golden):
    """5-frame sliding windows of PWCLite (models/pwclite.py:274-281) on the gfx950 kernels against the reference's flows
    (tests/golden/models5.npz): EPE <= 1e-3 px at every level of every window."""
    import arflow_amd.models as M
    from arflow_amd.config import AttrDict as C
    g = golden('models5')
    x = (g['x5'].float() / 255).cuda()
    model = fill_deterministic(M.PWCLite(C(upsample=True, n_frames=3, reduce_dense=True))).cuda().eval()
    with torch.no_grad():
        res = model(x, with_bk=True)
    for k in ('flows_fw', 'flows_bw'):
        for w, flows in enumerate(res[k]):
            for i, f in enumerate(flows):
                e = epe(pool_to_quarter(f, x.shape[2]), g['pwclite5_%s_%d_%d' % (k, w, i)])
                assert e <= 1e-3, '5-frame %s window %d level %d: EPE %.3e px' % (k, w, i, e)
