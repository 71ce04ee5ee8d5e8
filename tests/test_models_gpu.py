"""GPU: the product host models running on the gfx950 kernels against the REFERENCE's flows,
losses and parameter-gradient fingerprints for identical deterministic weights
(tests/golden/models.npz).  Gate: flow EPE <= 1e-3 px (BASELINE north star)."""
import numpy as np
import pytest
import torch

from oracle.fixture_common import fill_deterministic, loss_cfgs, pool_to_quarter
from tests.helpers import epe, model_cases

pytestmark = pytest.mark.gpu
# parameter gradients pass through floor(), thresholded masks and |.|: an EPE-level (1e-3 px) flow difference
# flips a few taps, so whole-model fingerprints are compared at 2 %; op-level gradients are pinned tightly
# in test_hip_parity.py
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
    assert abs(float(lres[0]) - ref_loss) <= 2e-4 * abs(ref_loss) + 1e-5, (float(lres[0]), ref_loss)
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
    # fingerprints of every parameter gradient: |g|_1 within 0.5 %, signed sum within 0.5 % of |g|_1
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
