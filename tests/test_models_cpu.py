"""CPU: the product host models (arflow_amd/models) with the oracle ops patched in by tests/helpers.py,
against the flows the REFERENCE models produced for the same deterministic weights
(tests/golden/models.npz).  Pins: state_dict key order and shapes (reference checkpoints load by
name), parameter counts, per-level data flow, and the 2B fw/bw batching trick."""
import pytest
import torch

import arflow_amd.models as M
from oracle.fixture_common import fill_deterministic, pool_to_quarter
from tests.helpers import epe, model_cases, oracle_ops

PARAMS = {'pwclite2': 2236660, 'pwclite3': 2371444, 'pwclite_uflow_1': 7227874, 'pwcflow': 5734634}


@pytest.mark.parametrize('case', model_cases(), ids=lambda c: c[0])
def test_host_model_matches_reference_flows(golden, case):
    tag, cls, cfg, frames, with_bk = case
    g = golden('models')
    x3 = g['x3'].float() / 255
    x = x3 if frames == 3 else x3[:, :6].contiguous()
    model = getattr(M, cls)(cfg)
    assert list(model.state_dict().keys()) == g[tag + '_keys'], 'state_dict key order differs from the reference'
    n = sum(p.numel() for p in model.parameters())
    assert n == int(g[tag + '_nparams'])
    if tag in PARAMS:
        assert n == PARAMS[tag]
    fill_deterministic(model)
    model.eval()
    torch.set_num_threads(8)
    with torch.no_grad(), oracle_ops(model):
        res = model(x, with_bk=with_bk)
    for k in ('flows_fw', 'flows_bw'):
        if (tag + '_%s_0' % k) not in g:
            assert k not in res
            continue
        for i, f in enumerate(res[k]):
            ref = g['%s_%s_%d' % (tag, k, i)]
            got = pool_to_quarter(f, x.shape[2])
            assert got.shape == ref.shape
            e = epe(got, ref)
            assert e <= 1e-3, '%s %s level %d: EPE %.3e px vs the reference' % (tag, k, i, e)


def test_level_dropout_consumes_rng_like_the_reference():
    """Training-mode level dropout draws torch.rand(1) once per level + once for the context net, per
    direction, from the CPU RNG (models/pwclite_uflow.py:226-229,240-242)."""
    from arflow_amd.config import AttrDict as C
    cfg = C(level_dropout=0.5, feature_norm=True, align_corners=True, warp_pad='zeros', n_frames=2, reduce_dense=True)
    m = M.PWCLiteUflow(cfg).train()
    torch.manual_seed(123)
    d = m._drops(2, 3, torch.device('cpu'))
    torch.manual_seed(123)
    expect = [[float(torch.rand(1) > 0.5) for _ in range(5)] for _ in range(2)]
    assert d.shape == (5, 6, 1, 1, 1)
    for lvl in range(5):
        assert d[lvl, :3].flatten().tolist() == [expect[0][lvl]] * 3
        assert d[lvl, 3:].flatten().tolist() == [expect[1][lvl]] * 3
    m.eval()
    assert m._drops(2, 3, torch.device('cpu')) is None


def test_pwclite_five_frame_windows_match_reference(golden):
    """PWCLite.forward with 5 frames (models/pwclite.py:274-281): flows_fw = [window(0,1,2) 1->2, window(1,2,3) 2->3],
    flows_bw = [window(1,2,3) 2->1, window(2,3,4) 3->2], each a finest-first list -- against the flows the REFERENCE
    model produced for the same deterministic weights (tests/golden/models5.npz, oracle/make_golden.py::gen_models5)."""
    from arflow_amd.config import AttrDict as C
    g = golden('models5')
    x = g['x5'].float() / 255
    model = fill_deterministic(M.PWCLite(C(upsample=True, n_frames=3, reduce_dense=True))).eval()  # 3-frame estimators, 5 input frames
    torch.set_num_threads(8)
    with torch.no_grad(), oracle_ops(model):
        res = model(x, with_bk=True)
    for k in ('flows_fw', 'flows_bw'):
        assert len(res[k]) == 2
        for w, flows in enumerate(res[k]):
            for i, f in enumerate(flows):
                ref = g['pwclite5_%s_%d_%d' % (k, w, i)]
                got = pool_to_quarter(f, x.shape[2])
                assert got.shape == ref.shape
                e = epe(got, ref)
                assert e <= 1e-3, '5-frame %s window %d level %d: EPE %.3e px vs the reference' % (k, w, i, e)
