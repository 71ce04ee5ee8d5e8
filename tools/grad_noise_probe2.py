"""Which op carries the HIP-vs-oracle gradient difference?  Swap ONE op family at a time to the oracle (on GPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import arflow_amd.models as M
import arflow_amd.models.pwclite_uflow as mpu
import arflow_amd.models.blocks as mb
from arflow_amd import loss_blocks as LB
from arflow_amd.warp_utils import flow_warp
from oracle import ops as O
from oracle.fixture_common import synth_pair, fill_deterministic
from oracle.host_models import OracleCorrelation
from tests.helpers import model_cases
from tests.test_models_gpu import _smooth_objective

tag = 'pwclite_uflow_0'
_, cls, cfg, frames, _ = [c for c in model_cases() if c[0] == tag][0]
x = synth_pair(2, 128, 192, torch.Generator().manual_seed(5))[0].cuda()


def run(model_warp, corr, loss_warp, smooth, act):
    m = fill_deterministic(getattr(M, cls)(cfg)).cuda().train()
    old = (mpu.flow_warp, mb.bias_act)
    try:
        mpu.flow_warp = model_warp
        if act:
            mb.bias_act = lambda y, b, s: torch.nn.functional.leaky_relu(y + b.view(1, -1, 1, 1), s)
        if corr:
            m.corr = OracleCorrelation(4)
        r = m(x, with_bk=True)
        l = _smooth_objective(r['flows_fw'], r['flows_bw'], x, loss_warp, smooth)
        g = torch.autograd.grad(l, list(m.parameters()), allow_unused=True)
    finally:
        mpu.flow_warp, mb.bias_act = old
    return [None if t is None else t.detach().double() for t in g], [n for n, _ in m.named_parameters()], \
        [f.detach() for f in r['flows_fw']]


def worst(g1, g2, names):
    w = (0, None)
    for n, a, b in zip(names, g1, g2):
        if a is None or b is None:
            continue
        r = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)
        if r > w[0]:
            w = (r, n)
    return w


ref, names, fref = run(O.flow_warp, True, O.flow_warp, O.smooth_grad_1st, True)
for label, args in [('all hip', (flow_warp, False, flow_warp, LB.smooth_grad_1st, False)),
                    ('hip model warp only', (flow_warp, True, O.flow_warp, O.smooth_grad_1st, True)),
                    ('hip corr only', (O.flow_warp, False, O.flow_warp, O.smooth_grad_1st, True)),
                    ('hip loss warp only', (O.flow_warp, True, flow_warp, O.smooth_grad_1st, True)),
                    ('hip smooth only', (O.flow_warp, True, O.flow_warp, LB.smooth_grad_1st, True)),
                    ('hip act only', (O.flow_warp, True, O.flow_warp, O.smooth_grad_1st, False))]:
    g, _, fl = run(*args)
    print('%-22s' % label, worst(g, ref, names), 'flow diff', max(float((a - b).abs().max()) for a, b in zip(fl, fref)))
