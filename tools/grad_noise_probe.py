"""Diagnostic: where does the whole-model parameter-gradient difference HIP-vs-CPU-oracle come from?
Compares (a) product model + HIP ops on GPU, (b) product model + oracle ops ON THE GPU (MIOpen convs, ATen ops),
(c) product model + oracle ops on CPU.  (a)-(b) isolates the hot-path kernels; (b)-(c) is conv/ATen noise."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import arflow_amd.models as M
from arflow_amd import loss_blocks as LB
from arflow_amd.warp_utils import flow_warp
from oracle import ops as O
from oracle.fixture_common import synth_pair, fill_deterministic
from oracle.host_models import oracle_ops
from tests.helpers import model_cases
from tests.test_models_gpu import _smooth_objective

torch.set_num_threads(16)
for tag in sys.argv[1:] or ['pwclite_uflow_0', 'pwclite2']:
    _, cls, cfg, frames, _ = [c for c in model_cases() if c[0] == tag][0]
    x = synth_pair(2, 192, 256, torch.Generator().manual_seed(5))[0]

    LINEAR = os.environ.get('LINEAR') == '1'
    WS = float(os.environ.get('WSCALE', '1'))
    import arflow_amd.models.blocks as mb

    def run(dev, oracle):
        m = fill_deterministic(getattr(M, cls)(cfg)).to(dev).train()
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if n_.endswith('weight'):
                    p_.mul_(WS)
        xx = x.to(dev)
        if oracle:
            with oracle_ops(m):
                if LINEAR:
                    mb.bias_act = lambda y, b, s: y + b.view(1, -1, 1, 1)
                r = m(xx, with_bk=True)
                l = _smooth_objective(r['flows_fw'], r['flows_bw'], xx, O.flow_warp, O.smooth_grad_1st)
        else:
            old = mb.bias_act
            if LINEAR:
                mb.bias_act = lambda y, b, s: old(y, b, 1.0)
            try:
                r = m(xx, with_bk=True)
            finally:
                mb.bias_act = old
            l = _smooth_objective(r['flows_fw'], r['flows_bw'], xx, flow_warp, LB.smooth_grad_1st)
        print('   |flow| per level', [round(float(f.abs().mean()), 3) for f in r['flows_fw']])
        g = torch.autograd.grad(l, list(m.parameters()), allow_unused=True)
        return float(l), [None if t is None else t.detach().cpu().double() for t in g], [n for n, _ in m.named_parameters()]

    la, ga, names = run('cuda', False)
    lb, gb, _ = run('cuda', True)
    lb2, gb2, _ = run('cuda', True)
    lc, gc, _ = run('cpu', True)

    def worst(g1, g2):
        w = (0, None, 0, None)
        for n, a, b in zip(names, g1, g2):
            if a is None or b is None:
                continue
            r = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-20)
            l2 = float((a - b).norm()) / (float(b.norm()) + 1e-20)
            if r > w[0]:
                w = (r, n) + w[2:]
            if l2 > w[2]:
                w = w[:2] + (l2, n)
        return w
    print(tag, 'loss hip %.8f gpu-oracle %.8f cpu-oracle %.8f' % (la, lb, lc))
    print('  hip vs gpu-oracle     ', worst(ga, gb))
    print('  gpu-oracle vs itself  ', worst(gb2, gb))
    print('  gpu-oracle vs cpu     ', worst(gb, gc))
    print('  hip vs cpu-oracle     ', worst(ga, gc))
