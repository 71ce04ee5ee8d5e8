"""GPU: every BASELINE.json workload END TO END at its own resolution -- the product model on the HIP kernels + the
product loss against the same host model with the oracle ops patched in on the CPU + the oracle loss
(oracle.host_models.oracle_ops, oracle/losses.py), identical deterministic weights and synthetic pairs, batch 1:

  config 2   pwclite_uflow + UFlowLoss   384 x 640        config 2 (literal model)  pwclite + unFlowLoss  384 x 640
  config 3   pwclite_uflow + UFlowLoss   448 x 1024       config 4 (per-GPU shape)  pwcflow + UFlowLoss   256 x 448
  config 5   pwclite 3-frame + mv loss   384 x 640

Gates: flow EPE <= 1e-3 px at every pyramid level (the north star's gate), loss terms within 2e-4 relative (the loss
sums ~1e6 thresholded mask pixels; a 1e-6 px flow difference flips a handful)."""
import pytest
import torch

from tests.helpers import epe

pytestmark = pytest.mark.gpu

CASES = [('config2', 'pwclite_uflow+uflow_loss', 384, 640), ('config2-literal', 'pwclite+unflow_loss', 384, 640),
         ('config3', 'pwclite_uflow+uflow_loss', 448, 1024), ('config4', 'pwcflow+uflow_loss', 256, 448),
         ('config5', 'pwclite3+mv_loss', 384, 640)]


@pytest.mark.parametrize('case', CASES, ids=lambda c: c[0])
def test_workload_end_to_end_vs_oracle_twin(case):
    from arflow_amd.config import AttrDict
    from arflow_amd.losses import get_loss
    from arflow_amd.models import get_model
    from arflow_amd.train_step import WORKLOADS, synthetic_pairs
    from oracle import losses as OL
    from oracle.fixture_common import fill_deterministic
    from oracle.host_models import oracle_ops
    _, workload, H, W = case
    mcfg, lcfg = WORKLOADS[workload]
    mcfg = dict(mcfg)
    if 'level_dropout' in mcfg:
        mcfg['level_dropout'] = 0.0
    frames = mcfg.get('n_frames', 2)
    x = synthetic_pairs(1, H, W, frames=frames, device='cpu', seed=7)
    torch.set_num_threads(16)

    def flows_of(res):
        if lcfg['type'] == 'mv':
            return None
        return [torch.cat([a, b], 1) for a, b in zip(res['flows_fw'], res['flows_bw'])]

    # oracle twin on the CPU
    cpu_model = fill_deterministic(get_model(AttrDict(mcfg))).eval()
    ocls = {'uflow': OL.UFlowLoss, 'unflow': OL.unFlowLoss, 'mv': OL.MvLoss}[lcfg['type']]
    with torch.no_grad(), oracle_ops(cpu_model):
        rref = cpu_model(x, with_bk=True)
        if lcfg['type'] == 'mv':
            lref = ocls(AttrDict(lcfg))(rref['flows_fw'], rref['flows_bw'], x)
        else:
            lref = ocls(AttrDict(lcfg))(flows_of(rref), x)
    # product on the GPU
    gpu_model = fill_deterministic(get_model(AttrDict(mcfg))).cuda().eval()
    xc = x.cuda()
    with torch.no_grad():
        res = gpu_model(xc, with_bk=True)
        if lcfg['type'] == 'mv':
            lgot = get_loss(AttrDict(lcfg))(res['flows_fw'], res['flows_bw'], xc)
        else:
            lgot = get_loss(AttrDict(lcfg))(flows_of(res), xc)
    for k in ('flows_fw', 'flows_bw'):
        assert len(res[k]) == len(rref[k])
        for i, (a, b) in enumerate(zip(res[k], rref[k])):
            e = epe(a, b)
            assert e <= 1e-3, '%s %s level %d: EPE %.3e px vs the oracle twin' % (workload, k, i, e)
    for i in range(4):
        a, b = float(lgot[i]), float(lref[i])
        assert abs(a - b) <= 2e-4 * abs(b) + 1e-6, '%s loss term %d: %.8g vs %.8g' % (workload, i, a, b)
