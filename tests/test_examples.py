"""BASELINE config 1: the reference's example frames (examples/img{0,1,2}.png resized to 384x640) through
PWCLite 2-frame and 3-frame.  CPU: product host model + oracle ops vs the reference's flows;
GPU: the same through the HIP kernels (EPE <= 1e-3 px).  Plus the .flo writer / reader round trip."""
import os

import numpy as np
import pytest
import torch

from arflow_amd import flow_io
from arflow_amd.config import AttrDict
from oracle.fixture_common import fill_deterministic
from tests.helpers import epe


def _run(golden, device, patch):
    import arflow_amd.models as M
    g = golden('examples')
    x = (g['frames_u8'].float() / 255).to(device)
    out = {}
    for tag, frames, sl in (('two', 2, slice(3, 9)), ('three', 3, slice(0, 9))):
        model = fill_deterministic(M.PWCLite(AttrDict(upsample=True, n_frames=frames, reduce_dense=True))).to(device).eval()
        with torch.no_grad():
            if patch:
                from oracle.host_models import oracle_ops
                with oracle_ops(model):
                    res = model(x[:, sl].contiguous(), with_bk=True)
            else:
                res = model(x[:, sl].contiguous(), with_bk=True)
        for k in ('flows_fw', 'flows_bw'):
            full = torch.nn.functional.avg_pool2d(res[k][0], 8)
            q = res[k][1] if res[k][1].shape[2] == 96 else torch.nn.functional.avg_pool2d(res[k][1], 2)
            out['%s_%s_full_pooled8' % (tag, k)] = full
            out['%s_%s_quarter' % (tag, k)] = q
    for k, v in out.items():
        e = epe(v, g[k])
        assert e <= 1e-3, '%s: EPE %.3e px vs the reference' % (k, e)


def test_examples_cpu_host_model_with_oracle_ops(golden):
    torch.set_num_threads(8)
    _run(golden, 'cpu', patch=True)


@pytest.mark.gpu
def test_examples_on_hip_kernels(golden):
    _run(golden, 'cuda', patch=False)


def test_flo_roundtrip_and_epe(tmp_path):
    rng = np.random.default_rng(0)
    flow = rng.standard_normal((37, 53, 2)).astype(np.float32)
    p = os.path.join(tmp_path, 'a.flo')
    flow_io.write_flow(p, flow)
    assert os.path.getsize(p) == 12 + 37 * 53 * 2 * 4
    back = flow_io.read_flow(p)
    assert back.shape == (37, 53, 2) and np.array_equal(back, flow)
    assert flow_io.epe(back, flow) == 0.0
    assert abs(flow_io.epe(flow + np.array([3.0, 4.0], np.float32), flow) - 5.0) < 1e-6
    with open(p, 'r+b') as f:
        f.write(b'\x00\x00\x00\x00')
    with pytest.raises(ValueError):
        flow_io.read_flow(p)
