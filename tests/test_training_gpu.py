"""GPU end-to-end checks that no parity test covers: (1) the whole training step (model on the HIP ops + UFlowLoss +
backward + Adam, chairs_uflow.json hyper-parameters) descends: the unsupervised loss falls and stays finite over 40
steps on a fixed batch; (2) `python bench.py --gpus 2` from a plain shell starts its own ranks and prints one JSON line (on a one-GPU box
the ranks share the device and reduce over gloo: the launch / reduction path of the driver's scaling run)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_step_descends():
    from arflow_amd.train_step import TrainStep
    dev = torch.device('cuda')
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(3)
    B, H, W, dx, dy = 4, 128, 192, 3, 2
    base = torch.rand(B, 3, (H + 16) // 8, (W + 16) // 8, generator=g)
    tex = torch.nn.functional.interpolate(base, (H + 16, W + 16), mode='bicubic', align_corners=False).clamp(0, 1)
    tex = (tex + 0.1 * torch.rand(B, 3, H + 16, W + 16, generator=g)).clamp(0, 1)
    im1 = tex[:, :, 8:8 + H, 8:8 + W]
    im2 = tex[:, :, 8 - dy:8 - dy + H, 8 - dx:8 - dx + W]  # im2(p + (dx, dy)) = im1(p): forward flow = (+dx, +dy)
    x = torch.cat([im1, im2], 1).contiguous().to(dev)
    step = TrainStep('pwclite_uflow+uflow_loss', dev, lr=1e-4, seed=1)  # Adam 1e-4: configs/chairs_uflow.json:29-48
    step.model.level_dropout = 0.0
    losses = [float(step(x)) for _ in range(40)]
    assert all(l == l and abs(l) < 1e6 for l in losses), 'loss went non-finite'
    first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    assert last < 0.99 * first, 'the unsupervised loss did not fall: %.4f -> %.4f' % (first, last)


def test_bench_self_launches_two_ranks():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--size', '128', '192', '--batch', '2'], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and len(d['per_rank_ms_per_step']) == 2
    assert d['config']['global_batch'] == 4 and d['config']['loss_finite']
    assert d['collective_backend'] in ('nccl', 'gloo')
    assert d['oversubscribed'] == (torch.cuda.device_count() < 2)
