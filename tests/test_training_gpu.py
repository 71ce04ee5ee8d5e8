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


def _worker_global_norm_gpu(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from arflow_amd import ddp
    from arflow_amd.config import AttrDict
    from arflow_amd.losses import UFlowLoss
    from oracle.fixture_common import synth_pair
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    gen = torch.Generator().manual_seed(9)
    B, H, W = 4, 64, 96
    img = synth_pair(B, H, W, gen)[0]
    flows = [2.0 * torch.randn(B, 4, H // s, W // s, generator=gen) for s in (1, 2, 4)]
    flows[0][:2, :, :, :20] += 40.0  # the first two samples lose a quarter of their pixels: unequal mask sums per rank
    cfg = AttrDict(edge_constant=150, w_smooth=4.0, w_census=1.0, with_bk=True, smooth_order=1)
    loss = UFlowLoss(cfg)

    def run(sl):
        f = [t[sl].to(dev).requires_grad_(True) for t in flows]
        out = loss(f, img[sl].to(dev))
        g = torch.autograd.grad(out[0], [f[0], f[2]])
        return out[0].detach(), g

    res = {}
    full_loss, full_g = run(slice(0, B))  # single process, gathered batch (the reference's trainer semantics)
    half = slice(rank * B // world, (rank + 1) * B // world)
    for on in (False, True):
        ddp.enable_global_loss_norm(on)
        l, g = run(half)
        lt = l.cpu().clone()
        dist.all_reduce(lt)
        # gradient of the rank-averaged loss w.r.t. THIS rank's flows = (1/world) * local gradient
        err = max(float((a / world - b[half]).abs().max()) / float(b[half].abs().max()) for a, b in zip(g, full_g))
        res[on] = (abs(float(lt) / world - float(full_loss)) / abs(float(full_loss)), err)
    ddp.enable_global_loss_norm(False)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_global_loss_normalisation_product_loss_two_ranks():
    """The PRODUCT UFlowLoss (HIP kernels) on a batch sharded over 2 ranks with ARFLOW_GLOBAL_LOSS_NORM semantics
    (ddp.enable_global_loss_norm): rank-averaged loss and gradients equal the single-process loss on the gathered batch
    (trainer/uflow_trainer.py:48-54, utils/uflow_utils.py:293); per-rank normalisation (the default) does not."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_global_norm_gpu, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, 'rank failed'
    out = [q.get() for _ in range(2)]
    for _, res in out:
        assert res[True][0] < 2e-6 and res[True][1] < 1e-4, 'global normalisation != gathered batch: %s' % (res,)
        assert res[False][1] > 1e-3, 'the masks were meant to be unequal across ranks: %s' % (res,)  # per-rank default differs
