"""Synthetic-data training step: the inner loop of the reference trainer
(trainer/uflow_trainer.py:30-91: model(img_pair, with_bk=True) -> cat(fw,bw) per level -> loss ->
zero_grad / backward / optimizer.step) as a reusable object, with one-process-per-GPU gradient
all-reduce instead of DataParallel and without the per-step host syncs (``loss.item()`` NaN assert,
trainer/uflow_trainer.py:57-61, is replaced by an on-device isfinite flag read at the end)."""
import torch

from .config import AttrDict
from .ddp import FlatGradAllReduce
from .losses import get_loss
from .models import get_model

# name -> (model cfg, loss cfg); hyper-parameters from configs/chairs_uflow.json:17-48 where present
WORKLOADS = {
    # BASELINE config 2: "PWCLite 2-frame fwd+bwd ... (correlation d=4 + warp + uflow_loss)".  The ARFlow
    # PWCLite class cannot feed UFlowLoss, so it is the UFlow-structured PWCLite.
    'pwclite_uflow+uflow_loss': (
        dict(type='pwclite_uflow', level_dropout=0.1, feature_norm=True, align_corners=True, warp_pad='zeros',
             n_frames=2, reduce_dense=False),
        dict(type='uflow', edge_constant=150, w_smooth=4.0, w_census=1.0, with_bk=True, smooth_order=1)),
    'pwclite+unflow_loss': (
        dict(type='pwclite', upsample=True, n_frames=2, reduce_dense=True),
        dict(type='unflow', w_l1=0.15, w_ssim=0.85, w_ternary=0.0, warp_pad='border', alpha=10, occ_from_back=True,
             with_bk=True, w_smooth=75.0, w_scales=[1.0, 1.0, 1.0, 1.0, 1.0, 0.0],
             w_sm_scales=[1.0, 0.0, 0.0, 0.0, 0.0, 0.0])),
    # BASELINE config 5: 3-frame PWCLite (forward_3_frames) + the build-defined multi-view objective
    'pwclite3+mv_loss': (
        dict(type='pwclite', upsample=True, n_frames=3, reduce_dense=True),
        dict(type='mv', w_l1=0.15, w_ssim=0.85, alpha=10, w_smooth=75.0, w_scales=[1.0, 1.0, 1.0, 1.0, 0.0],
             w_sm_scales=[1.0, 0.0, 0.0, 0.0, 0.0])),
    # BASELINE config 4: configs/chairs_uflow.json (model type 'uflow' = PWCFlow)
    'pwcflow+uflow_loss': (
        dict(type='uflow', feature_norm=True, level_dropout=0.1),
        dict(type='uflow', edge_constant=150, w_smooth=4.0, w_census=1.0, with_bk=True, smooth_order=1)),
}


def synthetic_pairs(batch, height, width, frames=2, device='cuda', seed=0):
    """U[0,1) images smoothed by one 5x5 box blur so census / SSIM see structure."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = torch.rand(batch, 3 * frames, height, width, generator=g)
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode='reflect'), 5, 1)
    return x.to(device)


class TrainStep:
    def __init__(self, workload, device, lr=1e-4, seed=0, n_buckets=4, feature_storage='fp32'):
        mcfg, lcfg = WORKLOADS[workload]
        if feature_storage != 'fp32':
            if mcfg['type'] != 'pwclite_uflow':
                raise ValueError('feature_storage=%r is wired into the pwclite_uflow model only' % feature_storage)
            mcfg = dict(mcfg, feature_storage=feature_storage)
        self.model_cfg, self.loss_cfg = AttrDict(mcfg), AttrDict(lcfg)
        torch.manual_seed(seed)
        self.model = get_model(self.model_cfg)
        self.model.init_weights()
        self.model.to(device).train()
        self.loss = get_loss(self.loss_cfg)
        import os
        self.reducer = FlatGradAllReduce(self.model, n_buckets=n_buckets,
                                         force_collectives=os.environ.get('ARFLOW_FORCE_COLLECTIVES') == '1')
        self.reducer.broadcast_parameters(0)
        # optional gathered-batch loss normalisation (one scalar all-reduce per masked term)
        from . import ddp
        ddp.enable_global_loss_norm(os.environ.get('ARFLOW_GLOBAL_LOSS_NORM') == '1')
        # Adam, lr 1e-4, betas (0.9, 0.999), eps 1e-8, no decay: configs/chairs_uflow.json:29-48
        kw = dict(lr=lr, betas=(0.9, 0.999), eps=1e-8)
        try:
            self.opt = torch.optim.Adam(self.model.parameters(), fused=(device.type == 'cuda'), **kw)
        except (TypeError, RuntimeError):
            self.opt = torch.optim.Adam(self.model.parameters(), **kw)
        self.last = None

    def __call__(self, img_pair):
        res = self.model(img_pair, with_bk=True)
        if self.loss_cfg.type == 'mv':
            out = self.loss(res['flows_fw'], res['flows_bw'], img_pair)
        else:
            flows = [torch.cat([fw, bw], 1) for fw, bw in zip(res['flows_fw'], res['flows_bw'])]
            out = self.loss(flows, img_pair)
        self.reducer.zero_grad()
        out[0].backward()
        self.reducer.finish()
        self.opt.step()
        self.last = out[0].detach()
        return self.last
