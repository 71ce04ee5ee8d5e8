#!/usr/bin/env python3
"""How smooth are the flows the warp backward sees in the bench workload?  For the default workload (random-init
PWCLiteUflow, synthetic pairs) prints, per pyramid level, statistics of the upsampled flow that enters the level's warp:
|flow|, the difference between the flow at a target p and at its tap source q, and the spread of q - round(flow(q)) over
8 x 32 tiles -- the quantities that decide whether a gather over the inverse-flow window can replace the scatter
(DESIGN.md 4.1).   python tools/flow_roughness.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from arflow_amd import functional as AF  # noqa: E402
from arflow_amd.config import AttrDict  # noqa: E402
from arflow_amd.models import get_model  # noqa: E402
from arflow_amd.train_step import WORKLOADS, synthetic_pairs  # noqa: E402


def main():
    mcfg, _ = WORKLOADS['pwclite_uflow+uflow_loss']
    torch.manual_seed(0)
    model = get_model(AttrDict(mcfg))
    model.init_weights()
    model = model.cuda().train()
    x = synthetic_pairs(8, 384, 640, frames=2, device='cuda')
    seen = []
    orig = AF.level

    def spy(x1, x2, flow, cfg, *members, **kw):
        out = orig(x1, x2, flow, cfg, *members, **kw)
        if flow is not None:
            seen.append(out[1].detach())
        return out
    AF.level = spy
    with torch.no_grad():
        model(x, with_bk=True)
    AF.level = orig
    for f in seen:
        B, _, H, W = f.shape
        ys, xs = torch.meshgrid(torch.arange(H, device=f.device), torch.arange(W, device=f.device), indexing='ij')
        tx = (xs[None] + f[:, 0]).round().clamp(0, W - 1).long()
        ty = (ys[None] + f[:, 1]).round().clamp(0, H - 1).long()
        idx = (ty * W + tx).view(B, 1, -1).expand(B, 2, -1)
        fq = torch.gather(f.view(B, 2, -1), 2, idx).view_as(f)  # flow at the (nearest) source pixel of each target
        d = (f - fq).abs().amax(1)
        cx = xs[None] - f[:, 0].round()
        cy = ys[None] - f[:, 1].round()
        tiles_x = cx.unfold(1, 8, 8).unfold(2, 32, 32) if H % 8 == 0 and W % 32 == 0 else None
        msg = '%3dx%-3d |flow| mean %.2f max %.1f   |f(p)-f(q)|_inf: mean %.2f  <=2: %.1f%%  <=4: %.1f%%' % (
            H, W, float(f.abs().mean()), float(f.abs().max()), float(d.mean()), 100 * float((d <= 2).float().mean()),
            100 * float((d <= 4).float().mean()))
        if tiles_x is not None:
            ty_ = cy.unfold(1, 8, 8).unfold(2, 32, 32)
            sx = tiles_x.amax((-1, -2)) - tiles_x.amin((-1, -2)) - 31
            sy = ty_.amax((-1, -2)) - ty_.amin((-1, -2)) - 7
            msg += '   tile spread of the window centres beyond the tile: x mean %.1f max %.0f, y mean %.1f max %.0f' % (
                float(sx.mean()), float(sx.max()), float(sy.mean()), float(sy.max()))
        print(msg)


if __name__ == '__main__':
    main()
