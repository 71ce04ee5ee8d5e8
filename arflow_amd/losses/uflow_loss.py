"""UFlowLoss on the gfx950 kernels -- same constructor, inputs and 5-tuple result as
losses/uflow_loss.py:8-109.

Per direction: 1 warp launch (3-channel image, also emits the validity mask), 1 splat launch
(level-2 range map), 1 fused clamp + x4 upsample + mask multiply, 1 fused census-loss launch,
1 image x1/4 launch and 1 fused smoothness launch; the reference issues ~150 ATen kernels and
~25 full-resolution 49-channel temporaries for the same work.
"""
import torch
import torch.nn as nn

from .. import functional as AF
from ..uflow_utils import census_loss


class UFlowLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def _direction(self, im_a, im_b, flow_ab0, flow_ba2, flow_ab2):
        cfg = self.cfg
        # im_a ~ warp(im_b, flow_ab0); only d/d flow is needed (source detached, uflow_loss.py:31,34)
        recons, valid = AF.warp_with_valid(im_b.detach(), flow_ab0, pad='zeros', align_corners=True, norm=AF.NORM_UFLOW)
        occ_small = AF.splat_map(flow_ba2, 0)
        mask = AF.up4_clamp_mul(occ_small, valid)
        l_census = cfg.w_census * census_loss(im_a, recons, mask)
        im_small = AF.down4(im_a) if (im_a.shape[2] % 4 == 0 and im_a.shape[3] % 4 == 0) else \
            torch.nn.functional.interpolate(im_a.detach(), scale_factor=0.25, mode='bilinear', align_corners=False)
        order = int(cfg.smooth_order)
        if order not in (1, 2):
            raise NotImplementedError('smooth_order must be 1 or 2')
        s = AF.smooth_sums(flow_ab2, im_small, 1.0, float(cfg.edge_constant), order, 1, 1)
        b, _, h, w = flow_ab2.shape
        nx, ny = float(b * 2 * h * (w - order)), float(b * 2 * (h - order) * w)
        l_smooth = cfg.w_smooth * (s[0] / nx + s[1] / ny) / 2.
        return l_census, l_smooth, mask

    def forward(self, output, target):
        """output: list of [B,4,h,w] (fw,bw) flows, finest first; target: [B,6,H,W] image pair."""
        f12_0, f21_0 = output[0][:, 0:2], output[0][:, 2:4]
        f12_2, f21_2 = output[2][:, 0:2], output[2][:, 2:4]
        im1, im2 = target[:, :3], target[:, 3:]
        loss_warp, loss_smooth, mask1 = self._direction(im1, im2, f12_0, f21_2, f12_2)
        if self.cfg.with_bk:
            lw, ls, _ = self._direction(im2, im1, f21_0, f12_2, f21_2)
            loss_warp = loss_warp + lw
            loss_smooth = loss_smooth + ls
        return loss_warp + loss_smooth, loss_warp, loss_smooth, output[0].abs().mean(), mask1
