"""UFlowLoss on the gfx950 kernels -- same constructor, inputs and 5-tuple result as
losses/uflow_loss.py:8-109.

Per step (with_bk, the usual case): BOTH directions as one pass over 2B samples -- 1 launch for the x1/4 copies + grey planes
of both images, 1 splat launch (level-2 range maps), 1 launch for warp + validity mask + clamp/x4 upsample + census loss
(arflow_census_warp_pair_fwd; one launch backward) and 1 fused smoothness launch (`_both_directions`).  Otherwise, per
direction: one launch per image for its x1/4 copy + grey plane (arflow_down4_gray); 1 splat launch
(level-2 range map), ONE launch for warp + validity mask + clamp/x4 upsample of the range map + census loss
(arflow_census_warp_fwd; its backward is one launch too) and 1 fused smoothness launch; the reference issues ~150
ATen kernels and ~25 full-resolution 49-channel temporaries for the same work.
"""
import torch
import torch.nn as nn

from .. import functional as AF
from ..uflow_utils import census_loss


class UFlowLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.fused = True  # False: the unfused photometric path (warp, mask upsample and census as separate launches)
        self.pair = True   # False: the two directions one after the other (the reference's order)

    def _smooth(self, flow_ab2, im_small):
        cfg = self.cfg
        order = int(cfg.smooth_order)
        if order not in (1, 2):
            raise NotImplementedError('smooth_order must be 1 or 2')
        s = AF.smooth_sums(flow_ab2, im_small, 1.0, float(cfg.edge_constant), order, 1, 1)
        b, _, h, w = flow_ab2.shape
        nx, ny = float(b * 2 * h * (w - order)), float(b * 2 * (h - order) * w)
        return cfg.w_smooth * (s[0] / nx + s[1] / ny) / 2.

    def _direction(self, a, b, flow_ab0, flow_ba2, flow_ab2):
        """a, b: dicts of one image each: 'im' [B,3,H,W], 'small' its x1/4 copy, 'gray' its grey plane or None."""
        cfg = self.cfg
        occ_small = AF.splat_map(flow_ba2, 0)
        H, W = a['im'].shape[2:]
        if a['gray'] is not None and occ_small.shape[2:] == (H // 4, W // 4):
            # warp + validity + x4 mask upsample + census loss as ONE launch (and one backward launch)
            l_c, mask = AF.census_warp_loss(a['gray'], b['gray'], flow_ab0, occ_small, 7)
        else:
            # im_a ~ warp(im_b, flow_ab0); only d/d flow is needed (source detached, uflow_loss.py:31,34)
            recons, valid = AF.warp_with_valid(b['im'].detach(), flow_ab0, pad='zeros', align_corners=True, norm=AF.NORM_UFLOW)
            mask = AF.up4_clamp_mul(occ_small, valid)
            l_c = census_loss(a['im'], recons, mask)
        return cfg.w_census * l_c, self._smooth(flow_ab2, a['small']), mask

    def _prepare(self, im):
        H, W = im.shape[2:]
        if H % 4 == 0 and W % 4 == 0:
            if self.fused and AF.census_warp_supported(H, W):
                small, gray = AF.down4_gray(im)
            else:
                small, gray = AF.down4(im), None
        else:
            small = torch.nn.functional.interpolate(im.detach(), scale_factor=0.25, mode='bilinear', align_corners=False)
            gray = None
        return {'im': im, 'small': small, 'gray': gray}

    def _both_directions(self, output, target):
        """with_bk, as ONE pass over a batch of 2B samples s = 2 b + direction: [B,4,h,w] flows ARE [2B,2,h,w], the
        [B,6,H,W] pair IS [2B,3,H,W] (views, no copy); sample s reads its second image / its occlusion map from its
        partner s ^ 1.  3 launches forward (grey planes + x1/4 copies, range maps + smoothness sums, census) and ONE
        backward for what the per-direction path issues twice; the smoothness term only needs the directions' sum."""
        cfg = self.cfg
        B, _, H, W = target.shape
        h, w = output[2].shape[2:]
        order = int(cfg.smooth_order)
        if order not in (1, 2):
            raise NotImplementedError('smooth_order must be 1 or 2')
        small, gray, occ = AF.down4_gray(target.view(2 * B, 3, H, W), zero_plane=True)  # occ: cleared, the splat target
        f0, f2 = output[0].view(2 * B, 2, H, W), output[2].view(2 * B, 2, h, w)
        # range maps + smoothness sums, then census of both directions; ONE launch backward for all of it
        l_fw, l_bw, s, mask = AF.uflow_pair_loss(gray, small, f0, f2, occ, float(cfg.edge_constant), order, 7)
        loss_warp = cfg.w_census * l_fw + cfg.w_census * l_bw
        nx, ny = float(B * 2 * h * (w - order)), float(B * 2 * (h - order) * w)  # elements per DIRECTION
        loss_smooth = cfg.w_smooth * (s[0] / nx + s[1] / ny) / 2.
        return loss_warp + loss_smooth, loss_warp, loss_smooth, output[0].abs().mean(), mask.view(B, 2, 1, H, W)[:, 0]

    def forward(self, output, target):
        """output: list of [B,4,h,w] (fw,bw) flows, finest first; target: [B,6,H,W] image pair."""
        H, W = target.shape[2:]
        if (self.fused and self.pair and self.cfg.with_bk and target.is_cuda and target.is_contiguous()
                and output[0].is_contiguous() and output[2].is_contiguous() and output[0].shape[1] == 4
                and H % 4 == 0 and W % 4 == 0 and AF.census_warp_supported(H, W)
                and tuple(output[2].shape[2:]) == (H // 4, W // 4)):
            return self._both_directions(output, target)
        f12_0, f21_0 = output[0][:, 0:2], output[0][:, 2:4]
        f12_2, f21_2 = output[2][:, 0:2], output[2][:, 2:4]
        one, two = self._prepare(target[:, :3]), self._prepare(target[:, 3:])
        loss_warp, loss_smooth, mask1 = self._direction(one, two, f12_0, f21_2, f12_2)
        if self.cfg.with_bk:
            lw, ls, _ = self._direction(two, one, f21_0, f12_2, f21_2)
            loss_warp = loss_warp + lw
            loss_smooth = loss_smooth + ls
        return loss_warp + loss_smooth, loss_warp, loss_smooth, output[0].abs().mean(), mask1
