"""unFlowLoss (ARFlow pyramid loss) on the gfx950 kernels -- same constructor, inputs and 4-tuple
result as losses/flow_loss.py:8-114."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF
from ..loss_blocks import smooth_grad_1st, smooth_grad_2nd
from ..ddp import global_denominator
from ..warp_utils import flow_warp, get_occu_mask_backward, get_occu_mask_bidirection


class unFlowLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def loss_photomatric(self, im1_scaled, im1_recons, occu_mask1):
        """losses/flow_loss.py:13-27 as ONE fused launch: L1, SSIM and mask sums together."""
        cfg = self.cfg
        if cfg.w_ternary > 0:
            # float * (dist, mask) tuple in the reference (flow_loss.py:23-25, loss_blocks.py:62)
            raise TypeError("unsupported operand type(s) for *: 'float' and 'tuple' "
                            '(unFlowLoss with w_ternary > 0 is broken in the reference as well)')
        b, c, h, w = im1_scaled.shape
        s = AF.PhotoSumsFunction.apply(im1_scaled, im1_recons, occu_mask1)
        total = 0.
        if cfg.w_l1 > 0:
            total = total + cfg.w_l1 * s[0] / float(b * c * h * w)
        if cfg.w_ssim > 0:
            total = total + cfg.w_ssim * s[1] / float(b * c * (h - 2) * (w - 2))
        # mask.mean() of the gathered batch when global normalisation is on (ddp.global_denominator), else this rank's
        return total / (global_denominator(s[2]) / float(b * h * w))

    def loss_smooth(self, flow, im1_scaled, scale):
        cfg = self.cfg
        if 'smooth_2nd' in cfg and cfg.smooth_2nd:
            s = AF.smooth_sums(flow, im1_scaled, scale, cfg.alpha, 2, 0, 0)
            b, _, h, w = flow.shape
            return (s[0] / float(b * 2 * h * (w - 2))) / 2. + (s[1] / float(b * 2 * (h - 2) * w)) / 2.
        s = AF.smooth_sums(flow, im1_scaled, scale, cfg.alpha, 1, 0, 0)
        b, _, h, w = flow.shape
        return (s[0] / float(b * 2 * h * (w - 1)) / 2.) / 2. + (s[1] / float(b * 2 * (h - 1) * w) / 2.) / 2.

    pair = True  # False: the two directions one after the other (the reference's order)

    def _forward_stacked(self, output, target):
        """with_bk as ONE pass over 2B samples (first B: direction 1 -> 2, last B: direction 2 -> 1; every op on this path is
        per sample): one area resize, one warp, one occlusion / mask resize and one smoothness launch per pyramid scale instead
        of two -- the photometric sums stay per direction (each is divided by its own mask mean, losses/flow_loss.py:27)."""
        cfg = self.cfg
        B = target.shape[0]
        imgs = torch.cat([target[:, :3], target[:, 3:]], 0)
        warp_losses, smooth_losses = [], []
        self.pyramid_occu_mask1, self.pyramid_occu_mask2 = [], []
        s, m0 = 1., None
        for i, flow in enumerate(output):
            if cfg.w_scales[i] == 0:
                warp_losses.append(0)
                smooth_losses.append(0)
                continue
            _, _, h, w = flow.shape
            im = F.interpolate(imgs, (h, w), mode='area')
            f = torch.cat([flow[:, :2], flow[:, 2:]], 0)
            rec = flow_warp(torch.roll(im, B, 0), f, pad=cfg.warp_pad)
            if i == 0:
                fsw = torch.roll(f, B, 0)
                m = 1 - (get_occu_mask_backward(fsw, th=0.2) if cfg.occ_from_back else get_occu_mask_bidirection(f, fsw))
                m0 = m
                s = min(h, w)
            else:
                m = F.interpolate(m0, (h, w), mode='nearest')
            self.pyramid_occu_mask1.append(m[:B])
            self.pyramid_occu_mask2.append(m[B:])
            warp_losses.append((self.loss_photomatric(im[:B], rec[:B], m[:B]) + self.loss_photomatric(im[B:], rec[B:], m[B:])) / 2.)
            smooth_losses.append(self.loss_smooth(f, im, 1.0 / s))  # the mean over 2B samples IS the directions' average
        warp_loss = sum(l * w for l, w in zip(warp_losses, cfg.w_scales))
        smooth_loss = cfg.w_smooth * sum(l * w for l, w in zip(smooth_losses, cfg.w_sm_scales))
        return warp_loss + smooth_loss, warp_loss, smooth_loss, output[0].abs().mean()

    def forward(self, output, target):
        cfg = self.cfg
        if self.pair and cfg.with_bk and target.is_cuda and all(f.shape[1] == 4 for f in output):
            return self._forward_stacked(output, target)
        im1_origin, im2_origin = target[:, :3], target[:, 3:]
        warp_losses, smooth_losses = [], []
        self.pyramid_occu_mask1, self.pyramid_occu_mask2 = [], []
        s = 1.
        for i, flow in enumerate(output):
            if cfg.w_scales[i] == 0:
                warp_losses.append(0)
                smooth_losses.append(0)
                continue
            _, _, h, w = flow.shape
            im1 = F.interpolate(im1_origin, (h, w), mode='area')
            im2 = F.interpolate(im2_origin, (h, w), mode='area')
            rec1 = flow_warp(im2, flow[:, :2], pad=cfg.warp_pad)
            rec2 = flow_warp(im1, flow[:, 2:], pad=cfg.warp_pad)
            if i == 0:
                if cfg.occ_from_back:
                    m1 = 1 - get_occu_mask_backward(flow[:, 2:], th=0.2)
                    m2 = 1 - get_occu_mask_backward(flow[:, :2], th=0.2)
                else:
                    m1 = 1 - get_occu_mask_bidirection(flow[:, :2], flow[:, 2:])
                    m2 = 1 - get_occu_mask_bidirection(flow[:, 2:], flow[:, :2])
            else:
                m1 = F.interpolate(self.pyramid_occu_mask1[0], (h, w), mode='nearest')
                m2 = F.interpolate(self.pyramid_occu_mask2[0], (h, w), mode='nearest')
            self.pyramid_occu_mask1.append(m1)
            self.pyramid_occu_mask2.append(m2)

            l_warp = self.loss_photomatric(im1, rec1, m1)
            if i == 0:
                s = min(h, w)
            l_smooth = self.loss_smooth(flow[:, :2], im1, 1.0 / s)
            if cfg.with_bk:
                l_warp = (l_warp + self.loss_photomatric(im2, rec2, m2)) / 2.
                l_smooth = (l_smooth + self.loss_smooth(flow[:, 2:], im2, 1.0 / s)) / 2.
            warp_losses.append(l_warp)
            smooth_losses.append(l_smooth)

        warp_loss = sum(l * w for l, w in zip(warp_losses, cfg.w_scales))
        smooth_loss = cfg.w_smooth * sum(l * w for l, w in zip(smooth_losses, cfg.w_sm_scales))
        return warp_loss + smooth_loss, warp_loss, smooth_loss, output[0].abs().mean()
