"""Multi-view (3-frame) photometric + smoothness objective for BASELINE config 5.

The fork has NO consumer of the 3-frame model output (`n_frames==3` appears only in
models/pwclite.py:271-280 and the dataset collectors).  This wiring is defined by the build and composed
ONLY of reference hot-path functions so that every piece has an oracle: for the two neighbour frames
k in {0, 2} and every pyramid level with a non-zero weight,

    I_hat = flow_warp(area_resize(img_k), f_1k, pad='border')          (utils/warp_utils.py:83-90)
    M     = border_mask(f_1k)                                           (utils/warp_utils.py:119-134)
    photo = (w_l1*mean(|I1 - I_hat|*M) + w_ssim*mean(SSIM(I_hat*M, I1*M))) / (mean(M) + 1e-6)
                                                                        (losses/flow_loss.py:13-27 + eps)
    smooth = smooth_grad_1st(f_1k / min(h0, w0), I1, alpha)             (losses/loss_blocks.py:93-109)

summed over k and halved, weighted by w_scales / w_sm_scales like unFlowLoss (losses/flow_loss.py:105-111).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF
from ..ddp import global_denominator
from ..warp_utils import border_mask, flow_warp


class MvLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    pair = True  # False: the two views one after the other

    def _forward_stacked(self, flows_12, flows_10, target):
        """Both views (centre -> 0, centre -> 2) as one pass over 2B samples: one area resize, warp, border mask and smoothness
        launch per pyramid scale instead of two; the photometric sums stay per view (each has its own mask mean)."""
        cfg = self.cfg
        B = target.shape[0]
        im1 = target[:, 3:6]
        nb = torch.cat([target[:, 0:3], target[:, 6:9]], 0)  # the neighbours of view a (frame 0) and view b (frame 2)
        warp_loss, smooth_loss = 0., 0.
        s = 1.
        for i, (f12, f10) in enumerate(zip(flows_12, flows_10)):
            if cfg.w_scales[i] == 0:
                continue
            b, _, h, w = f12.shape
            if i == 0:
                s = min(h, w)
            i1 = F.interpolate(im1, (h, w), mode='area')
            flow = torch.cat([f10, f12], 0)
            rec = flow_warp(F.interpolate(nb, (h, w), mode='area'), flow, pad='border')
            m = border_mask(flow)
            l_warp = 0.
            for k in (0, 1):
                sums = AF.PhotoSumsFunction.apply(i1, rec[k * B:(k + 1) * B], m[k * B:(k + 1) * B])
                photo = 0.
                if cfg.w_l1 > 0:
                    photo = photo + cfg.w_l1 * sums[0] / float(b * 3 * h * w)
                if cfg.w_ssim > 0:
                    photo = photo + cfg.w_ssim * sums[1] / float(b * 3 * (h - 2) * (w - 2))
                l_warp = l_warp + photo / (global_denominator(sums[2]) / float(b * h * w) + 1e-6)
            l_smooth = 0.
            if cfg.w_sm_scales[i] != 0:  # the sums over 2B samples ARE the two views' sums added
                sm = AF.smooth_sums(flow, torch.cat([i1, i1], 0), 1.0 / s, cfg.alpha, 1, 0, 0)
                l_smooth = (sm[0] / float(b * 2 * h * (w - 1)) / 2.) / 2. + (sm[1] / float(b * 2 * (h - 1) * w) / 2.) / 2.
            warp_loss = warp_loss + cfg.w_scales[i] * l_warp / 2.
            smooth_loss = smooth_loss + cfg.w_sm_scales[i] * l_smooth / 2.
        smooth_loss = cfg.w_smooth * smooth_loss
        mean_flow = (flows_12[0].abs().mean() + flows_10[0].abs().mean()) / 2.
        return warp_loss + smooth_loss, warp_loss, smooth_loss, mean_flow

    def forward(self, flows_12, flows_10, target):
        """flows_12 / flows_10: lists of [B,2,h,w] (finest first) from the 3-frame model
        (res['flows_fw'] / res['flows_bw']); target: [B,9,H,W] = frames 0,1,2."""
        cfg = self.cfg
        if self.pair and target.is_cuda:
            return self._forward_stacked(flows_12, flows_10, target)
        im0, im1, im2 = target[:, 0:3], target[:, 3:6], target[:, 6:9]
        warp_loss, smooth_loss = 0., 0.
        s = 1.
        for i, (f12, f10) in enumerate(zip(flows_12, flows_10)):
            if cfg.w_scales[i] == 0:
                continue
            b, _, h, w = f12.shape
            if i == 0:
                s = min(h, w)
            i1 = F.interpolate(im1, (h, w), mode='area')
            l_warp, l_smooth = 0., 0.
            for flow, im_k in ((f10, im0), (f12, im2)):
                ik = F.interpolate(im_k, (h, w), mode='area')
                rec = flow_warp(ik, flow, pad='border')
                m = border_mask(flow)
                sums = AF.PhotoSumsFunction.apply(i1, rec, m)
                photo = 0.
                if cfg.w_l1 > 0:
                    photo = photo + cfg.w_l1 * sums[0] / float(b * 3 * h * w)
                if cfg.w_ssim > 0:
                    photo = photo + cfg.w_ssim * sums[1] / float(b * 3 * (h - 2) * (w - 2))
                l_warp = l_warp + photo / (global_denominator(sums[2]) / float(b * h * w) + 1e-6)
                if cfg.w_sm_scales[i] != 0:
                    sm = AF.smooth_sums(flow, i1, 1.0 / s, cfg.alpha, 1, 0, 0)
                    l_smooth = l_smooth + (sm[0] / float(b * 2 * h * (w - 1)) / 2.) / 2. + \
                        (sm[1] / float(b * 2 * (h - 1) * w) / 2.) / 2.
            warp_loss = warp_loss + cfg.w_scales[i] * l_warp / 2.
            smooth_loss = smooth_loss + cfg.w_sm_scales[i] * l_smooth / 2.
        smooth_loss = cfg.w_smooth * smooth_loss
        mean_flow = (flows_12[0].abs().mean() + flows_10[0].abs().mean()) / 2.
        return warp_loss + smooth_loss, warp_loss, smooth_loss, mean_flow
