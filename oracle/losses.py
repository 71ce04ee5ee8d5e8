"""Oracle loss modules: plain-PyTorch CPU restatement of the reference's unsupervised losses.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``cfg`` is any object with attribute access
and ``in`` support (arflow_amd.config.AttrDict, or the dict subclass the tests use).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _smooth_terms_uflow(im_small, flow, edge_constant, order):
    """One direction of losses/uflow_loss.py:62-102."""
    if order == 1:
        igx, igy = ops.image_grads(im_small)
        fgx, fgy = ops.image_grads(flow)
    elif order == 2:
        igx, igy = ops.image_grads(im_small, stride=2)
        fx, fy = ops.image_grads(flow)
        fgx, _ = ops.image_grads(fx)
        _, fgy = ops.image_grads(fy)
    else:
        raise NotImplementedError(order)
    wx = torch.exp(-(edge_constant * igx).abs().mean(1, keepdim=True))
    wy = torch.exp(-(edge_constant * igy).abs().mean(1, keepdim=True))
    return ((wx * ops.robust_l1(fgx ** 2)).mean() + (wy * ops.robust_l1(fgy ** 2)).mean()) / 2.


class UFlowLoss(nn.Module):
    """losses/uflow_loss.py:8-109.  Returns (total, l_ph, l_sm, mean|flow_0|, mask1)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def forward(self, output, target):
        cfg = self.cfg
        f12_0, f21_0 = output[0][:, 0:2], output[0][:, 2:4]
        f12_2, f21_2 = output[2][:, 0:2], output[2][:, 2:4]
        im1, im2 = target[:, :3], target[:, 3:]
        pairs = [(im1, im2, f12_0, f21_2, f12_2)]
        if cfg.with_bk:
            pairs.append((im2, im1, f21_0, f12_2, f21_2))

        loss_warp, loss_smooth, masks = 0., 0., []
        for im_a, im_b, f_ab0, f_ba2, f_ab2 in pairs:
            warp = ops.flow_to_warp(f_ab0)
            recons = ops.resample(im_b.detach(), warp)
            valid = ops.mask_invalid(warp)
            occ = torch.clamp(ops.compute_range_map(f_ba2), 0., 1.)
            occ = ops.upsample(occ, is_flow=False, scale_factor=4.0)
            mask = (occ * valid).detach()
            masks.append(mask)
            loss_warp = loss_warp + cfg.w_census * ops.census_loss(im_a, recons, mask)
            im_small = ops.downsample(im_a, is_flow=False, scale_factor=4.0).detach()
            loss_smooth = loss_smooth + cfg.w_smooth * _smooth_terms_uflow(
                im_small, f_ab2, cfg.edge_constant, cfg.smooth_order)
        return loss_warp + loss_smooth, loss_warp, loss_smooth, output[0].abs().mean(), masks[0]


class unFlowLoss(nn.Module):
    """losses/flow_loss.py:8-114 (ARFlow pyramid loss).  Returns (total, warp, smooth, mean|flow|)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def loss_photomatric(self, im_scaled, im_recons, occu_mask):
        cfg = self.cfg
        terms = []
        if cfg.w_l1 > 0:
            terms.append(cfg.w_l1 * (im_scaled - im_recons).abs() * occu_mask)
        if cfg.w_ssim > 0:
            terms.append(cfg.w_ssim * ops.ssim(im_recons * occu_mask, im_scaled * occu_mask))
        if cfg.w_ternary > 0:
            # the reference multiplies a float with the (dist, mask) tuple here and raises
            # TypeError (losses/flow_loss.py:23-25 vs loss_blocks.py:62); mirror the failure.
            raise TypeError("unFlowLoss with w_ternary > 0 is broken in the reference")
        return sum(t.mean() for t in terms) / occu_mask.mean()

    def loss_smooth(self, flow, im_scaled):
        if 'smooth_2nd' in self.cfg and self.cfg.smooth_2nd:
            return ops.smooth_grad_2nd(flow, im_scaled, self.cfg.alpha)
        return ops.smooth_grad_1st(flow, im_scaled, self.cfg.alpha)

    def forward(self, output, target):
        cfg = self.cfg
        im1_o, im2_o = target[:, :3], target[:, 3:]
        warp_losses, smooth_losses = [], []
        self.pyramid_occu_mask1, self.pyramid_occu_mask2 = [], []
        s = 1.
        for i, flow in enumerate(output):
            if cfg.w_scales[i] == 0:
                warp_losses.append(0)
                smooth_losses.append(0)
                continue
            _, _, h, w = flow.shape
            im1 = F.interpolate(im1_o, (h, w), mode='area')
            im2 = F.interpolate(im2_o, (h, w), mode='area')
            rec1 = ops.flow_warp(im2, flow[:, :2], pad=cfg.warp_pad)
            rec2 = ops.flow_warp(im1, flow[:, 2:], pad=cfg.warp_pad)
            if i == 0:
                if cfg.occ_from_back:
                    m1 = 1 - ops.get_occu_mask_backward(flow[:, 2:], th=0.2)
                    m2 = 1 - ops.get_occu_mask_backward(flow[:, :2], th=0.2)
                else:
                    m1 = 1 - ops.get_occu_mask_bidirection(flow[:, :2], flow[:, 2:])
                    m2 = 1 - ops.get_occu_mask_bidirection(flow[:, 2:], flow[:, :2])
            else:
                m1 = F.interpolate(self.pyramid_occu_mask1[0], (h, w), mode='nearest')
                m2 = F.interpolate(self.pyramid_occu_mask2[0], (h, w), mode='nearest')
            self.pyramid_occu_mask1.append(m1)
            self.pyramid_occu_mask2.append(m2)

            l_warp = self.loss_photomatric(im1, rec1, m1)
            if i == 0:
                s = min(h, w)
            l_smooth = self.loss_smooth(flow[:, :2] / s, im1)
            if cfg.with_bk:
                l_warp = (l_warp + self.loss_photomatric(im2, rec2, m2)) / 2.
                l_smooth = (l_smooth + self.loss_smooth(flow[:, 2:] / s, im2)) / 2.
            warp_losses.append(l_warp)
            smooth_losses.append(l_smooth)

        warp_loss = sum(l * w for l, w in zip(warp_losses, cfg.w_scales))
        smooth_loss = cfg.w_smooth * sum(l * w for l, w in zip(smooth_losses, cfg.w_sm_scales))
        return warp_loss + smooth_loss, warp_loss, smooth_loss, output[0].abs().mean()


class FullResLoss(nn.Module):
    """losses/fullres_loss.py:8-107.  Returns (total, warp, smooth, mean|flow_0|)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def loss_photometric(self, im, recons, mask):
        cfg = self.cfg
        loss = 0
        if cfg.w_l1 > 0:
            loss = loss + (cfg.w_l1 * (im - recons).abs() * mask).sum() / (mask.sum() + 1e-6)
        if cfg.w_ssim > 0:
            # shape-mismatches in the reference (un-padded SSIM times full-size mask,
            # losses/fullres_loss.py:22); kept so the same error surfaces.
            loss = loss + (cfg.w_ssim * ops.ssim(recons, im) * mask).sum() / (mask.sum() + 1e-6)
        if cfg.w_ternary > 0:
            dist, valid = ops.ternary_loss(im, recons, max_distance=cfg.ternary_distance, sum_dist=True)
            m = (valid * mask).detach()
            loss = loss + (cfg.w_ternary * ops.abs_robust_loss(dist) * m).sum() / (m.sum() + 1e-6)
        return loss

    def loss_smooth(self, flow, im):
        if 'smooth_2nd' in self.cfg and self.cfg.smooth_2nd:
            # the reference passes penalty= to smooth_grad_2nd, which does not accept it
            # (losses/fullres_loss.py:37 vs loss_blocks.py:112) -> TypeError there as well.
            raise TypeError("smooth_grad_2nd() got an unexpected keyword argument 'penalty'")
        return ops.smooth_grad_1st(flow, im, self.cfg.alpha, penalty='uflow') * 2.0

    def forward(self, output, target):
        cfg = self.cfg
        f12_0, f21_0 = output[0][:, 0:2], output[0][:, 2:4]
        f12_2, f21_2 = output[2][:, 0:2], output[2][:, 2:4]
        im1, im2 = target[:, :3], target[:, 3:]
        dirs = [(im1, im2, f12_0, f21_0, f12_2)]
        if cfg.with_bk:
            dirs.append((im2, im1, f21_0, f12_0, f21_2))
        loss_warp, loss_smooth = 0., 0.
        for im_a, im_b, f_ab, f_ba, f_ab2 in dirs:
            recons = ops.flow_warp(im_b.detach(), f_ab, pad=cfg.warp_pad, align_corners=cfg.align_corners)
            bmask = ops.border_mask(f_ab)
            if cfg.occ_type == 'wang':
                occ = 1. - ops.get_occu_mask_backward(f_ba, th=cfg.wang_thr)
            elif cfg.occ_type == 'wang1':
                occ = torch.clamp(ops.compute_range_map(f_ba), 0., 1.)
            elif cfg.occ_type == 'brox':
                occ = 1. - ops.get_occu_mask_bidirection(f_ab, f_ba)
            elif cfg.occ_type == 'none':
                occ = torch.ones_like(f_ab)
            else:
                raise NotImplementedError(cfg.occ_type)
            loss_warp = loss_warp + self.loss_photometric(im_a, recons, occ * bmask)
            h, w = f_ab2.shape[2:]
            im_s = F.interpolate(im_a, (h, w), mode='bilinear', align_corners=cfg.align_corners)
            loss_smooth = loss_smooth + self.loss_smooth(f_ab2, im_s.detach())
        return loss_warp + cfg.w_smooth * loss_smooth, loss_warp, loss_smooth, output[0].abs().mean()


class MvLoss(nn.Module):
    """CPU restatement of arflow_amd/losses/mv_loss.py -- the build-defined multi-view objective of
    SURVEY App. B-10, composed of oracle functions only (flow_warp border, border_mask, L1 + SSIM,
    smooth_grad_1st)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def forward(self, flows_12, flows_10, target):
        cfg = self.cfg
        im0, im1, im2 = target[:, 0:3], target[:, 3:6], target[:, 6:9]
        warp_loss, smooth_loss = 0., 0.
        s = 1.
        for i, (f12, f10) in enumerate(zip(flows_12, flows_10)):
            if cfg.w_scales[i] == 0:
                continue
            _, _, h, w = f12.shape
            if i == 0:
                s = min(h, w)
            i1 = F.interpolate(im1, (h, w), mode='area')
            l_warp, l_smooth = 0., 0.
            for flow, im_k in ((f10, im0), (f12, im2)):
                ik = F.interpolate(im_k, (h, w), mode='area')
                rec = ops.flow_warp(ik, flow, pad='border')
                m = ops.border_mask(flow)
                photo = 0.
                if cfg.w_l1 > 0:
                    photo = photo + (cfg.w_l1 * (i1 - rec).abs() * m).mean()
                if cfg.w_ssim > 0:
                    photo = photo + (cfg.w_ssim * ops.ssim(rec * m, i1 * m)).mean()
                l_warp = l_warp + photo / (m.mean() + 1e-6)
                if cfg.w_sm_scales[i] != 0:
                    l_smooth = l_smooth + ops.smooth_grad_1st(flow / s, i1, cfg.alpha)
            warp_loss = warp_loss + cfg.w_scales[i] * l_warp / 2.
            smooth_loss = smooth_loss + cfg.w_sm_scales[i] * l_smooth / 2.
        smooth_loss = cfg.w_smooth * smooth_loss
        mean_flow = (flows_12[0].abs().mean() + flows_10[0].abs().mean()) / 2.
        return warp_loss + smooth_loss, warp_loss, smooth_loss, mean_flow
