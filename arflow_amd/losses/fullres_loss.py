"""FullResLoss on the gfx950 kernels (losses/fullres_loss.py:8-107; SURVEY section 8f "next" row).
Same constructor, inputs and 4-tuple result."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF
from ..loss_blocks import SSIM, TernaryLoss, penalty_ddflow, smooth_grad_1st
from ..ddp import global_denominator, world_size
from ..warp_utils import (border_mask, compute_range_map, flow_warp, get_occu_mask_backward,
                          get_occu_mask_bidirection)


class FullResLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def loss_photometric(self, im, recons, mask):
        cfg = self.cfg
        loss = 0
        if cfg.w_l1 > 0:
            loss = loss + torch.sum(cfg.w_l1 * (im - recons).abs() * mask) / (global_denominator(torch.sum(mask)) + 1e-6 / world_size())
        if cfg.w_ssim > 0:
            # shape mismatch in the reference too (fullres_loss.py:22): un-padded SSIM x full mask
            loss = loss + torch.sum(cfg.w_ssim * SSIM(recons, im) * mask) / (global_denominator(torch.sum(mask)) + 1e-6 / world_size())
        if cfg.w_ternary > 0:
            dist, valid = TernaryLoss(im, recons, max_distance=cfg.ternary_distance, sum_dist=True)
            m = torch.detach(valid * mask)
            loss = loss + torch.sum(cfg.w_ternary * penalty_ddflow(dist) * m) / (global_denominator(torch.sum(m)) + 1e-6 / world_size())
        return loss

    def loss_smooth(self, flow, im):
        if 'smooth_2nd' in self.cfg and self.cfg.smooth_2nd:
            raise TypeError("smooth_grad_2nd() got an unexpected keyword argument 'penalty'")  # fullres_loss.py:33-37
        return smooth_grad_1st(flow, im, self.cfg.alpha, penalty='uflow') * 2.0

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
            recons = flow_warp(im_b.detach(), f_ab, pad=cfg.warp_pad, align_corners=cfg.align_corners)
            bmask = border_mask(f_ab)
            if cfg.occ_type == 'wang':
                occ = 1. - get_occu_mask_backward(f_ba, th=cfg.wang_thr)
            elif cfg.occ_type == 'wang1':
                occ = torch.clamp(compute_range_map(f_ba), min=0., max=1.)
            elif cfg.occ_type == 'brox':
                occ = 1. - get_occu_mask_bidirection(f_ab, f_ba)
            elif cfg.occ_type == 'none':
                occ = torch.ones_like(f_ab)
            else:
                raise NotImplementedError(cfg.occ_type)
            loss_warp = loss_warp + self.loss_photometric(im_a, recons, occ * bmask)
            h, w = f_ab2.shape[2:]
            im_s = F.interpolate(im_a, (h, w), mode='bilinear', align_corners=cfg.align_corners)
            loss_smooth = loss_smooth + self.loss_smooth(f_ab2, im_s.detach())
        return loss_warp + cfg.w_smooth * loss_smooth, loss_warp, loss_smooth, output[0].abs().mean()
