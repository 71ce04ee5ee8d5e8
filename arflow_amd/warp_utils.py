"""GPU mirror of the reference's utils/warp_utils.py (same names, arguments and return shapes)."""
import torch

from . import functional as AF


def flow_warp(x, flow, pad='zeros', mode='bilinear', align_corners=True, storage_dtype=None):
    """utils/warp_utils.py:83-90.  storage_dtype=torch.bfloat16 (opt-in, not in the reference's signature): keep the
    warped source as bf16 in HBM, fp32 arithmetic / output / gradients."""
    if mode == 'nearest':
        return AF.warp_nearest(x, flow, pad=pad, align_corners=align_corners, norm=AF.NORM_ARFLOW)
    if mode == 'bicubic':
        return AF.warp_bicubic(x, flow, pad=pad, align_corners=align_corners, norm=AF.NORM_ARFLOW)
    if mode != 'bilinear':
        raise ValueError("mode must be 'bilinear', 'nearest' or 'bicubic' (F.grid_sample's modes for 4-D input)")
    return AF.warp(x, flow, pad=pad, align_corners=align_corners, norm=AF.NORM_ARFLOW, storage=storage_dtype)


def get_corresponding_map(data):
    """utils/warp_utils.py:26-80; ``data`` holds absolute coordinates [B,2,H,W]."""
    return AF.splat_map(data, 1 | 2)


def get_occu_mask_backward(flow21, th=0.2):
    """utils/warp_utils.py:103-116 -- 1 at occluded pixels."""
    corr_map = AF.splat_map(flow21, 1)
    if th > 0:
        return (corr_map.clamp(min=0., max=1.) < th).float()
    return 1. - corr_map.clamp(min=0., max=1.)


def get_occu_mask_bidirection(flow12, flow21, scale=0.01, bias=0.5):
    """utils/warp_utils.py:93-100."""
    return AF.occ_bidir(flow12, flow21, scale, bias)


def border_mask(flow):
    """utils/warp_utils.py:119-134."""
    return AF.coord_mask(flow, 1)


def compute_range_map(flow):
    """utils/warp_utils.py:158-239 (NCHW in, [B,1,H,W] out, as the reference's loss calls it)."""
    return AF.splat_map(flow, 0)
