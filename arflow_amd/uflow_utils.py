"""GPU mirror of the hot-path part of the reference's utils/uflow_utils.py."""
import torch

from . import functional as AF


def flow_to_warp(flow):
    """utils/uflow_utils.py:6-32 -- absolute sampling coordinates.  Kept for API parity; the fused
    paths (``resample_flow``, ``mask_invalid_flow``) take the flow directly and never build it."""
    B, _, H, W = flow.shape
    xs = torch.arange(W, device=flow.device, dtype=flow.dtype).view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H, device=flow.device, dtype=flow.dtype).view(1, 1, H, 1).expand(B, 1, H, W)
    return torch.cat([xs, ys], 1) + flow


def mask_invalid(coords):
    """utils/uflow_utils.py:35-50."""
    return AF.coord_mask(coords, 0 | 2)


def mask_invalid_flow(flow):
    """mask_invalid(flow_to_warp(flow)) without materialising the coordinates."""
    return AF.coord_mask(flow, 0)


def resample(source, coords):
    """utils/uflow_utils.py:53-77."""
    return AF.warp(source, coords, pad='zeros', align_corners=True, norm=2)


def resample_flow(source, flow):
    """resample(source, flow_to_warp(flow)) in one launch."""
    return AF.warp(source, flow, pad='zeros', align_corners=True, norm=AF.NORM_UFLOW)


def compute_range_map(flow):
    """utils/uflow_utils.py:80-160."""
    return AF.splat_map(flow, 0)


def upsample(img, is_flow, scale_factor=2.0):
    """utils/uflow_utils.py:163-182 (plain ATen resize: not a hot-path kernel)."""
    out = torch.nn.functional.interpolate(img, scale_factor=scale_factor, mode='bilinear', align_corners=False)
    return out * scale_factor if is_flow else out


def downsample(img, is_flow, scale_factor=2.0):
    """utils/uflow_utils.py:185-204."""
    if not is_flow and scale_factor == 4.0 and img.shape[2] % 4 == 0 and img.shape[3] % 4 == 0 \
            and not img.requires_grad:
        return AF.down4(img)
    out = torch.nn.functional.interpolate(img, scale_factor=1 / scale_factor, mode='bilinear', align_corners=False)
    return out * (1 / scale_factor) if is_flow else out


def image_grads(image_batch, stride=1):
    """utils/uflow_utils.py:207-210."""
    return (image_batch[:, :, :, stride:] - image_batch[:, :, :, :-stride],
            image_batch[:, :, stride:] - image_batch[:, :, :-stride])


def robust_l1(x):
    """utils/uflow_utils.py:337-338."""
    return (x + 0.001 ** 2) ** 0.5


def census_loss(image_a, image_b, mask, patch_size=7):
    """utils/uflow_utils.py:282-293 -- fused forward / backward kernels."""
    return AF.CensusLossFunction.apply(image_a, image_b, mask.detach(), patch_size)
