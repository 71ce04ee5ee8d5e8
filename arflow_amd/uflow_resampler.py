"""GPU mirror of utils/uflow_resampler.py (TF-style NHWC resampler; unused by the reference's own
callers, named by the north star).  Thin NHWC shim over the warp kernel."""
from . import functional as AF


def resampler(data, warp):
    """utils/uflow_resampler.py:137-152.  data [B,H,W,C], warp [B,H',W',2] (x,y) -> [B,H',W',C]."""
    src = data.permute(0, 3, 1, 2).contiguous()
    coords = warp.permute(0, 3, 1, 2).contiguous()
    out = AF.warp(src, coords, pad='zeros', align_corners=True, norm=2)
    return out.permute(0, 2, 3, 1).contiguous()


def resampler_with_unstacked_warp(data, warp_x, warp_y, safe=True):
    """utils/uflow_resampler.py:155-241."""
    import torch
    return resampler(data, torch.stack([warp_x, warp_y], -1))
