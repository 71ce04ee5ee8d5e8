"""GPU mirror of losses/loss_blocks.py."""
import torch
import torch.nn.functional as F

from . import functional as AF


def penalty_ddflow(diff, eps=0.01, q=0.4):
    """losses/loss_blocks.py:5-6."""
    return torch.pow(torch.abs(diff) + eps, q)


def penalty_uflow(x):
    """losses/loss_blocks.py:8-9."""
    return torch.sqrt(torch.pow(x, 2.0) + 0.001 ** 2)


def TernaryLoss(im, im_warp, max_distance=1, sum_dist=False):
    """losses/loss_blocks.py:12-62 -> (dist, mask)."""
    if not 1 <= max_distance <= 16:
        raise ValueError('max_distance must be in 1..16')  # 1..3: the tiled kernels; beyond: the plain ones
    dist = AF.TernaryDistFunction.apply(im, im_warp, max_distance)
    if not sum_dist:
        dist = dist / float((2 * max_distance + 1) ** 2)
    n, _, h, w = im.shape
    m = max_distance
    mask = F.pad(torch.ones(n, 1, h - 2 * m, w - 2 * m, device=im.device, dtype=im.dtype), [m] * 4)
    return dist, mask


def SSIM(x, y, md=1):
    """losses/loss_blocks.py:65-84."""
    if md != 1:
        return AF.SSIMAnyFunction.apply(x, y, md)  # any window; md = 1 (every shipped config) runs the tiled kernel
    return AF.SSIMFunction.apply(x, y)


def gradient(data):
    """losses/loss_blocks.py:87-90."""
    return data[:, :, :, 1:] - data[:, :, :, :-1], data[:, :, 1:] - data[:, :, :-1]


def _mean_counts(flo, order):
    b, c, h, w = flo.shape
    return float(b * c * h * (w - order)), float(b * c * (h - order) * w)


def smooth_grad_1st(flo, image, alpha, penalty='abs'):
    """losses/loss_blocks.py:93-109."""
    if penalty not in ('abs', 'uflow'):
        raise NotImplementedError()
    s = AF.smooth_sums(flo, image, 1.0, alpha, 1, 0, 0 if penalty == 'abs' else 1)
    nx, ny = _mean_counts(flo, 1)
    return (s[0] / nx / 2.) / 2. + (s[1] / ny / 2.) / 2.


def smooth_grad_2nd(flo, image, alpha):
    """losses/loss_blocks.py:112-124."""
    s = AF.smooth_sums(flo, image, 1.0, alpha, 2, 0, 0)
    nx, ny = _mean_counts(flo, 2)
    return (s[0] / nx) / 2. + (s[1] / ny) / 2.
