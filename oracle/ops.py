"""Oracle ops: plain-PyTorch CPU restatement of the reference hot-path operators.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  All functions are dtype-generic (fp32 for
parity, fp64 for gradcheck) and differentiable through ordinary torch autograd so that the HIP
backward kernels can be checked against ``torch.autograd.grad`` of these functions.

Reference paths are relative to deu439/ARFlow.
"""
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# a1/a2/a3  cost volume
# --------------------------------------------------------------------------------------
def correlation(x1, x2, max_displacement=4):
    """81-channel (for d=4) cost volume, channel index = row_shift*(2d+1) + col_shift.

    out[b, i*(2d+1)+j, y, x] = (1/C) * sum_c x1[b,c,y,x] * x2[b,c,y+i-d,x+j-d]   (zero outside)

    Follows models/correlation_native.py:13-23 (``Correlation.forward``) and the identical
    arithmetic of models/uflow_model.py:53-92 (``compute_cost_volume``); the legacy CUDA
    extension models/correlation_package/correlation_cuda_kernel.cu:41-114 computes the same
    quantity for (pad,kernel,max_disp,stride1,stride2) = (4,1,4,1,1).
    """
    d = int(max_displacement)
    B, C, H, W = x1.shape
    n = 2 * d + 1
    x2p = F.pad(x2, (d, d, d, d))
    out = x1.new_empty(B, n * n, H, W)
    for i in range(n):
        rows = x2p[:, :, i:i + H]
        for j in range(n):
            out[:, i * n + j] = (x1 * rows[:, :, :, j:j + W]).sum(1) / C
    return out


def correlation_backward(gout, x1, x2, max_displacement=4):
    """Closed-form gradients of :func:`correlation` (checked against autograd in the tests).

    gx1[b,c,y,x] = (1/C) sum_{i,j} gout[b,i*n+j,y,x] * x2[b,c,y+i-d,x+j-d]
    gx2[b,c,y,x] = (1/C) sum_{i,j} gout[b,i*n+j,y-i+d,x-j+d] * x1[b,c,y-i+d,x-j+d]
    Same quantities as correlation_cuda_kernel.cu:116-207 / :209-300.
    """
    d = int(max_displacement)
    B, C, H, W = x1.shape
    n = 2 * d + 1
    x2p = F.pad(x2, (d, d, d, d))
    gx1 = torch.zeros_like(x1)
    g2p = x1.new_zeros(B, C, H + 2 * d, W + 2 * d)
    for i in range(n):
        for j in range(n):
            g = gout[:, i * n + j].unsqueeze(1)
            gx1 += g * x2p[:, :, i:i + H, j:j + W]
            g2p[:, :, i:i + H, j:j + W] += g * x1
    return gx1 / C, g2p[:, :, d:d + H, d:d + W] / C


# --------------------------------------------------------------------------------------
# bilinear sampler (restatement of torch's grid_sample, the reference's third-party kernel)
# --------------------------------------------------------------------------------------
def _pixel_grid(B, H, W, like):
    xs = torch.arange(W, dtype=like.dtype, device=like.device).view(1, 1, W).expand(B, H, W)
    ys = torch.arange(H, dtype=like.dtype, device=like.device).view(1, H, 1).expand(B, H, W)
    return xs, ys


def _unnormalize(g, size, align_corners):
    # ATen/native/GridSampler.h:27-36 (grid_sampler_unnormalize)
    if align_corners:
        return ((g + 1) / 2) * (size - 1)
    return ((g + 1) * size - 1) / 2


def _clip_border(coord, size):
    # ATen/native/GridSampler.h:58-83: clamp to [0,size-1]; the coordinate gradient is zero
    # when the input is <= 0 or >= size-1 (borders count as out of bounds).
    inside = (coord > 0) & (coord < size - 1)
    return torch.where(inside, coord, coord.detach().clamp(0, size - 1))


def sample_bilinear(src, ix, iy, pad='zeros'):
    """Bilinear lookup of ``src[B,C,H,W]`` at pixel coordinates ``ix, iy`` ([B,Ho,Wo]).

    Restates torch's ``grid_sample(mode='bilinear')`` after un-normalisation: corners from
    floor, weights ``(x_se - x)*(y_se - y)`` etc., out-of-range taps contribute zero
    (``zeros``) or the coordinates are clamped first (``border``).
    """
    B, C, H, W = src.shape
    if pad == 'border':
        ix = _clip_border(ix, W)
        iy = _clip_border(iy, H)
    elif pad != 'zeros':
        raise NotImplementedError(pad)
    x0 = torch.floor(ix.detach())
    y0 = torch.floor(iy.detach())
    x1 = x0 + 1
    y1 = y0 + 1
    wx0, wx1 = x1 - ix, ix - x0
    wy0, wy1 = y1 - iy, iy - y0
    flat = src.reshape(B, C, H * W)
    out_shape = (B, C) + tuple(ix.shape[1:])

    def tap(xi, yi):
        ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(B, 1, -1)
        v = flat.gather(2, idx.expand(B, C, idx.shape[-1])).reshape(out_shape)
        return v * ok.unsqueeze(1).to(src.dtype)

    out = tap(x0, y0) * (wx0 * wy0).unsqueeze(1)
    out = out + tap(x1, y0) * (wx1 * wy0).unsqueeze(1)
    out = out + tap(x0, y1) * (wx0 * wy1).unsqueeze(1)
    out = out + tap(x1, y1) * (wx1 * wy1).unsqueeze(1)
    return out


def sample_nearest(src, ix, iy, pad='zeros'):
    """torch's ``grid_sample(mode='nearest')`` after un-normalisation (ATen/native/cpu/GridSamplerKernel.cpp /
    GridSampler.cpp nearest branch): border padding clips the coordinate first, the index is
    ``nearbyint`` (round half to even), out-of-range indices give zero.  No gradient w.r.t. the coordinates."""
    B, C, H, W = src.shape
    ix, iy = ix.detach(), iy.detach()
    if pad == 'border':
        ix, iy = ix.clamp(0, W - 1), iy.clamp(0, H - 1)
    elif pad != 'zeros':
        raise NotImplementedError(pad)
    xi, yi = torch.round(ix), torch.round(iy)  # torch.round rounds half to even, like nearbyint
    ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
    idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(B, 1, -1)
    v = src.reshape(B, C, H * W).gather(2, idx.expand(B, C, idx.shape[-1])).reshape((B, C) + tuple(ix.shape[1:]))
    return v * ok.unsqueeze(1).to(src.dtype)


def _cubic_coeffs(t):
    """ATen/native/GridSampler.h get_cubic_upsample_coefficients (A = -0.75) -> 4 weights for taps -1, 0, +1, +2."""
    A = -0.75

    def conv1(x):
        return ((A + 2) * x - (A + 3)) * x * x + 1

    def conv2(x):
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return [conv2(t + 1.0), conv1(t), conv1(1.0 - t), conv2((1.0 - t) + 1.0)]


def sample_bicubic(src, ix, iy, pad='zeros'):
    """torch's ``grid_sample(mode='bicubic')`` after un-normalisation (ATen/native/GridSampler.h get_value_bounded,
    ATen/native/cpu/GridSamplerKernel.cpp / cuda/GridSampler.cu bicubic branch): 4 x 4 taps around floor(coordinate), each read
    at its BOUNDED position (border: the tap position is clipped; zeros: 0 outside), rows interpolated first, then the
    column.  Differentiable w.r.t. the source and -- through the coefficients only -- the coordinates."""
    B, C, H, W = src.shape
    fx, fy = torch.floor(ix).detach(), torch.floor(iy).detach()
    cx, cy = _cubic_coeffs(ix - fx), _cubic_coeffs(iy - fy)
    flat = src.reshape(B, C, H * W)
    out_shape = (B, C) + tuple(ix.shape[1:])

    def tap(x, y):
        if pad == 'border':
            x, y = x.clamp(0, W - 1), y.clamp(0, H - 1)
        elif pad != 'zeros':
            raise NotImplementedError(pad)
        ok = (x >= 0) & (x <= W - 1) & (y >= 0) & (y <= H - 1)
        idx = (y.clamp(0, H - 1) * W + x.clamp(0, W - 1)).long().reshape(B, 1, -1)
        v = flat.gather(2, idx.expand(B, C, idx.shape[-1])).reshape(out_shape)
        return v * ok.unsqueeze(1).to(src.dtype)

    rows = []
    for j in range(4):
        v = [tap(fx - 1 + i, fy - 1 + j) for i in range(4)]
        rows.append(((v[0] * cx[0].unsqueeze(1) + v[1] * cx[1].unsqueeze(1)) + v[2] * cx[2].unsqueeze(1)) + v[3] * cx[3].unsqueeze(1))
    return ((rows[0] * cy[0].unsqueeze(1) + rows[1] * cy[1].unsqueeze(1)) + rows[2] * cy[2].unsqueeze(1)) + rows[3] * cy[3].unsqueeze(1)


# --------------------------------------------------------------------------------------
# a4  flow_warp (ARFlow)
# --------------------------------------------------------------------------------------
def flow_warp(x, flow, pad='zeros', mode='bilinear', align_corners=True):
    """utils/warp_utils.py:83-90 with mesh_grid :7-13 and norm_grid :16-23.

    The grid is *always* normalised with (W-1, H-1) (warp_utils.py:21-22), also when
    ``align_corners`` is False -- the sampling positions are then scaled, not pixel exact;
    this is reproduced, not fixed.  The normalise -> un-normalise round trip is kept in the
    working precision exactly as the reference + grid_sample perform it.
    """
    if mode not in ('bilinear', 'nearest', 'bicubic'):
        raise NotImplementedError(mode)
    B, _, H, W = flow.shape
    xs, ys = _pixel_grid(B, H, W, flow)
    gx = 2.0 * (xs + flow[:, 0]) / (W - 1) - 1.0
    gy = 2.0 * (ys + flow[:, 1]) / (H - 1) - 1.0
    ix = _unnormalize(gx, x.shape[3], align_corners)
    iy = _unnormalize(gy, x.shape[2], align_corners)
    if mode == 'nearest':
        return sample_nearest(x, ix, iy, pad)
    if mode == 'bicubic':
        return sample_bicubic(x, ix, iy, pad)
    return sample_bilinear(x, ix, iy, pad)


# --------------------------------------------------------------------------------------
# a5  UFlow warp helpers
# --------------------------------------------------------------------------------------
def flow_to_warp(flow):
    """utils/uflow_utils.py:6-32 -- absolute sampling coordinates, channel 0 = x, 1 = y."""
    B, _, H, W = flow.shape
    xs, ys = _pixel_grid(B, H, W, flow)
    return torch.stack([xs, ys], 1) + flow


def mask_invalid(coords):
    """utils/uflow_utils.py:35-50 -- 1 where 0 <= x <= W-1 and 0 <= y <= H-1 (closed)."""
    H, W = coords.shape[2], coords.shape[3]
    ok = (coords[:, 0] >= 0) & (coords[:, 0] <= float(W - 1)) & \
         (coords[:, 1] >= 0) & (coords[:, 1] <= float(H - 1))
    return ok.unsqueeze(1).to(coords.dtype)


def resample(source, coords):
    """utils/uflow_utils.py:53-77 -- grid_sample(align_corners=True, zeros) at absolute coords,
    normalised with max(W-1,1) / max(H-1,1)."""
    _, _, H, W = source.shape
    gx = 2.0 * coords[:, 0] / max(W - 1, 1) - 1.0
    gy = 2.0 * coords[:, 1] / max(H - 1, 1) - 1.0
    return sample_bilinear(source, _unnormalize(gx, W, True), _unnormalize(gy, H, True), 'zeros')


# --------------------------------------------------------------------------------------
# a6  TF-style NHWC resampler
# --------------------------------------------------------------------------------------
def resampler_nhwc(data, warp):
    """utils/uflow_resampler.py:137-241 -- NHWC bilinear with floor/ceil corners, corners
    outside the image contribute zero (``safe_gather_nd`` :105-134).  Note ceil (not floor+1):
    at integer coordinates both corners coincide and the right/down weight is 0.
    data [B,H,W,C], warp [B,...,2] (x,y) -> [B,...,C].
    """
    B, H, W, C = data.shape
    wx, wy = warp[..., 0], warp[..., 1]
    fx, fy = torch.floor(wx), torch.floor(wy)
    cx, cy = torch.ceil(wx), torch.ceil(wy)
    rw, dw = wx - fx, wy - fy
    lw, uw = 1.0 - rw, 1.0 - dw
    flat = data.reshape(B, H * W, C)

    def tap(xi, yi):
        ok = (xi >= 0) & (xi <= W - 1) & (yi >= 0) & (yi <= H - 1)
        idx = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long().reshape(B, -1, 1)
        v = flat.gather(1, idx.expand(B, idx.shape[1], C)).reshape(tuple(wx.shape) + (C,))
        return v * ok.unsqueeze(-1).to(data.dtype)

    top = tap(fx, fy) * lw.unsqueeze(-1) + tap(cx, fy) * rw.unsqueeze(-1)
    bot = tap(fx, cy) * lw.unsqueeze(-1) + tap(cx, cy) * rw.unsqueeze(-1)
    return top * uw.unsqueeze(-1) + bot * dw.unsqueeze(-1)


# --------------------------------------------------------------------------------------
# a7  range map (UFlow forward splat)
# --------------------------------------------------------------------------------------
def compute_range_map(flow):
    """utils/uflow_utils.py:80-160 (duplicate at utils/warp_utils.py:158-239).

    Every source pixel p adds its four bilinear weights to the integer pixels around
    p + flow(p); taps outside the image are dropped.  Returns [B,1,H,W].
    """
    B, _, H, W = flow.shape
    xs, ys = _pixel_grid(B, H, W, flow)
    cx, cy = xs + flow[:, 0], ys + flow[:, 1]
    fx, fy = torch.floor(cx), torch.floor(cy)
    ox, oy = cx - fx, cy - fy
    out = flow.new_zeros(B, H * W)
    for di in range(2):
        for dj in range(2):
            yi, xj = fy + di, fx + dj
            wi = oy if di else 1.0 - oy
            wj = ox if dj else 1.0 - ox
            ok = (yi >= 0) & (yi < H) & (xj >= 0) & (xj < W)
            idx = (yi.clamp(0, H - 1) * W + xj.clamp(0, W - 1)).long().reshape(B, -1)
            out.scatter_add_(1, idx, (wi * wj * ok.to(flow.dtype)).reshape(B, -1))
    return out.view(B, 1, H, W)


# --------------------------------------------------------------------------------------
# a8  ARFlow occlusion / border masks
# --------------------------------------------------------------------------------------
def get_corresponding_map(data):
    """utils/warp_utils.py:26-80 -- splat with *clamped* indices; a tap whose un-clamped
    coordinate left the image gets weight 0 (``values[invalid] = 0`` :74)."""
    B, _, H, W = data.shape
    x, y = data[:, 0].reshape(B, -1), data[:, 1].reshape(B, -1)
    x1, y1 = torch.floor(x), torch.floor(y)
    x0, y0 = x1 + 1, y1 + 1
    xf, yf = x1.clamp(0, W - 1), y1.clamp(0, H - 1)
    xc, yc = x0.clamp(0, W - 1), y0.clamp(0, H - 1)
    out = data.new_zeros(B, H * W)
    for xi, xraw in ((xc, x0), (xf, x1)):
        for yi, yraw in ((yc, y0), (yf, y1)):
            w = (1 - (x - xi).abs()) * (1 - (y - yi).abs())
            bad = (xi != xraw) | (yi != yraw)
            w = torch.where(bad, torch.zeros_like(w), w)
            out.scatter_add_(1, (xi + yi * W).long(), w)
    return out.view(B, 1, H, W)


def get_occu_mask_backward(flow21, th=0.2):
    """utils/warp_utils.py:103-116 -- 1 at occluded pixels."""
    B, _, H, W = flow21.shape
    xs, ys = _pixel_grid(B, H, W, flow21)
    corr_map = get_corresponding_map(torch.stack([xs, ys], 1) + flow21)
    if th > 0:
        return (corr_map.clamp(0., 1.) < th).to(flow21.dtype)
    return 1. - corr_map.clamp(0., 1.).detach()


def get_occu_mask_bidirection(flow12, flow21, scale=0.01, bias=0.5):
    """utils/warp_utils.py:93-100 -- forward/backward consistency check."""
    f21w = flow_warp(flow21, flow12, pad='zeros')
    diff = flow12 + f21w
    mag = (flow12 * flow12).sum(1, keepdim=True) + (f21w * f21w).sum(1, keepdim=True)
    return ((diff * diff).sum(1, keepdim=True) > scale * mag + bias).to(flow12.dtype)


def border_mask(flow):
    """utils/warp_utils.py:119-134 -- strict 0 < x' < W-1 and 0 < y' < H-1."""
    B, _, H, W = flow.shape
    xs, ys = _pixel_grid(B, H, W, flow)
    xp, yp = xs + flow[:, 0], ys + flow[:, 1]
    ok = (xp > 0.0) & (xp < W - 1.0) & (yp > 0.0) & (yp < H - 1.0)
    return ok.view(B, 1, H, W).to(flow.dtype)


# --------------------------------------------------------------------------------------
# a9  SSIM
# --------------------------------------------------------------------------------------
def _box_mean_valid(t, k):
    return F.avg_pool2d(t, k, 1, 0)


def ssim(x, y, md=1):
    """losses/loss_blocks.py:65-84 -- un-padded (2md+1)^2 box SSIM, clamp((1-SSIM)/2, 0, 1)."""
    k = 2 * md + 1
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    mx, my = _box_mean_valid(x, k), _box_mean_valid(y, k)
    sx = _box_mean_valid(x * x, k) - mx * mx
    sy = _box_mean_valid(y * y, k) - my * my
    sxy = _box_mean_valid(x * y, k) - mx * my
    num = (2 * mx * my + c1) * (2 * sxy + c2)
    den = (mx * mx + my * my + c1) * (sx + sy + c2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


# --------------------------------------------------------------------------------------
# a10 / a13  census (ternary) transform
# --------------------------------------------------------------------------------------
def rgb_to_grayscale(image):
    """utils/uflow_utils.py:227-231, losses/loss_blocks.py:15-19."""
    return (image[:, 0] * 0.2989 + image[:, 1] * 0.5870 + image[:, 2] * 0.1140).unsqueeze(1)


def _neighbour_stack(gray, r):
    """[B,1,H,W] -> [B,(2r+1)^2,H,W]; channel k=(dy+r)*(2r+1)+(dx+r) holds gray[y+dy, x+dx]
    with zero padding -- what the identity-kernel conv2d of uflow_utils.py:255-257 /
    loss_blocks.py:23-26 produces."""
    H, W = gray.shape[2], gray.shape[3]
    gp = F.pad(gray, (r, r, r, r))
    return torch.cat([gp[:, :, i:i + H, j:j + W]
                      for i in range(2 * r + 1) for j in range(2 * r + 1)], 1)


def census_transform(image, patch_size):
    """utils/uflow_utils.py:241-261."""
    inten = rgb_to_grayscale(image) * 255
    diff = _neighbour_stack(inten, patch_size // 2) - inten
    return diff / torch.sqrt(.81 + diff * diff)


def soft_hamming(a, b, thresh=.1):
    """utils/uflow_utils.py:264-279 (sum over the patch channels)."""
    sq = (a - b) ** 2
    return (sq / (thresh + sq)).sum(1, keepdim=True)


def zero_mask_border(mask, patch_size):
    """utils/uflow_utils.py:234-238."""
    p = patch_size // 2
    return F.pad(mask[:, :, p:-p, p:-p], (p, p, p, p))


def abs_robust_loss(diff, eps=0.01, q=0.4):
    """utils/uflow_utils.py:213-214, losses/loss_blocks.py:5-6 (penalty_ddflow)."""
    return (diff.abs() + eps) ** q


def census_loss(image_a, image_b, mask, patch_size=7):
    """utils/uflow_utils.py:282-293."""
    ham = soft_hamming(census_transform(image_a, patch_size), census_transform(image_b, patch_size))
    pm = zero_mask_border(mask, patch_size)
    return (abs_robust_loss(ham) * pm).sum() / (pm.detach().sum() + 1e-6)


def ternary_loss(im, im_warp, max_distance=1, sum_dist=False):
    """losses/loss_blocks.py:12-62 -> (dist [B,1,H,W], mask [B,1,H,W])."""
    k = 2 * max_distance + 1
    t1, t2 = census_transform(im, k), census_transform(im_warp, k)
    sq = (t1 - t2) ** 2
    dn = sq / (0.1 + sq)
    dist = dn.sum(1, keepdim=True) if sum_dist else dn.mean(1, keepdim=True)
    B, _, H, W = im.shape
    m = max_distance
    mask = F.pad(im.new_ones(B, 1, H - 2 * m, W - 2 * m), (m, m, m, m))
    return dist, mask


# --------------------------------------------------------------------------------------
# a11  edge-aware smoothness (ARFlow blocks)
# --------------------------------------------------------------------------------------
def penalty_uflow(x):
    """losses/loss_blocks.py:8-9."""
    return torch.sqrt(x * x + 0.001 ** 2)


def gradient(data):
    """losses/loss_blocks.py:87-90 -> (d/dx, d/dy) forward differences."""
    return data[:, :, :, 1:] - data[:, :, :, :-1], data[:, :, 1:] - data[:, :, :-1]


def _edge_weights(image, alpha):
    ix, iy = gradient(image)
    return (torch.exp(-ix.abs().mean(1, keepdim=True) * alpha),
            torch.exp(-iy.abs().mean(1, keepdim=True) * alpha))


def smooth_grad_1st(flo, image, alpha, penalty='abs'):
    """losses/loss_blocks.py:93-109."""
    wx, wy = _edge_weights(image, alpha)
    dx, dy = gradient(flo)
    if penalty == 'abs':
        lx, ly = wx * dx.abs() / 2., wy * dy.abs() / 2.
    elif penalty == 'uflow':
        lx, ly = wx * penalty_uflow(dx) / 2., wy * penalty_uflow(dy) / 2.
    else:
        raise NotImplementedError(penalty)
    return lx.mean() / 2. + ly.mean() / 2.


def smooth_grad_2nd(flo, image, alpha):
    """losses/loss_blocks.py:112-124."""
    wx, wy = _edge_weights(image, alpha)
    dx, dy = gradient(flo)
    dx2, _ = gradient(dx)
    _, dy2 = gradient(dy)
    return (wx[:, :, :, 1:] * dx2.abs()).mean() / 2. + (wy[:, :, 1:, :] * dy2.abs()).mean() / 2.


# --------------------------------------------------------------------------------------
# a14  UFlow resize helpers
# --------------------------------------------------------------------------------------
def upsample(img, is_flow, scale_factor=2.0):
    """utils/uflow_utils.py:163-182 -- bilinear, align_corners=False."""
    out = F.interpolate(img, scale_factor=scale_factor, mode='bilinear', align_corners=False)
    return out * scale_factor if is_flow else out


def downsample(img, is_flow, scale_factor=2.0):
    """utils/uflow_utils.py:185-204 -- bilinear at 1/scale, align_corners=False."""
    out = F.interpolate(img, scale_factor=1 / scale_factor, mode='bilinear', align_corners=False)
    return out * (1 / scale_factor) if is_flow else out


def downsample4_explicit(img):
    """What ``downsample(img, False, 4.0)`` evaluates to when H, W are multiples of 4: the mean
    of the central 2x2 of every 4x4 block (verified against F.interpolate in
    tests)."""
    return 0.25 * (img[:, :, 1::4, 1::4] + img[:, :, 1::4, 2::4] +
                   img[:, :, 2::4, 1::4] + img[:, :, 2::4, 2::4])


def image_grads(image_batch, stride=1):
    """utils/uflow_utils.py:207-210."""
    return (image_batch[:, :, :, stride:] - image_batch[:, :, :, :-stride],
            image_batch[:, :, stride:] - image_batch[:, :, :-stride])


def robust_l1(x):
    """utils/uflow_utils.py:337-338."""
    return (x + 0.001 ** 2) ** 0.5


# --------------------------------------------------------------------------------------
# a16  feature normalisation
# --------------------------------------------------------------------------------------
def normalize_features_joint(features_list):
    """models/pwclite_uflow.py:30-38 -- per-sample moments of the channel-concatenated pair,
    unbiased variance, eps 1e-16 under the sqrt."""
    feats = torch.cat(features_list, 1)
    mean = feats.mean(dim=(1, 2, 3), keepdim=True)
    var = feats.var(dim=(1, 2, 3), keepdim=True)
    std = torch.sqrt(var + 1e-16)
    return [(f - mean) / std for f in features_list]


def normalize_features_uflow(feature_list, normalize=True, center=True,
                             moments_across_channels=True, moments_across_images=True):
    """models/uflow_model.py:8-50 -- per-tensor moments, optionally averaged across the images."""
    dim = [1, 2, 3] if moments_across_channels else [2, 3]
    means = [f.mean(dim=dim, keepdim=True) for f in feature_list]
    vars_ = [f.var(dim=dim, keepdim=True) for f in feature_list]
    if moments_across_images:
        means = [torch.stack(means).mean(0)] * len(means)
        vars_ = [torch.stack(vars_).mean(0)] * len(vars_)
    stds = [torch.sqrt(v + 1e-16) for v in vars_]
    if center:
        feature_list = [f - m for f, m in zip(feature_list, means)]
    if normalize:
        feature_list = [f / s for f, s in zip(feature_list, stds)]
    return feature_list


# --------------------------------------------------------------------------------------
# a3  correlation with the CUDA extension's full parameter set
# --------------------------------------------------------------------------------------
def correlation_general(x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """Forward of models/correlation_package/correlation_cuda_kernel.cu:41-114 with the output geometry of
    correlation_cuda.cc:19-34, restated on NCHW tensors (differentiable: torch autograd of this expression is the
    gradient).  PARITY UNPINNED for non-default parameters: the CUDA extension cannot be built here (no nvcc) and no
    caller in the reference passes anything but (pad=d, kernel=1, stride1=stride2=1), which IS pinned through
    models/correlation_native.py (``correlation`` above; tests compare the two on the default)."""
    B, C, H, W = x1.shape
    kr = (kernel_size - 1) // 2
    dr = max_displacement // stride2
    border = kr + max_displacement
    ph, pw = H + 2 * pad_size, W + 2 * pad_size
    ho = -(-(ph - 2 * border) // stride1)
    wo = -(-(pw - 2 * border) // stride1)
    # pad generously so that every index the loops touch exists (the kernel reads inside its padded buffers)
    ext = max_displacement + kr + dr * stride2 + stride1
    p1 = F.pad(x1, [pad_size + ext] * 4)
    p2 = F.pad(x2, [pad_size + ext] * 4)
    ys = torch.arange(ho) * stride1 + max_displacement + ext
    xs = torch.arange(wo) * stride1 + max_displacement + ext
    outs = []
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            acc = 0.
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    a = p1[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    b = p2[:, :, (ys + j + tj * stride2)[:, None], (xs + i + ti * stride2)[None, :]]
                    acc = acc + (a * b).sum(1)
            outs.append(acc / float(kernel_size * kernel_size * C))
    return torch.stack(outs, 1)
