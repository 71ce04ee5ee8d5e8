"""torch.autograd.Function wrappers over the C ABI (libarflow_hip.so).

Every op here runs ONLY on CUDA(ROCm) fp32 tensors through the hand-written gfx950 kernels; a CPU
tensor, a wrong dtype or a missing library raises.  Outputs are allocated with the torch caching
allocator and kernels are enqueued on ``torch.cuda.current_stream()`` -- no synchronisation.
"""
import torch

from . import _lib
from . import ddp as _ddp

PAD = {'zeros': 0, 'border': 1}
NORM_ARFLOW, NORM_UFLOW = 0, 1


def _need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.ArflowHipError('arflow_amd ops run on the GPU only (got a %s tensor); there is no CPU '
                                      'fallback -- the CPU oracle lives in oracle/ and is test-only' % t.device)
        if t.dtype != torch.float32:
            raise _lib.ArflowHipError('arflow_amd ops are fp32 only (got %s)' % t.dtype)


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _flow_view(flow):
    """Return (tensor, batch_stride) so that a [B,2,H,W] slice of a wider tensor is used in place."""
    B, two, H, W = flow.shape
    assert two == 2, 'flow must have 2 channels'
    st = flow.stride()
    if st[3] == 1 and st[2] == W and st[1] == H * W and (B == 1 or st[0] >= 2 * H * W):
        return flow, (st[0] if B > 1 else 2 * H * W)
    flow = flow.contiguous()
    return flow, 2 * H * W


_timing = None  # list of (name, key, start_event, end_event) while kernel timing is on


_paused = False


def start_kernel_timing():
    """Bracket every C-ABI launch with HIP events recorded on the launch stream (torch's current
    stream).  Used by bench.py for the per-kernel roofline; off by default."""
    global _timing
    _timing = []


def pause_kernel_timing(paused):
    """Keep the records but stop / resume bracketing launches (bench.py samples a subset of its steps: two
    event records per launch cost ~10 us of host time each, which a launch-bound step feels)."""
    global _paused
    _paused = bool(paused)


def stop_kernel_timing():
    """-> {(name, shape_key): [ms, ...]} ; synchronises."""
    global _timing
    rec, _timing = _timing or [], None
    torch.cuda.synchronize()
    out = {}
    for name, key, e0, e1 in rec:
        out.setdefault((name, key), []).append(e0.elapsed_time(e1))
    return out


SUM_COLS = 4  # ARFLOW_SUM_COLS of include/arflow_hip.h


def _new_sums(device, B, H, W):
    """Per-workgroup partial rows of a reduction kernel over a [B, *, H, W] problem (every row is written by the
    call: no initialisation needed)."""
    rows = _lib.load().arflow_sums_rows(int(B), int(H), int(W))
    if rows <= 0:
        _lib.check(rows, 'arflow_sums_rows')
    return torch.empty(rows * SUM_COLS, device=device, dtype=torch.float32)


def _fold_sums(buf, k):
    """Partial rows -> the k reduced quantities (one tiny reduction kernel, fixed order: reproducible)."""
    return buf.view(-1, SUM_COLS)[:, :k].sum(0)


def _call(name, *args, key=None):
    lib = _lib.load()
    if _timing is not None and key is not None and not _paused:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        _timing.append((name, key, e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    _lib.check(rc, name)
    _lib.poll_stale_error(lib, name)


# ------------------------------------------------------------------------------------------------
class CorrelationFunction(torch.autograd.Function):
    """Cost volume; mirrors CorrelationFunction of models/correlation_package/correlation.py:6-44
    (there an instance-style Function unusable on modern torch) with the arithmetic of
    models/correlation_native.py:13-23."""

    @staticmethod
    def forward(ctx, x1, x2, max_displacement, negative_slope=1.0):
        _need_gpu(x1, x2)
        if x1.shape != x2.shape or x1.dim() != 4:
            raise ValueError('correlation expects two [B,C,H,W] tensors of equal shape')
        B, C, H, W = x1.shape
        d = int(max_displacement)
        slope = float(negative_slope)
        # The fast kernels need 16-byte aligned rows (W % 4 == 0).  Coarse pyramid levels such as 6x10 or
        # 8x14 are not: zero-pad the width (identical results on the real columns -- the volume's own
        # padding is zero as well), run the aligned kernel, crop.  The C ABI itself accepts any W.
        wp = (-W) % 4 if (d == 4 and C % 4 == 0) else 0
        if wp:
            x1 = torch.nn.functional.pad(x1, (0, wp))
            x2 = torch.nn.functional.pad(x2, (0, wp))
        x1, x2 = x1.contiguous(), x2.contiguous()
        Wk = W + wp
        out = torch.empty(B, (2 * d + 1) ** 2, H, Wk, device=x1.device, dtype=torch.float32)
        # fused LeakyReLU: the backward selects the derivative from 12 bytes of sign words per pixel (fast
        # path) instead of keeping and re-reading the 324-byte volume; shapes without the fast path keep `out`
        planes = _lib.load().arflow_corr_sign_planes(C, Wk, d) if slope != 1.0 else 0
        sign = torch.empty(B, planes, H, Wk, device=x1.device, dtype=torch.int32) if planes else None
        with torch.cuda.device_of(x1):
            _call('arflow_corr_fwd', _p(x1), _p(x2), _p(out), _p(sign), B, C, H, Wk, d, slope, _stream(),
                  key=(B, C, H, Wk, d, planes))
        if slope != 1.0:
            ctx.save_for_backward(x1, x2, sign if planes else out)
        else:
            ctx.save_for_backward(x1, x2)
        ctx.d, ctx.slope, ctx.w, ctx.wp, ctx.planes = d, slope, W, wp, planes
        return out[..., :W].contiguous() if wp else out

    @staticmethod
    def backward(ctx, gout):
        x1, x2 = ctx.saved_tensors[:2]
        act = ctx.saved_tensors[2] if ctx.slope != 1.0 else None
        fout, sign = (None, act) if ctx.planes else (act, None)
        B, C, H, Wk = x1.shape
        if ctx.wp:
            gout = torch.nn.functional.pad(gout, (0, ctx.wp))
        gout = gout.contiguous()
        g1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        with torch.cuda.device_of(x1):
            _call('arflow_corr_bwd', _p(gout), _p(fout), _p(sign), _p(x1), _p(x2), _p(g1), _p(g2), B, C, H, Wk, ctx.d,
                  ctx.slope, _stream(), key=(B, C, H, Wk, ctx.d, 0 if act is None else (ctx.planes or (2 * ctx.d + 1) ** 2)))
        if ctx.wp:
            g1 = None if g1 is None else g1[..., :ctx.w].contiguous()
            g2 = None if g2 is None else g2[..., :ctx.w].contiguous()
        return g1, g2, None, None


def _storage(storage):
    if storage in (None, 'fp32', torch.float32):
        return None
    if storage in ('bf16', torch.bfloat16):
        return 'bf16'
    raise ValueError('storage must be None / "fp32" or "bf16" / torch.bfloat16, got %r' % (storage,))


class CorrelationBF16Function(torch.autograd.Function):
    """Cost volume with the features STORED as bf16 (opt-in; the reference's native path
    dispatches half as well, correlation_cuda_kernel.cu:352,369): the fp32 inputs are rounded to bf16 once, the kernels
    read (and autograd keeps) 2 bytes per feature, products are accumulated in fp32, volume and gradients are fp32."""

    @staticmethod
    def forward(ctx, x1, x2, max_displacement, negative_slope):
        for t in (x1, x2):
            if not t.is_cuda:
                raise _lib.ArflowHipError('arflow_amd ops run on the GPU only')
        if x1.shape != x2.shape or x1.dim() != 4:
            raise ValueError('correlation expects two [B,C,H,W] tensors of equal shape')
        B, C, H, W = x1.shape
        d, slope = int(max_displacement), float(negative_slope)
        b1, b2 = x1.to(torch.bfloat16).contiguous(), x2.to(torch.bfloat16).contiguous()
        out = torch.empty(B, (2 * d + 1) ** 2, H, W, device=x1.device, dtype=torch.float32)
        with torch.cuda.device_of(x1):
            _call('arflow_corr_fwd_bf16', _p(b1), _p(b2), _p(out), B, C, H, W, d, slope, _stream(), key=(B, C, H, W, d, 0))
        ctx.save_for_backward(b1, b2, out if slope != 1.0 else None)
        ctx.d, ctx.slope = d, slope
        return out

    @staticmethod
    def backward(ctx, gout):
        b1, b2, fout = ctx.saved_tensors
        B, C, H, W = b1.shape
        gout = gout.contiguous()
        g1 = torch.empty(B, C, H, W, device=b1.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        g2 = torch.empty(B, C, H, W, device=b1.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        with torch.cuda.device_of(b1):
            _call('arflow_corr_bwd_bf16', _p(gout), _p(fout), _p(b1), _p(b2), _p(g1), _p(g2), B, C, H, W, ctx.d, ctx.slope,
                  _stream(), key=(B, C, H, W, ctx.d, 0 if fout is None else (2 * ctx.d + 1) ** 2))
        return g1, g2, None, None


class CorrelationGeneralFunction(torch.autograd.Function):
    """Correlation with the CUDA extension's full parameter set (correlation_cuda.cc:10-16: pad_size, kernel_size,
    max_displacement, stride1, stride2); the configuration every model uses goes through CorrelationFunction."""

    @staticmethod
    def forward(ctx, x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2):
        _need_gpu(x1, x2)
        if x1.shape != x2.shape or x1.dim() != 4:
            raise ValueError('correlation expects two [B,C,H,W] tensors of equal shape')
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, C, H, W = x1.shape
        cfg = (int(pad_size), int(kernel_size), int(max_displacement), int(stride1), int(stride2))
        import ctypes
        oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().arflow_corr_general_out_size(H, W, *cfg, ctypes.addressof(oc), ctypes.addressof(oh),
                                                            ctypes.addressof(ow)), 'arflow_corr_general_out_size')
        out = torch.empty(B, oc.value, oh.value, ow.value, device=x1.device, dtype=torch.float32)
        with torch.cuda.device_of(x1):
            _call('arflow_corr_general_fwd', _p(x1), _p(x2), _p(out), B, C, H, W, *cfg, _stream())
        ctx.save_for_backward(x1, x2)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, gout):
        x1, x2 = ctx.saved_tensors
        B, C, H, W = x1.shape
        gout = gout.contiguous()
        g1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        with torch.cuda.device_of(x1):
            _call('arflow_corr_general_bwd', _p(gout), _p(x1), _p(x2), _p(g1), _p(g2), B, C, H, W, *ctx.cfg, _stream())
        return g1, g2, None, None, None, None, None


def correlation_general(x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2):
    return CorrelationGeneralFunction.apply(x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2)


def correlation(x1, x2, max_displacement=4, negative_slope=1.0, storage=None):
    """Cost volume, optionally with the LeakyReLU every caller applies right after it fused into the
    kernel's store stage (and its derivative into the backward's load stage).  storage='bf16' (opt-in) keeps the
    features as bf16 in HBM: fp32 accumulation, fp32 volume and gradients."""
    if _storage(storage) == 'bf16':
        return CorrelationBF16Function.apply(x1, x2, max_displacement, negative_slope)
    return CorrelationFunction.apply(x1, x2, max_displacement, negative_slope)


class CorrConcatFunction(torch.autograd.Function):
    """cat([*before, leaky_relu(corr(x1, x2)), *after], dim=1) with the cost volume written by the kernel STRAIGHT
    into its channel slot of the concatenated buffer (arflow_corr_fwd_strided), and its gradient read in place from
    the gradient of that buffer (arflow_corr_bwd_strided): the decoders of the reference concatenate the volume with
    features and flow right after computing it (models/pwclite_uflow.py:218-222, models/pwclite.py:187-189,
    models/uflow_model.py:192-198) -- 81 of the 115-147 channels of that copy, forward, and a contiguous() of the
    81-channel gradient slice, backward, disappear.  The other members are copied into their slots."""

    @staticmethod
    def forward(ctx, x1, x2, max_displacement, negative_slope, n_before, *others):
        _need_gpu(x1, x2, *others)
        B, C, H, W = x1.shape
        d, slope = int(max_displacement), float(negative_slope)
        x1, x2 = x1.contiguous(), x2.contiguous()
        nvol = (2 * d + 1) ** 2
        chans = [int(t.shape[1]) for t in others]
        c0 = sum(chans[:n_before])
        ctot = nvol + sum(chans)
        buf = torch.empty(B, ctot, H, W, device=x1.device, dtype=torch.float32)
        planes = _lib.load().arflow_corr_sign_planes(C, W, d) if slope != 1.0 else 0
        sign = torch.empty(B, planes, H, W, device=x1.device, dtype=torch.int32) if planes else None
        vol = buf[:, c0:c0 + nvol]
        with torch.cuda.device_of(x1):
            _call('arflow_corr_fwd_strided', _p(x1), _p(x2), vol.data_ptr(), ctot * H * W, _p(sign), B, C, H, W, d, slope,
                  _stream(), key=(B, C, H, W, d, planes))
        off = 0
        for k, t in enumerate(others):
            if k == n_before:
                off += nvol
            buf[:, off:off + chans[k]].copy_(t)
            off += chans[k]
        # fast path with sign words keeps 12 B/px for the LeakyReLU derivative; otherwise the buffer itself (its
        # volume slot) is the saved forward output
        ctx.save_for_backward(x1, x2, sign if planes else (buf if slope != 1.0 else None))
        ctx.cfg = (d, slope, planes, n_before, chans, c0, nvol, ctot)
        return buf

    @staticmethod
    def backward(ctx, gbuf):
        x1, x2, act = ctx.saved_tensors
        d, slope, planes, n_before, chans, c0, nvol, ctot = ctx.cfg
        B, C, H, W = x1.shape
        gbuf = gbuf.contiguous()
        sign, fout = (act, None) if planes else (None, act)
        g1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        gvol = gbuf[:, c0:c0 + nvol]
        with torch.cuda.device_of(x1):
            _call('arflow_corr_bwd_strided', gvol.data_ptr(), ctot * H * W,
                  None if fout is None else fout[:, c0:c0 + nvol].data_ptr(), ctot * H * W, _p(sign), _p(x1), _p(x2),
                  _p(g1), _p(g2), B, C, H, W, d, slope, _stream(),
                  key=(B, C, H, W, d, 0 if act is None else (planes or nvol)))
        gothers, off = [], 0
        for k, c in enumerate(chans):
            if k == n_before:
                off += nvol
            gothers.append(gbuf[:, off:off + c] if ctx.needs_input_grad[5 + k] else None)
            off += c
        return (g1, g2, None, None, None) + tuple(gothers)


_CORR_CONCAT = __import__('os').environ.get('ARFLOW_CORR_CONCAT', '1') != '0'  # A/B switch for tools/ and bench.py


def corr_concat_supported(C, W, max_displacement=4):
    return bool(_lib.load().arflow_corr_strided_supported(int(C), int(W), int(max_displacement)))


def correlation_concat(x1, x2, before=(), after=(), max_displacement=4, negative_slope=1.0):
    """torch.cat([*before, leaky_relu(correlation(x1, x2)), *after], 1) without ever copying the cost volume (shapes
    the strided kernels do not take fall back to the plain op + torch.cat)."""
    B, C, H, W = x1.shape
    if x1.is_cuda and _CORR_CONCAT and corr_concat_supported(C, W, max_displacement):
        return CorrConcatFunction.apply(x1, x2, max_displacement, negative_slope, len(before), *before, *after)
    return torch.cat(list(before) + [correlation(x1, x2, max_displacement, negative_slope)] + list(after), 1)


# ------------------------------------------------------------------------------------------------
FEATNORM = {'joint': 0, 'avg': 1}
def _featnorm_acc(B, device):  # ARFLOW_FEATNORM_ACC_DOUBLES(B)
    return torch.empty(4 * (2048 + B), device=device, dtype=torch.float64)


class FeatureNormFunction(torch.autograd.Function):
    """(x1, x2) -> ((x1 - mu) / std, (x2 - mu) / std) with per-sample moments over both tensors:
    normalize_features of models/pwclite_uflow.py:30-38 ('joint') and of models/uflow_model.py:8-50 as
    PWCFlow calls it ('avg').  Two launches forward, two backward."""

    @staticmethod
    def forward(ctx, x1, x2, mode):
        _need_gpu(x1, x2)
        if x1.shape != x2.shape or x1.dim() < 2:
            raise ValueError('feature normalisation expects two tensors of equal shape [B, ...]')
        x1, x2 = x1.contiguous(), x2.contiguous()
        B = x1.shape[0]
        n = x1[0].numel()
        y1, y2 = torch.empty_like(x1), torch.empty_like(x2)
        acc = _featnorm_acc(B, x1.device)
        stats = torch.empty(B, 4, device=x1.device, dtype=torch.float32)
        with torch.cuda.device_of(x1):
            _call('arflow_featnorm_fwd', _p(x1), _p(x2), _p(y1), _p(y2), _p(acc), _p(stats), B, n, mode, _stream(),
                  key=(B, n))
        ctx.save_for_backward(x1, x2, stats)
        ctx.mode = mode
        return y1, y2

    @staticmethod
    def backward(ctx, g1, g2):
        x1, x2, stats = ctx.saved_tensors
        B = x1.shape[0]
        n = x1[0].numel()
        g1, g2 = g1.contiguous(), g2.contiguous()
        d1 = torch.empty_like(x1) if ctx.needs_input_grad[0] else None
        d2 = torch.empty_like(x2) if ctx.needs_input_grad[1] else None
        acc = _featnorm_acc(B, x1.device)
        with torch.cuda.device_of(x1):
            _call('arflow_featnorm_bwd', _p(g1), _p(g2), _p(x1), _p(x2), _p(stats), _p(acc), _p(d1), _p(d2), B, n,
                  ctx.mode, _stream(), key=(B, n))
        return d1, d2, None


def normalize_pair(x1, x2, mode='joint'):
    return FeatureNormFunction.apply(x1, x2, FEATNORM[mode])


# ------------------------------------------------------------------------------------------------
class LevelCfg:
    """Static description of one fused pyramid level (not a tensor: travels through autograd as a plain argument).

    layout: the channel order of the decoder's concatenated input as a list of 'vol' (cost volume + LeakyReLU), 'x1n'
    (normalised first map), 'flow' (the x2-upsampled flow, only with a coarse flow) and integers (index into the extra
    member tensors)."""

    def __init__(self, layout, norm_mode, slope=0.1, max_displacement=4, flow_is_coarse=True, up_align_corners=True,
                 pad='zeros', align_corners=True, coord_norm=NORM_ARFLOW):
        self.layout = list(layout)
        self.mode = FEATNORM[norm_mode]
        self.slope, self.d = float(slope), int(max_displacement)
        self.flow_is_coarse, self.up_align = bool(flow_is_coarse), bool(up_align_corners)
        self.pad, self.align, self.norm = PAD[pad], int(bool(align_corners)), int(coord_norm)


def level_supported(x1, flow, flow_is_coarse, max_displacement=4):
    """The fused level launches take the shapes of the tuned correlation kernels (W % 4 == 0, C % 4 == 0, d = 4)."""
    B, C, H, W = x1.shape
    if not (x1.is_cuda and x1.dtype == torch.float32 and _LEVEL_FUSED):
        return False
    if not _lib.load().arflow_corr_strided_supported(int(C), int(W), int(max_displacement)):
        return False
    if flow is not None and flow_is_coarse and (H % 2 or W % 2 or tuple(flow.shape[2:]) != (H // 2, W // 2)):
        return False
    return True


_LEVEL_FUSED = __import__('os').environ.get('ARFLOW_LEVEL_FUSED', '1') != '0'  # A/B switch for tools/ and tests


class LevelFunction(torch.autograd.Function):
    """One pyramid level in front of its flow estimator (models/pwclite_uflow.py:203-222,
    models/uflow_model.py:160-198):

        flow_up = interpolate(flow * 2, x2)                 (flow_is_coarse)
        x2w     = warp(x2, flow_up)                         (no flow: x2w = x2)
        x1n, x2n = normalize_features([x1, x2w])
        buf     = cat([..., leaky_relu(corr(x1n, x2n)), x1n, flow_up, ...], 1)

    forward = arflow_level_warp_fwd (or arflow_level_moments) + arflow_level_corr_fwd: the normalised second map, the
    moment / apply passes of the normalisation and the interpolate / mul / cat launches do not exist.  Returns
    (buf, flow_up) with a coarse flow, else buf."""

    @staticmethod
    def forward(ctx, x1, x2, flow, cfg, x1_rows, x2_rows, *members):
        """x1_rows / x2_rows (optional [B,rows,2] float64): partial moments of x1 / (at a level without flow) x2 from
        bias_leaky_relu_moments -- the launch that produced the maps; the warp launch then skips reading x1 and the
        level without a warp needs no moment pass."""
        _need_gpu(x1, x2, flow, *members)
        if x1.shape != x2.shape or x1.dim() != 4:
            raise ValueError('level expects two [B,C,H,W] feature maps of equal shape')
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, C, H, W = x1.shape
        lib = _lib.load()
        nvol = (2 * cfg.d + 1) ** 2
        has_flow = flow is not None
        chans, offs, off = [], {}, 0
        for item in cfg.layout:
            c = {'vol': nvol, 'x1n': C, 'flow': 2}[item] if isinstance(item, str) else int(members[item].shape[1])
            offs[item] = off
            chans.append(c)
            off += c
        ctot = off
        if 'flow' in offs and not (has_flow and cfg.flow_is_coarse):
            raise ValueError("layout item 'flow' needs a coarse flow")
        buf = torch.empty(B, ctot, H, W, device=x1.device, dtype=torch.float32)
        bs = ctot * H * W
        rows = lib.arflow_level_acc_rows(B, C, H, W, int(has_flow))
        acc = torch.empty(4 * B * rows, device=x1.device, dtype=torch.float64)
        stats = torch.empty(B, 4, device=x1.device, dtype=torch.float32)
        sign = torch.empty(B, 3, H, W, device=x1.device, dtype=torch.int32) if cfg.slope != 1.0 else None
        flow_full = x2w = x1n = None
        fbs, fslot, fup = 0, None, None
        if has_flow:
            flow, fbs = _flow_view(flow)
            x2w = torch.empty_like(x2)
            if cfg.flow_is_coarse:
                flow_full = fup = torch.empty(B, 2, H, W, device=x1.device, dtype=torch.float32)
                fslot = buf[:, offs['flow']:].data_ptr() if 'flow' in offs else None
            else:
                flow_full = flow
        if 'x1n' in offs:
            x1n_ptr, x1n_bs = buf[:, offs['x1n']:].data_ptr(), bs
        else:
            x1n = torch.empty_like(x1)
            x1n_ptr, x1n_bs = x1n.data_ptr(), C * H * W
        if x1_rows is not None:
            if x1_rows.dtype != torch.float64 or x1_rows.shape[0] != B or x1_rows.shape[2] != 2:
                raise ValueError('x1_rows must be [B, rows, 2] float64')
            x1_rows = x1_rows.contiguous()
        if x2_rows is not None:
            x2_rows = x2_rows.contiguous() if (not has_flow and x1_rows is not None) else None
        with torch.cuda.device_of(x1):
            _call('arflow_level_fwd_m', _p(x1), _p(x2), _p(flow), fbs, int(cfg.flow_is_coarse and has_flow), int(cfg.up_align),
                  _p(fup), fslot, bs, _p(x2w), cfg.mode, buf[:, offs['vol']:].data_ptr(), bs, x1n_ptr, x1n_bs, _p(sign),
                  _p(stats), _p(acc), _p(x1_rows), 0 if x1_rows is None else int(x1_rows.shape[1]), _p(x2_rows),
                  0 if x2_rows is None else int(x2_rows.shape[1]), B, C, H, W, cfg.d, cfg.slope, cfg.pad, cfg.align, cfg.norm,
                  _stream(), key=(B, C, H, W, cfg.d, 3 if sign is not None else 0,
                                  int(has_flow) + int(has_flow and cfg.flow_is_coarse), int(x1_rows is not None)))
        for item in cfg.layout:
            if not isinstance(item, str):
                buf[:, offs[item]:offs[item] + int(members[item].shape[1])].copy_(members[item])
        ctx.save_for_backward(x1, x2, x2w, flow_full, stats, sign, buf if x1n is None else x1n)
        ctx.cfg, ctx.offs, ctx.ctot, ctx.n_members, ctx.has_flow = cfg, offs, ctot, len(members), has_flow  # noqa
        ctx.member_chans = [int(m.shape[1]) for m in members]
        if has_flow and cfg.flow_is_coarse:
            return buf, flow_full
        return buf

    @staticmethod
    def backward(ctx, gbuf, gflow_ext=None):
        x1, x2, x2w, flow_full, stats, sign, x1n_holder = ctx.saved_tensors
        cfg, offs, ctot = ctx.cfg, ctx.offs, ctx.ctot
        B, C, H, W = x1.shape
        nvol = (2 * cfg.d + 1) ** 2
        gbuf = gbuf.contiguous()
        bs = ctot * H * W
        if 'x1n' in offs:
            x1n_ptr, x1n_bs = x1n_holder[:, offs['x1n']:].data_ptr(), bs
        else:
            x1n_ptr, x1n_bs = x1n_holder.data_ptr(), C * H * W
        d1, gx2 = torch.empty_like(x1), torch.empty_like(x1)
        ws = torch.empty(_lib.load().arflow_level_bwd_ws_bytes(B, C, H, W), device=x1.device, dtype=torch.uint8)
        gflow_in = fl = gslot = None
        fbs = 0
        coarse = ctx.has_flow and cfg.flow_is_coarse
        if ctx.has_flow:
            fl, fbs = _flow_view(flow_full)
            gflow_in = torch.empty(B, 2, H // 2, W // 2, device=x1.device, dtype=torch.float32) if coarse else \
                torch.empty(B, 2, H, W, device=x1.device, dtype=torch.float32)
            if 'flow' in offs:
                gslot = gbuf[:, offs['flow']:].data_ptr()
            if gflow_ext is not None:
                gflow_ext = gflow_ext.contiguous()
        gdir = gbuf[:, offs['x1n']:].data_ptr() if 'x1n' in offs else None
        with torch.cuda.device_of(x1):
            _call('arflow_level_bwd', gbuf[:, offs['vol']:].data_ptr(), bs, _p(sign), x1n_ptr, x1n_bs, gdir, bs, _p(x1),
                  _p(x2), _p(x2w), _p(fl), fbs, gslot, bs, _p(gflow_ext), _p(stats), cfg.mode, _p(d1), _p(gx2), _p(gflow_in),
                  int(coarse), int(cfg.up_align), _p(ws), B, C, H, W, cfg.d, cfg.slope, cfg.pad, cfg.align, cfg.norm,
                  _stream(), key=(B, C, H, W, cfg.d, 3 if sign is not None else 0, int(ctx.has_flow) + int(coarse)))
        gm, k = [], 0
        for item in cfg.layout:
            if not isinstance(item, str):
                gm.append((item, gbuf[:, offs[item]:offs[item] + ctx.member_chans[item]]))
        gmembers = [None] * ctx.n_members
        for idx, g in gm:
            gmembers[idx] = g if ctx.needs_input_grad[6 + idx] else None
        return (d1, gx2, gflow_in, None, None, None) + tuple(gmembers)


def level(x1, x2, flow, cfg, *members, x1_rows=None, x2_rows=None):
    return LevelFunction.apply(x1, x2, flow, cfg, x1_rows, x2_rows, *members)


# ------------------------------------------------------------------------------------------------
class BiasLeakyReLUFunction(torch.autograd.Function):
    """y = leaky_relu(x + bias[None, :, None, None], slope), IN PLACE on x (the bias-free output of a
    convolution, which autograd does not need again); backward = LeakyReLU derivative and bias gradient in
    one pass.  Replaces the bias add + nn.LeakyReLU(0.1, inplace=True) behind every conv of the reference
    models (models/pwclite.py:10-23)."""

    @staticmethod
    def forward(ctx, x, bias, slope):
        _need_gpu(x)
        if not x.is_contiguous():
            raise ValueError('bias_leaky_relu expects a contiguous [B,C,...] tensor')
        B, C = x.shape[0], x.shape[1]
        hw = x[0, 0].numel()
        if bias is not None:
            _need_gpu(bias)
            bias = bias.contiguous()
        with torch.cuda.device_of(x):
            _call('arflow_bias_act_fwd', _p(x), _p(bias), _p(x), B, C, hw, float(slope), _stream(), key=(B, C, hw))
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        ctx.slope, ctx.has_bias = float(slope), bias is not None
        return x

    @staticmethod
    def backward(ctx, gout):
        y, = ctx.saved_tensors
        B, C = y.shape[0], y.shape[1]
        hw = y[0, 0].numel()
        gout = gout.contiguous()
        gin = torch.empty_like(gout)
        gb = torch.empty(C, device=y.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        with torch.cuda.device_of(y):
            _call('arflow_bias_act_bwd', _p(gout), _p(y), _p(gin), _p(gb), B, C, hw, ctx.slope, _stream(), key=(B, C, hw))
        return gin, gb, None


def bias_leaky_relu(x, bias, slope=0.1):
    return BiasLeakyReLUFunction.apply(x, bias, slope)


class BiasLeakyReLUMomentsFunction(torch.autograd.Function):
    """bias_leaky_relu that also returns the partial moments (sum y, sum y^2) of its output as rows of 2 doubles
    ([B, rows, 2], arflow_bias_act_fwd_mom): normalize_features' moments taken where the feature map is produced."""

    @staticmethod
    def forward(ctx, x, bias, slope):
        _need_gpu(x)
        if not x.is_contiguous():
            raise ValueError('bias_leaky_relu expects a contiguous [B,C,...] tensor')
        B, C = x.shape[0], x.shape[1]
        hw = x[0, 0].numel()
        if bias is not None:
            _need_gpu(bias)
            bias = bias.contiguous()
        rows = _lib.load().arflow_bias_act_mom_rows(C, hw)
        mom = torch.empty(B, rows, 2, device=x.device, dtype=torch.float64)
        with torch.cuda.device_of(x):
            _call('arflow_bias_act_fwd_mom', _p(x), _p(bias), _p(x), _p(mom), B, C, hw, float(slope), _stream(), key=(B, C, hw))
        ctx.mark_dirty(x)
        ctx.mark_non_differentiable(mom)
        ctx.save_for_backward(x)
        ctx.slope, ctx.has_bias = float(slope), bias is not None
        return x, mom

    @staticmethod
    def backward(ctx, gout, gmom_unused):
        y, = ctx.saved_tensors
        B, C = y.shape[0], y.shape[1]
        hw = y[0, 0].numel()
        gout = gout.contiguous()
        gin = torch.empty_like(gout)
        gb = torch.empty(C, device=y.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[1]) else None
        with torch.cuda.device_of(y):
            _call('arflow_bias_act_bwd', _p(gout), _p(y), _p(gin), _p(gb), B, C, hw, ctx.slope, _stream(), key=(B, C, hw))
        return gin, gb, None


def bias_leaky_relu_moments(x, bias, slope=0.1):
    return BiasLeakyReLUMomentsFunction.apply(x, bias, slope)


# ------------------------------------------------------------------------------------------------
class WarpFunction(torch.autograd.Function):
    """out = bilinear(src, grid + flow); with ``want_valid`` also the in-image mask of the sampling
    positions (mask_invalid(flow_to_warp(flow)), utils/uflow_utils.py:35-50) from the same launch."""

    @staticmethod
    def forward(ctx, src, flow, pad, align_corners, norm, want_valid=False):
        _need_gpu(src, flow)
        src = src.contiguous()
        flow, fbs = _flow_view(flow)
        B, C, Hs, Ws = src.shape
        _, _, H, W = flow.shape
        if flow.shape[0] != B:
            raise ValueError('batch mismatch between source and flow')
        out = torch.empty(B, C, H, W, device=src.device, dtype=torch.float32)
        valid = torch.empty(B, 1, H, W, device=src.device, dtype=torch.float32) if want_valid else None
        with torch.cuda.device_of(src):
            _call('arflow_warp_fwd', _p(src), _p(flow), _p(out), _p(valid), B, C, Hs, Ws, H, W, fbs, pad,
                  int(bool(align_corners)), norm, _stream(), key=(B, C, H, W))
        ctx.save_for_backward(src, flow)
        ctx.cfg = (pad, int(bool(align_corners)), norm, fbs)
        if want_valid:
            ctx.mark_non_differentiable(valid)
            return out, valid
        return out

    @staticmethod
    def backward(ctx, gout, *unused):
        src, flow = ctx.saved_tensors
        pad, ac, norm, fbs = ctx.cfg
        B, C, Hs, Ws = src.shape
        _, _, H, W = flow.shape
        gout = gout.contiguous()
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        gflow = torch.empty(B, 2, H, W, device=src.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        with torch.cuda.device_of(src):
            _call('arflow_warp_bwd', _p(gout), _p(src), _p(flow), _p(gsrc), _p(gflow), B, C, Hs, Ws, H, W, fbs,
                  pad, ac, norm, _stream(), key=(B, C, H, W, gsrc is not None))
        return gsrc, gflow, None, None, None, None


class WarpUp2Function(torch.autograd.Function):
    """The front of a PWC pyramid level as the ARFlow model writes it (models/pwclite.py:178-180):

        flow_up = F.interpolate(flow * 2, scale_factor=2, mode='bilinear', align_corners=up_align)
        out     = flow_warp(src, flow_up)

    as ONE launch (arflow_level_warp_fwd: the upsample is evaluated in front of the warp's coordinate computation; the
    upsampled flow is still written once -- the estimator and the residual sum read it).  Backward: arflow_warp_bwd (both
    gradients w.r.t. the full-resolution flow) + arflow_up2_bwd (the upsample's adjoint as a gather).  Returns
    (out, flow_up)."""

    @staticmethod
    def forward(ctx, src, flow, pad, align_corners, norm, up_align):
        _need_gpu(src, flow)
        src = src.contiguous()
        flow, fbs = _flow_view(flow)
        B, C, H, W = src.shape
        if tuple(flow.shape) != (B, 2, H // 2, W // 2) or H % 2 or W % 2:
            raise ValueError('warp_up2 needs a [B,2,H/2,W/2] flow for a [B,C,H,W] source with even H, W')
        lib = _lib.load()
        out = torch.empty_like(src)
        flow_up = torch.empty(B, 2, H, W, device=src.device, dtype=torch.float32)
        acc = torch.empty(4 * B * lib.arflow_level_acc_rows(B, C, H, W, 1), device=src.device, dtype=torch.float64)  # (moments: unused)
        with torch.cuda.device_of(src):
            _call('arflow_level_warp_fwd', None, _p(src), _p(flow), fbs, 1, int(bool(up_align)), _p(flow_up), None, 0, _p(out),
                  _p(acc), B, C, H, W, pad, int(bool(align_corners)), norm, _stream(), key=(B, C, H, W, 1))
        ctx.save_for_backward(src, flow_up)
        ctx.cfg = (pad, int(bool(align_corners)), norm, int(bool(up_align)))
        return out, flow_up

    @staticmethod
    def backward(ctx, gout, gflow_up):
        src, flow_up = ctx.saved_tensors
        pad, ac, norm, up_align = ctx.cfg
        B, C, H, W = src.shape
        gout = gout.contiguous()
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        gcoarse = None
        with torch.cuda.device_of(src):
            if ctx.needs_input_grad[1]:
                gfull = torch.empty(B, 2, H, W, device=src.device, dtype=torch.float32)
                _call('arflow_warp_bwd', _p(gout), _p(src), _p(flow_up), _p(gsrc), _p(gfull), B, C, H, W, H, W, 2 * H * W, pad, ac,
                      norm, _stream(), key=(B, C, H, W, gsrc is not None))
                if gflow_up is not None:
                    gfull = gfull + gflow_up  # what the estimator and the residual sum sent to the upsampled flow
                gcoarse = torch.empty(B, 2, H // 2, W // 2, device=src.device, dtype=torch.float32)
                _call('arflow_up2_bwd', _p(gfull), _p(gcoarse), B, H, W, up_align, _stream(), key=(B, H, W))
            elif gsrc is not None:
                _call('arflow_warp_bwd', _p(gout), _p(src), _p(flow_up), _p(gsrc), None, B, C, H, W, H, W, 2 * H * W, pad, ac, norm,
                      _stream(), key=(B, C, H, W, True))
        return gsrc, gcoarse, None, None, None, None


def warp_up2(src, flow_coarse, pad='zeros', align_corners=True, norm=None, up_align=True):
    """(flow_warp(src, up), up) with up = interpolate(flow_coarse * 2, x2, bilinear, align_corners=up_align) in one launch."""
    return WarpUp2Function.apply(src, flow_coarse, PAD[pad], align_corners, NORM_ARFLOW if norm is None else norm, up_align)


def warp_up2_supported(src, flow_coarse):
    return (src.is_cuda and src.dtype == torch.float32 and flow_coarse is not None and flow_coarse.is_cuda and src.shape[2] % 2 == 0
            and src.shape[3] % 2 == 0 and tuple(flow_coarse.shape[2:]) == (src.shape[2] // 2, src.shape[3] // 2)
            and flow_coarse.shape[1] == 2)


class WarpBF16Function(torch.autograd.Function):
    """Bilinear warp with the SOURCE stored as bf16 (opt-in): sampling arithmetic, output and
    both gradients fp32; the source is rounded to bf16 once and kept as such for the backward."""

    @staticmethod
    def forward(ctx, src, flow, pad, align_corners, norm):
        if not (src.is_cuda and flow.is_cuda):
            raise _lib.ArflowHipError('arflow_amd ops run on the GPU only')
        sb = src.to(torch.bfloat16).contiguous()
        flow, fbs = _flow_view(flow.float())
        B, C, Hs, Ws = sb.shape
        _, _, H, W = flow.shape
        out = torch.empty(B, C, H, W, device=src.device, dtype=torch.float32)
        with torch.cuda.device_of(src):
            _call('arflow_warp_fwd_bf16', _p(sb), _p(flow), _p(out), None, B, C, Hs, Ws, H, W, fbs, pad,
                  int(bool(align_corners)), norm, _stream(), key=(B, C, H, W))
        ctx.save_for_backward(sb, flow)
        ctx.cfg = (pad, int(bool(align_corners)), norm, fbs)
        return out

    @staticmethod
    def backward(ctx, gout):
        sb, flow = ctx.saved_tensors
        pad, ac, norm, fbs = ctx.cfg
        B, C, Hs, Ws = sb.shape
        _, _, H, W = flow.shape
        gout = gout.contiguous()
        gsrc = torch.empty(B, C, Hs, Ws, device=sb.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        gflow = torch.empty(B, 2, H, W, device=sb.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        with torch.cuda.device_of(sb):
            _call('arflow_warp_bwd_bf16', _p(gout), _p(sb), _p(flow), _p(gsrc), _p(gflow), B, C, Hs, Ws, H, W, fbs, pad, ac,
                  norm, _stream(), key=(B, C, H, W, gsrc is not None))
        return gsrc, gflow, None, None, None


class WarpNearestFunction(torch.autograd.Function):
    """flow_warp(mode='nearest') (utils/warp_utils.py:83-90): gradient w.r.t. the source only, like grid_sample."""

    @staticmethod
    def forward(ctx, src, flow, pad, align_corners, norm):
        _need_gpu(src, flow)
        src = src.contiguous()
        flow, fbs = _flow_view(flow.detach())
        B, C, Hs, Ws = src.shape
        _, _, H, W = flow.shape
        out = torch.empty(B, C, H, W, device=src.device, dtype=torch.float32)
        with torch.cuda.device_of(src):
            _call('arflow_warp_nearest_fwd', _p(src), _p(flow), _p(out), B, C, Hs, Ws, H, W, fbs, pad,
                  int(bool(align_corners)), norm, _stream())
        ctx.save_for_backward(flow)
        ctx.cfg = (pad, int(bool(align_corners)), norm, fbs, (B, C, Hs, Ws))
        return out

    @staticmethod
    def backward(ctx, gout):
        flow, = ctx.saved_tensors
        pad, ac, norm, fbs, (B, C, Hs, Ws) = ctx.cfg
        _, _, H, W = flow.shape
        gsrc = None
        if ctx.needs_input_grad[0]:
            gout = gout.contiguous()
            gsrc = torch.empty(B, C, Hs, Ws, device=gout.device, dtype=torch.float32)
            with torch.cuda.device_of(gout):
                _call('arflow_warp_nearest_bwd', _p(gout), _p(flow), _p(gsrc), B, C, Hs, Ws, H, W, fbs, pad, ac, norm,
                      _stream())
        # grid_sample's nearest mode returns a ZERO gradient for the grid (not None)
        gflow = torch.zeros(B, 2, H, W, device=gout.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        return gsrc, gflow, None, None, None


def warp_nearest(src, flow, pad='zeros', align_corners=True, norm=NORM_ARFLOW):
    return WarpNearestFunction.apply(src, flow, PAD[pad], align_corners, norm)


class WarpBicubicFunction(torch.autograd.Function):
    """flow_warp(mode='bicubic') (utils/warp_utils.py:83-90 -> grid_sample bicubic): both gradients."""

    @staticmethod
    def forward(ctx, src, flow, pad, align_corners, norm):
        _need_gpu(src, flow)
        src = src.contiguous()
        flow, fbs = _flow_view(flow)
        B, C, Hs, Ws = src.shape
        _, _, H, W = flow.shape
        out = torch.empty(B, C, H, W, device=src.device, dtype=torch.float32)
        with torch.cuda.device_of(src):
            _call('arflow_warp_bicubic_fwd', _p(src), _p(flow), _p(out), B, C, Hs, Ws, H, W, fbs, pad, int(bool(align_corners)),
                  norm, _stream())
        ctx.save_for_backward(src, flow)
        ctx.cfg = (pad, int(bool(align_corners)), norm, fbs)
        return out

    @staticmethod
    def backward(ctx, gout):
        src, flow = ctx.saved_tensors
        pad, ac, norm, fbs = ctx.cfg
        B, C, Hs, Ws = src.shape
        _, _, H, W = flow.shape
        gout = gout.contiguous()
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        gflow = torch.empty(B, 2, H, W, device=src.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        with torch.cuda.device_of(src):
            _call('arflow_warp_bicubic_bwd', _p(gout), _p(src), _p(flow), _p(gsrc), _p(gflow), B, C, Hs, Ws, H, W, fbs, pad, ac,
                  norm, _stream())
        return gsrc, gflow, None, None, None


def warp_bicubic(src, flow, pad='zeros', align_corners=True, norm=NORM_ARFLOW):
    return WarpBicubicFunction.apply(src, flow, PAD[pad], align_corners, norm)


def warp(src, flow, pad='zeros', align_corners=True, norm=NORM_ARFLOW, storage=None):
    if _storage(storage) == 'bf16':
        return WarpBF16Function.apply(src, flow, PAD[pad], align_corners, norm)
    return WarpFunction.apply(src, flow, PAD[pad], align_corners, norm)


def warp_with_valid(src, flow, pad='zeros', align_corners=True, norm=NORM_ARFLOW):
    """(warped, valid mask) in one launch."""
    return WarpFunction.apply(src, flow, PAD[pad], align_corners, norm, True)


def _flow_map(name, flow, *extra):
    _need_gpu(flow)
    flow, fbs = _flow_view(flow.detach())
    B, _, H, W = flow.shape
    out = torch.empty(B, 1, H, W, device=flow.device, dtype=torch.float32)
    with torch.cuda.device_of(flow):
        _call(name, _p(flow), _p(out), B, H, W, fbs, *extra, _stream(), key=(B, H, W))
    return out


def splat_map(flow, variant):
    """variant 0: compute_range_map; 1: get_corresponding_map(grid + flow).  No gradient (the
    reference detaches / thresholds it: losses/uflow_loss.py:43, utils/warp_utils.py:112)."""
    return _flow_map('arflow_splat_map', flow, int(variant))


def coord_mask(flow, mode):
    """mode 0: mask_invalid(flow_to_warp(flow)); 1: border_mask(flow)."""
    return _flow_map('arflow_coord_mask', flow, int(mode))


def occ_bidir(flow12, flow21, scale=0.01, bias=0.5):
    _need_gpu(flow12, flow21)
    f12, s12 = _flow_view(flow12.detach())
    f21, s21 = _flow_view(flow21.detach())
    B, _, H, W = f12.shape
    out = torch.empty(B, 1, H, W, device=f12.device, dtype=torch.float32)
    with torch.cuda.device_of(f12):
        _call('arflow_occ_bidir', _p(f12), _p(f21), _p(out), B, H, W, s12, s21, float(scale), float(bias), _stream(),
              key=(B, H, W))
    return out


# ------------------------------------------------------------------------------------------------
class CensusLossFunction(torch.autograd.Function):
    """census_loss(image_a, image_b, mask, patch) of utils/uflow_utils.py:282-293 as one fused
    forward launch and one backward launch.  The mask is treated as a constant (the reference
    passes a detached mask, losses/uflow_loss.py:43,48)."""

    @staticmethod
    def forward(ctx, im_a, im_b, mask, patch_size):
        _need_gpu(im_a, im_b, mask)
        im_a, im_b, mask = im_a.contiguous(), im_b.contiguous(), mask.contiguous()
        B, C, H, W = im_a.shape
        if C != 3 or im_b.shape != im_a.shape or mask.shape != (B, 1, H, W):
            raise ValueError('census_loss expects [B,3,H,W] images and a [B,1,H,W] mask')
        r = int(patch_size) // 2
        buf = _new_sums(im_a.device, B, H, W)
        dham = torch.empty(B, 1, H, W, device=im_a.device, dtype=torch.float32)
        with torch.cuda.device_of(im_a):
            _call('arflow_census_fwd', _p(im_a), _p(im_b), _p(mask), None, _p(dham), _p(buf), B, H, W, r, _stream(),
                  key=(B, H, W))
        sums = _fold_sums(buf, 2)
        # sharded batch + global normalisation: divide by the mask sum of ALL ranks (ddp.global_denominator);
        # identity when that is off (the default)
        den = _ddp.global_denominator(sums[1]) + 1e-6 / _ddp.world_size()
        inv = 1.0 / den
        ctx.save_for_backward(im_a, im_b, dham, inv)
        ctx.r = r
        return sums[0] * inv

    @staticmethod
    def backward(ctx, gloss):
        im_a, im_b, dham, inv = ctx.saved_tensors
        B, _, H, W = im_a.shape
        scale = (gloss * inv).reshape(1).contiguous()
        ga = gb = None
        with torch.cuda.device_of(im_a):
            if ctx.needs_input_grad[1]:
                gb = torch.empty_like(im_b)
                _call('arflow_census_bwd', _p(im_a), _p(im_b), _p(dham), _p(scale), _p(gb), B, H, W, ctx.r, _stream(),
                      key=(B, H, W))
            if ctx.needs_input_grad[0]:
                ga = torch.empty_like(im_a)  # the distance is symmetric in (a, b)
                _call('arflow_census_bwd', _p(im_b), _p(im_a), _p(dham), _p(scale), _p(ga), B, H, W, ctx.r, _stream(),
                      key=(B, H, W))
        return ga, gb, None, None


class CensusWarpLossFunction(torch.autograd.Function):
    """One photometric direction of UFlowLoss (losses/uflow_loss.py:30-54) as ONE launch each way:
    census_loss(im_a, resample(im_b, flow_to_warp(flow)), upsample(clamp(occ_small,0,1), x4) * mask_invalid(...)),
    on the grey planes (x255) of the two images (``gray255``).  Returns (loss, mask).  Gradient w.r.t. the flow only
    (the reference detaches the sampled image and the mask, losses/uflow_loss.py:31,34,43,48)."""

    @staticmethod
    def forward(ctx, gray_a, gray_b, flow, occ_small, patch_size):
        _need_gpu(gray_a, gray_b, flow, occ_small)
        gray_a, gray_b = gray_a.detach().contiguous(), gray_b.detach().contiguous()
        flow, fbs = _flow_view(flow)
        B, C, H, W = gray_a.shape
        if C != 1 or gray_b.shape != gray_a.shape or flow.shape != (B, 2, H, W):
            raise ValueError('census_warp_loss expects [B,1,H,W] grey planes and a [B,2,H,W] flow')
        if occ_small is not None:
            occ_small = occ_small.detach().contiguous()
            if occ_small.shape != (B, 1, H // 4, W // 4):
                raise ValueError('occ_small must be the [B,1,H/4,W/4] range map')
        r = int(patch_size) // 2
        buf = _new_sums(gray_a.device, B, H, W)
        dham = torch.empty(B, 1, H, W, device=gray_a.device, dtype=torch.float32)
        mask = torch.empty(B, 1, H, W, device=gray_a.device, dtype=torch.float32)
        with torch.cuda.device_of(gray_a):
            _call('arflow_census_warp_fwd', _p(gray_a), _p(gray_b), _p(flow), fbs, _p(occ_small), _p(mask), _p(dham),
                  _p(buf), B, H, W, r, _stream(), key=(B, H, W))
        sums = _fold_sums(buf, 2)
        den = _ddp.global_denominator(sums[1]) + 1e-6 / _ddp.world_size()
        inv = 1.0 / den
        ctx.save_for_backward(gray_a, gray_b, flow, dham, inv)
        ctx.r, ctx.fbs = r, fbs
        ctx.mark_non_differentiable(mask)
        return sums[0] * inv, mask

    @staticmethod
    def backward(ctx, gloss, gmask_unused):
        gray_a, gray_b, flow, dham, inv = ctx.saved_tensors
        B, _, H, W = gray_a.shape
        scale = (gloss * inv).reshape(1).contiguous()
        gflow = torch.empty(B, 2, H, W, device=gray_a.device, dtype=torch.float32)
        with torch.cuda.device_of(gray_a):
            _call('arflow_census_warp_bwd', _p(gray_a), _p(gray_b), _p(flow), ctx.fbs, _p(dham), _p(scale), _p(gflow),
                  B, H, W, ctx.r, _stream(), key=(B, H, W))
        return None, None, gflow, None, None


class CensusWarpPairLossFunction(torch.autograd.Function):
    """BOTH photometric directions of UFlowLoss (losses/uflow_loss.py:30-54) as ONE launch each way.  The batch holds
    2B samples s = 2 b + direction: ``gray2`` the grey planes (x255) of the [B,6,H,W] pair viewed as [2B,3,H,W], ``flow2``
    the [B,4,H,W] (fw, bw) flows viewed as [2B,2,H,W], ``occ_small2`` the range maps of the 2B level-2 flows (sample s is
    masked by plane s ^ 1).  Returns (loss fw, loss bw, mask [2B,1,H,W]); gradient w.r.t. the flows only."""

    @staticmethod
    def forward(ctx, gray2, flow2, occ_small2, patch_size):
        _need_gpu(gray2, flow2, occ_small2)
        gray2 = gray2.detach().contiguous()
        flow2, fbs = _flow_view(flow2)
        B2, C, H, W = gray2.shape
        if C != 1 or B2 % 2 or flow2.shape != (B2, 2, H, W):
            raise ValueError('census_warp_pair_loss expects [2B,1,H,W] grey planes and [2B,2,H,W] flows')
        occ_small2 = occ_small2.detach().contiguous()
        if occ_small2.shape != (B2, 1, H // 4, W // 4):
            raise ValueError('occ_small2 must be the [2B,1,H/4,W/4] range maps')
        r = int(patch_size) // 2
        buf = _new_sums(gray2.device, B2, H, W)
        dham = torch.empty(B2, 1, H, W, device=gray2.device, dtype=torch.float32)
        mask = torch.empty(B2, 1, H, W, device=gray2.device, dtype=torch.float32)
        with torch.cuda.device_of(gray2):
            _call('arflow_census_warp_pair_fwd', _p(gray2), _p(flow2), fbs, _p(occ_small2), _p(mask), _p(dham), _p(buf), B2, H,
                  W, r, _stream(), key=(B2, H, W))
        sums = _fold_sums(buf, 4)
        den = torch.stack([_ddp.global_denominator(sums[1]), _ddp.global_denominator(sums[3])]) + 1e-6 / _ddp.world_size()
        inv = 1.0 / den
        ctx.save_for_backward(gray2, flow2, dham, inv)
        ctx.r, ctx.fbs = r, fbs
        ctx.mark_non_differentiable(mask)
        return sums[0] * inv[0], sums[2] * inv[1], mask

    @staticmethod
    def backward(ctx, g0, g1, gmask_unused):
        gray2, flow2, dham, inv = ctx.saved_tensors
        B2, _, H, W = gray2.shape
        scale = (torch.stack([g0.reshape(()), g1.reshape(())]) * inv).contiguous()
        gflow = torch.empty(B2, 2, H, W, device=gray2.device, dtype=torch.float32)
        with torch.cuda.device_of(gray2):
            _call('arflow_census_warp_pair_bwd', _p(gray2), _p(flow2), ctx.fbs, _p(dham), _p(scale), _p(gflow), B2, H, W, ctx.r,
                  _stream(), key=(B2, H, W))
        return None, gflow, None, None


def census_warp_pair_loss(gray2, flow2, occ_small2, patch_size=7):
    return CensusWarpPairLossFunction.apply(gray2, flow2, occ_small2, patch_size)


class UFlowPairLossFunction(torch.autograd.Function):
    """Both loss terms of UFlowLoss for both directions (losses/uflow_loss.py:30-102) behind ONE autograd node: forward =
    arflow_splat_smooth_fwd (range maps + smoothness sums of the level-2 flows) + arflow_census_warp_pair_fwd; backward = ONE
    launch (arflow_uflow_pair_bwd: census + warp backward and smoothness backward as workgroup roles).  Batch layout as
    CensusWarpPairLossFunction (sample s = 2 b + direction).  Returns (census loss fw, census loss bw, smoothness sums [2],
    mask [2B,1,H,W]); gradients w.r.t. the level-0 and level-2 flows."""

    @staticmethod
    def forward(ctx, gray2, small2, flow0, flow2, occ_zeroed, alpha, order, patch_size):
        _need_gpu(gray2, small2, flow0, flow2)
        gray2, small2 = gray2.detach().contiguous(), small2.detach().contiguous()
        flow0, fbs0 = _flow_view(flow0)
        flow2, fbs2 = _flow_view(flow2)
        B2, _, H, W = gray2.shape
        h, w = flow2.shape[2:]
        if B2 % 2 or flow0.shape != (B2, 2, H, W) or (h, w) != (H // 4, W // 4) or small2.shape != (B2, 3, h, w):
            raise ValueError('uflow_pair_loss: inconsistent shapes')
        r = int(patch_size) // 2
        pre = occ_zeroed is not None
        occ = occ_zeroed if pre else torch.empty(B2, 1, h, w, device=gray2.device, dtype=torch.float32)
        sbuf = _new_sums(gray2.device, B2, h, w)
        cbuf = _new_sums(gray2.device, B2, H, W)
        dham = torch.empty(B2, 1, H, W, device=gray2.device, dtype=torch.float32)
        mask = torch.empty(B2, 1, H, W, device=gray2.device, dtype=torch.float32)
        with torch.cuda.device_of(gray2):
            _call('arflow_splat_smooth_fwd', _p(flow2), _p(small2), _p(occ), _p(sbuf), B2, h, w, fbs2, 1.0, float(alpha),
                  int(order), 1, 1, int(pre), _stream(), key=(B2, 3, h, w))
            _call('arflow_census_warp_pair_fwd', _p(gray2), _p(flow0), fbs0, _p(occ), _p(mask), _p(dham), _p(cbuf), B2, H, W, r,
                  _stream(), key=(B2, H, W))
        sums = _fold_sums(cbuf, 4)
        den = torch.stack([_ddp.global_denominator(sums[1]), _ddp.global_denominator(sums[3])]) + 1e-6 / _ddp.world_size()
        inv = 1.0 / den
        ctx.save_for_backward(gray2, small2, flow0, flow2, dham, inv)
        ctx.cfg = (r, fbs0, fbs2, float(alpha), int(order))
        ctx.mark_non_differentiable(mask)
        return sums[0] * inv[0], sums[2] * inv[1], _fold_sums(sbuf, 2), mask

    @staticmethod
    def backward(ctx, g0, g1, gs, gmask_unused):
        gray2, small2, flow0, flow2, dham, inv = ctx.saved_tensors
        r, fbs0, fbs2, alpha, order = ctx.cfg
        B2, _, H, W = gray2.shape
        h, w = flow2.shape[2:]
        scale = (torch.stack([g0.reshape(()), g1.reshape(())]) * inv).contiguous()
        coef = gs.contiguous()
        gf0 = torch.empty(B2, 2, H, W, device=gray2.device, dtype=torch.float32)
        gf2 = torch.empty(B2, 2, h, w, device=gray2.device, dtype=torch.float32)
        with torch.cuda.device_of(gray2):
            _call('arflow_uflow_pair_bwd', _p(gray2), _p(flow0), fbs0, _p(dham), _p(scale), _p(gf0), B2, H, W, r, _p(flow2),
                  fbs2, _p(small2), _p(coef), _p(gf2), h, w, 1.0, alpha, order, 1, 1, _stream(), key=(B2, H, W))
        return None, None, gf0, gf2, None, None, None, None


def uflow_pair_loss(gray2, small2, flow0, flow2, occ_zeroed, alpha, order, patch_size=7):
    return UFlowPairLossFunction.apply(gray2, small2, flow0, flow2, occ_zeroed, alpha, order, patch_size)


def census_warp_supported(H, W):
    return bool(_lib.load().arflow_census_warp_supported(int(H), int(W)))


def census_warp_loss(gray_a, gray_b, flow, occ_small, patch_size=7):
    return CensusWarpLossFunction.apply(gray_a, gray_b, flow, occ_small, patch_size)


def down4_gray(img, want_small=True, zero_plane=False):
    """(downsample(img, x1/4) or None, rgb_to_grayscale(img) * 255) from ONE read of the image (no gradient: the
    reference applies both to data only, losses/uflow_loss.py:59-60, utils/uflow_utils.py:248).  zero_plane=True also
    returns a zero-filled [B,1,H/4,W/4] plane cleared by the same launch (the splat target of splat_smooth)."""
    _need_gpu(img)
    img = img.detach().contiguous()
    B, C, H, W = img.shape
    if C != 3:
        raise ValueError('down4_gray expects a [B,3,H,W] image')
    small = torch.empty(B, 3, H // 4, W // 4, device=img.device, dtype=torch.float32) if want_small else None
    gray = torch.empty(B, 1, H, W, device=img.device, dtype=torch.float32)
    zero = torch.empty(B, 1, H // 4, W // 4, device=img.device, dtype=torch.float32) if zero_plane else None
    with torch.cuda.device_of(img):
        _call('arflow_down4_gray_z', _p(img), _p(small), _p(gray), _p(zero), B, H, W, _stream(), key=(B, H, W))
    return (small, gray, zero) if zero_plane else (small, gray)


class TernaryDistFunction(torch.autograd.Function):
    """Per-pixel soft census distance (sum over the patch); TernaryLoss core,
    losses/loss_blocks.py:12-62."""

    @staticmethod
    def forward(ctx, im_a, im_b, radius):
        _need_gpu(im_a, im_b)
        im_a, im_b = im_a.contiguous(), im_b.contiguous()
        B, C, H, W = im_a.shape
        if C != 3 or im_b.shape != im_a.shape:
            raise ValueError('ternary distance expects two [B,3,H,W] images')
        ham = torch.empty(B, 1, H, W, device=im_a.device, dtype=torch.float32)
        with torch.cuda.device_of(im_a):
            _call('arflow_census_fwd', _p(im_a), _p(im_b), None, _p(ham), None, None, B, H, W, int(radius), _stream())
        ctx.save_for_backward(im_a, im_b)
        ctx.r = int(radius)
        return ham

    @staticmethod
    def backward(ctx, gham):
        im_a, im_b = ctx.saved_tensors
        B, _, H, W = im_a.shape
        gham = gham.contiguous()
        ga = gb = None
        with torch.cuda.device_of(im_a):
            if ctx.needs_input_grad[0]:
                ga = torch.empty_like(im_a)
                _call('arflow_census_bwd', _p(im_b), _p(im_a), _p(gham), None, _p(ga), B, H, W, ctx.r, _stream())
            if ctx.needs_input_grad[1]:
                gb = torch.empty_like(im_b)
                _call('arflow_census_bwd', _p(im_a), _p(im_b), _p(gham), None, _p(gb), B, H, W, ctx.r, _stream())
        return ga, gb, None


# ------------------------------------------------------------------------------------------------
class PhotoSumsFunction(torch.autograd.Function):
    """[sum |im-recons|*mask, sum SSIMdist(recons*mask, im*mask), sum mask] in one launch
    (losses/flow_loss.py:13-27).  Gradient w.r.t. recons only."""

    @staticmethod
    def forward(ctx, im, recons, mask):
        _need_gpu(im, recons, mask)
        im, recons = im.contiguous(), recons.contiguous()
        mask = None if mask is None else mask.contiguous()
        B, C, H, W = im.shape
        buf = _new_sums(im.device, B, H, W)
        with torch.cuda.device_of(im):
            _call('arflow_photo_fwd', _p(im), _p(recons), _p(mask), None, _p(buf), B, C, H, W, _stream(),
                  key=(B, C, H, W))
        ctx.save_for_backward(im, recons, mask)
        return _fold_sums(buf, 3)

    @staticmethod
    def backward(ctx, gsums):
        im, recons, mask = ctx.saved_tensors
        B, C, H, W = im.shape
        coef = gsums[:2].contiguous()
        g = torch.empty_like(recons)
        with torch.cuda.device_of(im):
            _call('arflow_photo_bwd', _p(im), _p(recons), _p(mask), None, _p(coef), _p(g), B, C, H, W, _stream(),
                  key=(B, C, H, W))
        return None, g, None


class SSIMAnyFunction(torch.autograd.Function):
    """SSIM(x, y, md) distance map of losses/loss_blocks.py:65-84 for any md (md = 1: SSIMFunction, the tiled kernel)."""

    @staticmethod
    def forward(ctx, x, y, md):
        _need_gpu(x, y)
        x, y = x.contiguous(), y.contiguous()
        B, C, H, W = x.shape
        md = int(md)
        out = torch.empty(B, C, H - 2 * md, W - 2 * md, device=x.device, dtype=torch.float32)
        with torch.cuda.device_of(x):
            _call('arflow_ssim_fwd', _p(x), _p(y), _p(out), B, C, H, W, md, _stream())
        ctx.save_for_backward(x, y)
        ctx.md = md
        return out

    @staticmethod
    def backward(ctx, gmap):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        gmap = gmap.contiguous()
        gx = gy = None
        with torch.cuda.device_of(x):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _call('arflow_ssim_bwd', _p(x), _p(y), _p(gmap), _p(gx), B, C, H, W, ctx.md, _stream())
            if ctx.needs_input_grad[1]:
                gy = torch.empty_like(y)  # SSIM is symmetric in its arguments
                _call('arflow_ssim_bwd', _p(y), _p(x), _p(gmap), _p(gy), B, C, H, W, ctx.md, _stream())
        return gx, gy, None


class SSIMFunction(torch.autograd.Function):
    """SSIM(x, y, md=1) distance map of losses/loss_blocks.py:65-84."""

    @staticmethod
    def forward(ctx, x, y):
        _need_gpu(x, y)
        x, y = x.contiguous(), y.contiguous()
        B, C, H, W = x.shape
        out = torch.empty(B, C, H - 2, W - 2, device=x.device, dtype=torch.float32)
        sums = _new_sums(x.device, B, H, W)
        with torch.cuda.device_of(x):
            # kernel convention: SSIM(recons*mask, im*mask) -> x plays "recons", y plays "im"
            _call('arflow_photo_fwd', _p(y), _p(x), None, _p(out), _p(sums), B, C, H, W, _stream())
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, gmap):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        gmap = gmap.contiguous()
        coef = torch.zeros(2, device=x.device, dtype=torch.float32)
        gx = gy = None
        with torch.cuda.device_of(x):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                _call('arflow_photo_bwd', _p(y), _p(x), None, _p(gmap), _p(coef), _p(gx), B, C, H, W, _stream())
            if ctx.needs_input_grad[1]:
                gy = torch.empty_like(y)  # SSIM is symmetric in its arguments
                _call('arflow_photo_bwd', _p(x), _p(y), None, _p(gmap), _p(coef), _p(gy), B, C, H, W, _stream())
        return gx, gy


# ------------------------------------------------------------------------------------------------
class SmoothSumsFunction(torch.autograd.Function):
    """[sum wx*pen(Dx flow), sum wy*pen(Dy flow)] (losses/loss_blocks.py:93-124,
    losses/uflow_loss.py:62-102).  Gradient w.r.t. flow only (the image is detached / data)."""

    @staticmethod
    def forward(ctx, flow, img, flow_scale, alpha, order, wmode, penalty):
        _need_gpu(flow, img)
        flow, fbs = _flow_view(flow)
        img = img.contiguous()
        B, _, H, W = flow.shape
        Ci = img.shape[1]
        if img.shape[0] != B or img.shape[2:] != flow.shape[2:]:
            raise ValueError('smoothness: image and flow must share batch and spatial size')
        buf = _new_sums(flow.device, B, H, W)
        args = (B, Ci, H, W, fbs, float(flow_scale), float(alpha), int(order), int(wmode), int(penalty))
        with torch.cuda.device_of(flow):
            _call('arflow_smooth_fwd', _p(flow), _p(img), _p(buf), *args, _stream(), key=(B, Ci, H, W))
        ctx.save_for_backward(flow, img)
        ctx.args = args
        return _fold_sums(buf, 2)

    @staticmethod
    def backward(ctx, gsums):
        flow, img = ctx.saved_tensors
        B, _, H, W = flow.shape
        coef = gsums.contiguous()
        g = torch.empty(B, 2, H, W, device=flow.device, dtype=torch.float32)
        with torch.cuda.device_of(flow):
            _call('arflow_smooth_bwd', _p(flow), _p(img), _p(coef), _p(g), *ctx.args, _stream(),
                  key=(B, img.shape[1], H, W))
        return g, None, None, None, None, None, None


def smooth_sums(flow, img, flow_scale, alpha, order, wmode, penalty):
    return SmoothSumsFunction.apply(flow, img, flow_scale, alpha, order, wmode, penalty)


class SplatSmoothFunction(torch.autograd.Function):
    """(smoothness sums of SmoothSumsFunction, compute_range_map(flow)) from ONE launch (arflow_splat_smooth_fwd): UFlowLoss
    takes both from the same level-2 flows.  ``range_out``: a ZERO-FILLED [B,1,H,W] plane (down4_gray(zero_plane=True)) or
    None (cleared here).  The range map carries no gradient (the reference detaches it, losses/uflow_loss.py:43)."""

    @staticmethod
    def forward(ctx, flow, img, range_out, flow_scale, alpha, order, wmode, penalty):
        _need_gpu(flow, img)
        flow, fbs = _flow_view(flow)
        img = img.contiguous()
        B, _, H, W = flow.shape
        if img.shape != (B, 3, H, W):
            raise ValueError('splat_smooth expects a [B,3,H,W] image at the resolution of the flow')
        pre = range_out is not None
        if not pre:
            range_out = torch.empty(B, 1, H, W, device=flow.device, dtype=torch.float32)
        buf = _new_sums(flow.device, B, H, W)
        args = (B, 3, H, W, fbs, float(flow_scale), float(alpha), int(order), int(wmode), int(penalty))
        with torch.cuda.device_of(flow):
            _call('arflow_splat_smooth_fwd', _p(flow), _p(img), _p(range_out), _p(buf), B, H, W, fbs, float(flow_scale),
                  float(alpha), int(order), int(wmode), int(penalty), int(pre), _stream(), key=(B, 3, H, W))
        ctx.save_for_backward(flow, img)
        ctx.args = args
        ctx.mark_non_differentiable(range_out)
        return _fold_sums(buf, 2), range_out

    @staticmethod
    def backward(ctx, gsums, g_unused):
        flow, img = ctx.saved_tensors
        B, _, H, W = flow.shape
        coef = gsums.contiguous()
        g = torch.empty(B, 2, H, W, device=flow.device, dtype=torch.float32)
        with torch.cuda.device_of(flow):
            _call('arflow_smooth_bwd', _p(flow), _p(img), _p(coef), _p(g), *ctx.args, _stream(), key=(B, 3, H, W))
        return g, None, None, None, None, None, None, None


def splat_smooth(flow, img, range_out, flow_scale, alpha, order, wmode, penalty):
    return SplatSmoothFunction.apply(flow, img, range_out, flow_scale, alpha, order, wmode, penalty)


# ------------------------------------------------------------------------------------------------
def down4(img):
    """downsample(img, is_flow=False, scale_factor=4) for H, W multiples of 4 (no gradient: the
    reference only applies it to images, losses/uflow_loss.py:59-60)."""
    _need_gpu(img)
    img = img.detach().contiguous()
    B, C, H, W = img.shape
    out = torch.empty(B, C, H // 4, W // 4, device=img.device, dtype=torch.float32)
    with torch.cuda.device_of(img):
        _call('arflow_down4', _p(img), _p(out), B * C, H, W, _stream(), key=(B * C, H, W))
    return out


def up4_clamp_mul(small, valid=None):
    """upsample(clamp(small,0,1), x4) * valid -- losses/uflow_loss.py:41-48, no gradient."""
    _need_gpu(small, valid)
    small = small.detach().contiguous()
    B, one, h, w = small.shape
    assert one == 1
    valid = None if valid is None else valid.detach().contiguous()
    out = torch.empty(B, 1, 4 * h, 4 * w, device=small.device, dtype=torch.float32)
    with torch.cuda.device_of(small):
        _call('arflow_up4_clamp_mul', _p(small), _p(valid), _p(out), B, h, w, _stream(), key=(B, h, w))
    return out
