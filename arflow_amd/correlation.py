"""Drop-in ``Correlation`` module backed by the gfx950 cost-volume kernels.

Swap point in the reference: the import pair at models/pwclite.py:6-7 (README.md:30).  Accepts
the keyword set every call site uses (models/pwclite.py:124-126, models/pwclite_uflow.py:147-149):
``Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1)``.
"""
import torch.nn as nn

from . import functional as AF


class Correlation(nn.Module):
    def __init__(self, max_displacement=4, *args, pad_size=None, kernel_size=1, stride1=1, stride2=1, corr_multiply=1,
                 storage_dtype=None, **kwargs):
        """Signature of models/correlation_native.py:7 -- ``max_displacement`` first, further POSITIONAL arguments swallowed
        like there (``Correlation(3)`` is the d = 3 volume) -- with the keyword arguments every model passes
        (models/pwclite.py:124-125: pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply).
        correlation_native ignores those keywords; here a NON-default (pad_size, kernel_size, stride1, stride2) is
        honoured with the semantics of the CUDA extension the keywords come from (correlation_cuda.cc:10-16; parity
        unpinned), and ``pad_size`` / ``output_dim`` describe the volume actually computed.
        storage_dtype=torch.bfloat16 (opt-in, not in the reference's signature): keep the features as bf16 in
        HBM -- fp32 accumulation, fp32 volume and gradients: footprint only, not speed."""
        super().__init__()
        self.storage = storage_dtype
        self.max_displacement = int(max_displacement)
        if pad_size is None:
            pad_size = self.max_displacement
        self.general = None
        self.output_dim = 2 * self.max_displacement + 1
        self.pad_size = self.max_displacement
        if (kernel_size, stride1, stride2) != (1, 1, 1) or pad_size != self.max_displacement:
            if kernel_size % 2 != 1:
                raise ValueError('kernel_size must be odd')
            self.general = (int(pad_size), int(kernel_size), self.max_displacement, int(stride1), int(stride2))
            self.pad_size = int(pad_size)
            self.output_dim = 2 * (self.max_displacement // int(stride2)) + 1  # displacements per axis (correlation_cuda.cc:31-34)

    def forward(self, x1, x2, negative_slope=1.0):
        """``negative_slope`` != 1 fuses the LeakyReLU the callers apply to the volume
        (models/pwclite.py:183-184) into the kernel; the default is the reference's plain volume."""
        if self.general is not None:
            out = AF.correlation_general(x1, x2, *self.general)
            return out if negative_slope == 1.0 else nn.functional.leaky_relu(out, negative_slope)
        return AF.correlation(x1, x2, self.max_displacement, negative_slope, storage=self.storage)

    def concat(self, x1, x2, before=(), after=(), negative_slope=1.0):
        """torch.cat([*before, forward(x1, x2, negative_slope), *after], 1) -- what every decoder does with the volume
        next (models/pwclite.py:187-189, models/pwclite_uflow.py:218-222) -- with the volume written by the kernel
        straight into its channel slot of the concatenated tensor instead of being copied there."""
        if self.storage is not None or self.general is not None:
            import torch
            return torch.cat(list(before) + [self.forward(x1, x2, negative_slope)] + list(after), 1)
        return AF.correlation_concat(x1, x2, before, after, self.max_displacement, negative_slope)


def cost_volume_concat(features1, features2, before, after, max_displacement, negative_slope=1.0):
    """torch.cat([*before, compute_cost_volume(...), *after], 1) without the copy of the volume
    (models/uflow_model.py:175-198)."""
    _, _, height, _ = features1.shape
    if max_displacement <= 0 or max_displacement >= height:
        raise ValueError(f'Max displacement of {max_displacement} is too large.')
    return AF.correlation_concat(features1, features2, before, after, max_displacement, negative_slope)


def compute_cost_volume(features1, features2, max_displacement, negative_slope=1.0):
    """models/uflow_model.py:53-92 (same arithmetic as Correlation, NCHW); optional fused LeakyReLU."""
    _, _, height, _ = features1.shape
    if max_displacement <= 0 or max_displacement >= height:
        raise ValueError(f'Max displacement of {max_displacement} is too large.')
    return AF.correlation(features1, features2, max_displacement, negative_slope)
