"""Shared conv building blocks of the PWC-style host models (plain torch.nn; MIOpen does the convs).
Module / parameter names follow the reference so its checkpoints load by name."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF

bias_act = AF.bias_leaky_relu  # (conv output, bias, slope) -> activation; substituted by oracle.host_models on CPU


class ConvAct(nn.Sequential):
    """Sequential(Conv2d(bias=True), LeakyReLU) with the reference's parameter names (``0.weight``,
    ``0.bias``), evaluated as bias-free MIOpen convolution + ONE fused bias/LeakyReLU pass (and one fused
    LeakyReLU-derivative/bias-gradient pass backward) instead of conv, bias add, LeakyReLU and a separate
    full-tensor reduction for the bias gradient."""

    want_moments = False  # set on the conv that OUTPUTS a pyramid level: its epilogue also leaves that map's moments
    last_moments = None

    def forward(self, x):
        c, act = self[0], self[1]
        y = F.conv2d(x, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
        if self.want_moments and y.is_cuda and bias_act is AF.bias_leaky_relu:
            y, self.last_moments = AF.bias_leaky_relu_moments(y, c.bias, act.negative_slope)
            return y
        self.last_moments = None
        return bias_act(y, c.bias, act.negative_slope)


def conv(in_planes, out_planes, kernel_size=3, stride=1, dilation=1, isReLU=True):
    """models/pwclite.py:10-23 -- Sequential(Conv2d[, LeakyReLU(0.1)]) -> keys ``<name>.0.weight``."""
    layers = [nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, dilation=dilation,
                        padding=((kernel_size - 1) * dilation) // 2, bias=True)]
    if isReLU:
        layers.append(nn.LeakyReLU(0.1, inplace=True))
        return ConvAct(*layers)
    return nn.Sequential(*layers)


def deconv(in_planes, out_planes, kernel_size=4, stride=2, padding=1):
    """models/pwclite_uflow.py:26-27."""
    return nn.ConvTranspose2d(in_planes, out_planes, kernel_size, stride, padding, bias=True)


class FeatureExtractor(nn.Module):
    """models/pwclite.py:26-45 (2 convs per level) / models/pwclite_uflow.py:40-62 (3 convs per level,
    input rescaled to [-1,1])."""

    def __init__(self, num_chs, convs_per_level=2, rescale_input=False, moments=False):
        """moments=True (not in the reference's signature): the last conv of every level also leaves the partial moments
        of its output (normalize_features' sums, taken in the fused bias/LeakyReLU epilogue) in `pyramid_moments`,
        coarsest first like the pyramid -- [B, rows, 2] float64 per level, or None off the GPU path."""
        super().__init__()
        self.num_chs = num_chs
        self.rescale_input = rescale_input
        self.convs = nn.ModuleList()
        for ch_in, ch_out in zip(num_chs[:-1], num_chs[1:]):
            layers = [conv(ch_in, ch_out, stride=2)] + [conv(ch_out, ch_out) for _ in range(convs_per_level - 1)]
            layers[-1].want_moments = bool(moments)
            self.convs.append(nn.Sequential(*layers))
        self.pyramid_moments = None

    def forward(self, x):
        if self.rescale_input:
            x = x * 2. - 1.
        pyramid, moms = [], []
        for level in self.convs:
            x = level(x)
            pyramid.append(x)
            moms.append(level[-1].last_moments)
        self.pyramid_moments = moms[::-1]
        return pyramid[::-1]


class FlowEstimatorDense(nn.Module):
    """models/pwclite.py:48-66."""

    def __init__(self, ch_in):
        super().__init__()
        self.conv1 = conv(ch_in, 128)
        self.conv2 = conv(ch_in + 128, 128)
        self.conv3 = conv(ch_in + 256, 96)
        self.conv4 = conv(ch_in + 352, 64)
        self.conv5 = conv(ch_in + 416, 32)
        self.feat_dim = ch_in + 448
        self.conv_last = conv(ch_in + 448, 2, isReLU=False)

    def forward(self, x):
        for layer in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            x = torch.cat([layer(x), x], dim=1)
        return x, self.conv_last(x)


class FlowEstimatorReduce(nn.Module):
    """models/pwclite.py:69-88."""

    def __init__(self, ch_in):
        super().__init__()
        self.conv1 = conv(ch_in, 128)
        self.conv2 = conv(128, 128)
        self.conv3 = conv(128 + 128, 96)
        self.conv4 = conv(128 + 96, 64)
        self.conv5 = conv(96 + 64, 32)
        self.feat_dim = 32
        self.predict_flow = conv(64 + 32, 2, isReLU=False)

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = self.conv2(x1)
        x3 = self.conv3(torch.cat([x1, x2], dim=1))
        x4 = self.conv4(torch.cat([x2, x3], dim=1))
        x5 = self.conv5(torch.cat([x3, x4], dim=1))
        return x5, self.predict_flow(torch.cat([x4, x5], dim=1))


class ContextNetwork(nn.Module):
    """models/pwclite.py:91-106 -- 7 dilated convs."""

    def __init__(self, ch_in):
        super().__init__()
        self.convs = nn.Sequential(
            conv(ch_in, 128, 3, 1, 1), conv(128, 128, 3, 1, 2), conv(128, 128, 3, 1, 4), conv(128, 96, 3, 1, 8),
            conv(96, 64, 3, 1, 16), conv(64, 32, 3, 1, 1), conv(32, 2, isReLU=False))

    def forward(self, x):
        return self.convs(x)


def init_conv_weights(module, scheme):
    """kaiming-normal (models/pwclite.py:149-159) or xavier-uniform (models/pwclite_uflow.py:179-191),
    zero bias, over Conv2d and ConvTranspose2d in named_modules() order."""
    for layer in module.modules():
        if isinstance(layer, (nn.Conv2d, nn.ConvTranspose2d)):
            if scheme == 'kaiming':
                nn.init.kaiming_normal_(layer.weight)
            else:
                nn.init.xavier_uniform_(layer.weight)
            if layer.bias is not None:
                nn.init.constant_(layer.bias, 0)


def pair_batches(pyr1, pyr2):
    """Stack (frame1, frame2) and (frame2, frame1) along the batch axis so that the forward and the
    backward flow are estimated in ONE pass at batch 2B -- every op on the path is per-sample, so the
    result equals the reference's two sequential passes (models/pwclite.py:267-269) while every launch
    covers twice the pixels."""
    a = [torch.cat([p, q], 0) for p, q in zip(pyr1, pyr2)]
    b = [torch.cat([q, p], 0) for p, q in zip(pyr1, pyr2)]
    return a, b
