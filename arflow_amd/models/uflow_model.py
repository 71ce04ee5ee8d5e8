"""PWCFlow (UFlow port) host model on the gfx950 ops; contract of models/uflow_model.py:96-470.
This is what configs/chairs_uflow.json instantiates (``"model": {"type": "uflow"}``)."""
import torch
import torch.nn as nn
import torch.nn.functional as func

from .. import functional as AF
from .. import uflow_utils
from ..correlation import cost_volume_concat
from . import blocks
from .blocks import ConvAct, init_conv_weights, pair_batches


def normalize_features(feature_list, normalize, center, moments_across_channels, moments_across_images):
    """models/uflow_model.py:8-50.  The configuration PWCFlow uses (:167-172) runs as one fused HIP op."""
    if (normalize and center and moments_across_channels and moments_across_images and len(feature_list) == 2
            and feature_list[0].shape == feature_list[1].shape):
        return list(AF.normalize_pair(feature_list[0], feature_list[1], 'avg'))
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


class PWCFeaturePyramid(nn.Module):
    """models/uflow_model.py:350-470 with the defaults PWCFlow uses: 5 levels of three 3x3 VALID convs
    (explicit zero pad 1), 32 filters, stride 2 on the first conv of each level, LeakyReLU(0.1)."""

    def __init__(self, leaky_relu_alpha=0.1, num_levels=5, num_channels=3):
        super().__init__()
        self._leaky_relu_alpha = leaky_relu_alpha
        self._convs = nn.ModuleList()
        c = num_channels
        for _ in range(num_levels):
            group = nn.ModuleList()
            for i in range(3):
                group.append(nn.Conv2d(c, 32, kernel_size=(3, 3), stride=2 if i == 0 else 1, padding='valid'))
                c = 32
            self._convs.append(group)

    pyramid_moments = None  # per level: [B, rows, 2] partial (sum, sum of squares) of that level's map, or None

    def forward(self, x):
        x = x * 2. - 1.
        features, moms = [], []
        for group in self._convs:
            mom = None
            for i, conv in enumerate(group):
                # the reference zero-pads by 1 explicitly and convolves 'valid' (models/uflow_model.py:453-457):
                # the same numbers as padding=1 inside the convolution, without the padded copy;
                # bias-free conv + fused bias / LeakyReLU pass
                y = func.conv2d(x, conv.weight, None, conv.stride, 1, conv.dilation)
                if i == len(group) - 1 and y.is_cuda and blocks.bias_act is AF.bias_leaky_relu:
                    # the conv that OUTPUTS a pyramid level: its epilogue also leaves normalize_features' sums of that map
                    x, mom = AF.bias_leaky_relu_moments(y, conv.bias, self._leaky_relu_alpha)
                else:
                    x = blocks.bias_act(y, conv.bias, self._leaky_relu_alpha)
            features.append(x)
            moms.append(mom)
        self.pyramid_moments = moms
        return features


class PWCFlow(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self._leaky_relu_alpha = 0.1
        self._drop_out_rate = cfg.level_dropout
        self._num_context_up_channels = 32
        self._num_levels = 5
        self._normalize_before_cost_volume = cfg.feature_norm
        # registration order matters for state_dict order: refine, flow layers, upsample, pyramid
        self._refine_model = self._build_refinement_model()
        self._flow_layers = self._build_flow_layers()
        self._context_up_layers = nn.ModuleList(
            [nn.ConvTranspose2d(32, 32, kernel_size=(4, 4), stride=2, padding=1) for _ in range(self._num_levels)])
        self._feature_pyramid_extractor = PWCFeaturePyramid()

    def init_weights(self):
        init_conv_weights(self, 'xavier')

    def _build_flow_layers(self):
        """models/uflow_model.py:297-323: per level 5 dense 3x3 blocks (128,128,96,64,32) and a 32->2
        head that sees only the last block."""
        result = nn.ModuleList([None])
        for i in range(1, self._num_levels):
            layers = nn.ModuleList()
            c_in = 81 + 32 + (0 if i == self._num_levels - 1 else 2 + self._num_context_up_channels)
            for c in (128, 128, 96, 64, 32):
                layers.append(ConvAct(nn.Conv2d(c_in, c, kernel_size=(3, 3), stride=1, padding='same'),
                                      nn.LeakyReLU(negative_slope=self._leaky_relu_alpha)))
                c_in += c
            layers.append(nn.Conv2d(32, 2, kernel_size=(3, 3), padding='same'))
            result.append(layers)
        return result

    def _build_refinement_model(self):
        """models/uflow_model.py:325-348."""
        layers = []
        c_in = 32 + 2
        for c, d in [(128, 1), (128, 2), (128, 4), (96, 8), (64, 16), (32, 1)]:
            layers.append(nn.Conv2d(c_in, c, kernel_size=(3, 3), stride=1, padding='same', dilation=d))
            layers.append(nn.LeakyReLU(negative_slope=self._leaky_relu_alpha))
            c_in = c
        layers.append(nn.Conv2d(c_in, 2, kernel_size=(3, 3), stride=1, padding='same'))
        return nn.ModuleList(layers)

    def _drops(self, n_passes, batch_per_pass, device):
        n = (self._num_levels - 1) + 1  # one per level 4..1, then one for the refinement
        if not (self.training and self._drop_out_rate > 0):
            return None
        vals = [[float(torch.rand(1) > self._drop_out_rate) for _ in range(n)] for _ in range(n_passes)]
        t = torch.tensor(vals, dtype=torch.float32).repeat_interleave(batch_per_pass, dim=0).t().contiguous()
        return t.to(device, non_blocking=True).view(n, -1, 1, 1, 1)

    def forward_2_frames(self, feature_pyramid1, feature_pyramid2, drops=None, moments=None):
        """models/uflow_model.py:138-245.  moments: optional (rows of the first maps, rows of the second maps) per pyramid
        level -- the feature maps' partial moments from the extractor's conv epilogues (PWCFeaturePyramid.pyramid_moments)."""
        context = flow = flow_up = context_up = None
        flows = []
        k = 0
        top = self._num_levels - 1
        for level in range(top, 0, -1):
            features1, features2 = feature_pyramid1[level], feature_pyramid2[level]
            fused = self._normalize_before_cost_volume and AF.level_supported(features1, None if level == top else flow, True)
            if fused:
                # the whole level in front of the flow layers as one call: x2 flow upsample
                # (uflow_utils.upsample: align_corners=False), resample, normalisation, cost volume + LeakyReLU
                r1 = moments[0][level] if moments is not None else None
                r2 = moments[1][level] if moments is not None else None
                if level == top:
                    cfg = AF.LevelCfg(['vol', 0], 'avg', self._leaky_relu_alpha, 4)
                    x_in = AF.level(features1, features2, None, cfg, features1, x1_rows=r1, x2_rows=r2)
                else:
                    cfg = AF.LevelCfg([0, 'flow', 'vol', 1], 'avg', self._leaky_relu_alpha, 4, True, False, 'zeros', True,
                                      AF.NORM_UFLOW)
                    x_in, flow_up = AF.level(features1, features2, flow, cfg, context_up, features1, x1_rows=r1)
            else:
                if level != top:
                    flow_up = uflow_utils.upsample(flow, is_flow=True)
                warped2 = features2 if level == top else uflow_utils.resample_flow(features2, flow_up)
                f1n, w2n = normalize_features([features1, warped2], normalize=self._normalize_before_cost_volume,
                                              center=self._normalize_before_cost_volume,
                                              moments_across_channels=True, moments_across_images=True)
                # cost volume + fused LeakyReLU, written straight into its slot of the decoder's concatenated input
                before = () if level == top else (context_up, flow_up)
                x_in = cost_volume_concat(f1n, w2n, before, (features1,), max_displacement=4,
                                          negative_slope=self._leaky_relu_alpha)
            layers = self._flow_layers[level]
            x_out = None
            for layer in layers[:-1]:
                x_out = layer(x_in)
                x_in = torch.cat([x_in, x_out], dim=1)
            context = x_out
            flow = layers[-1](context)
            if drops is not None:
                context = context * drops[k]
                flow = flow * drops[k]
            k += 1
            if level != top:
                flow = flow + flow_up
            context_up = self._context_up_layers[level](context)
            flows.insert(0, flow)
        refinement = torch.cat([context, flow], dim=1)
        mods = list(self._refine_model)
        i = 0
        while i < len(mods):  # Conv2d followed by LeakyReLU -> bias-free conv + fused bias / LeakyReLU pass
            m = mods[i]
            if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU):
                refinement = blocks.bias_act(func.conv2d(refinement, m.weight, None, m.stride, m.padding, m.dilation),
                                             m.bias, mods[i + 1].negative_slope)
                i += 2
            else:
                refinement = m(refinement)
                i += 1
        if drops is not None:
            refinement = refinement * drops[k]
        flows[0] = flow + refinement
        flows.insert(0, uflow_utils.upsample(flows[0], is_flow=True))
        flows.insert(0, uflow_utils.upsample(flows[0], is_flow=True))
        return flows

    def forward(self, x, with_bk=True):
        n_frames = x.size(1) // 3
        if n_frames != 2:
            raise NotImplementedError
        B = x.size(0)
        pyr_all = self._feature_pyramid_extractor(torch.cat([x[:, 0:3], x[:, 3:6]], 0))
        moms = self._feature_pyramid_extractor.pyramid_moments
        have_m = moms is not None and all(t is not None for t in moms)
        p1 = [p[:B] for p in pyr_all]
        p2 = [p[B:] for p in pyr_all]
        res = {}
        if with_bk:
            # first maps of the 2B (fw; bw) samples = the extractor's batch as it stands (no copy), second maps = its two
            # halves swapped; the same holds for the rows of partial moments
            a = list(pyr_all)
            b = [torch.cat([p[B:], p[:B]], 0) for p in pyr_all]
            m = (list(moms), [torch.roll(t, B, 0) for t in moms]) if have_m else None
            flows = self.forward_2_frames(a, b, self._drops(2, B, x.device), m)
            res['flows_fw'] = [f[:B] for f in flows]
            res['flows_bw'] = [f[B:] for f in flows]
        else:
            m = ([t[:B] for t in moms], [t[B:] for t in moms]) if have_m else None
            res['flows_fw'] = self.forward_2_frames(p1, p2, self._drops(1, B, x.device), m)
        return res
