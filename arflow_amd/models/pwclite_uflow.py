"""PWCLiteUflow host model on the gfx950 ops; contract of models/pwclite_uflow.py:126-267."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF
from ..correlation import Correlation
from ..warp_utils import flow_warp
from .blocks import (ContextNetwork, FeatureExtractor, FlowEstimatorDense, FlowEstimatorReduce, deconv,
                     init_conv_weights)


def normalize_features(features_list):
    """models/pwclite_uflow.py:30-38 -- per-sample moments of the channel-concatenated pair.  The pair the
    model normalises (two tensors of equal shape) runs as one fused HIP op; other list shapes keep the
    tensor expression."""
    if len(features_list) == 2 and features_list[0].shape == features_list[1].shape:
        return list(AF.normalize_pair(features_list[0], features_list[1], 'joint'))
    n = sum(f[0].numel() for f in features_list)
    s1 = sum(f.sum(dim=(-3, -2, -1), keepdim=True) for f in features_list)
    mean = s1 / n
    var = sum(((f - mean) ** 2).sum(dim=(-3, -2, -1), keepdim=True) for f in features_list) / (n - 1)
    std = torch.sqrt(var + 1e-16)
    return [(f - mean) / std for f in features_list]


class PWCLiteUflow(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.search_range = 4
        self.num_chs = [3, 16, 32, 32, 32, 32]
        self.output_level = 3
        self.num_levels = 6
        self.deconv_chs = 32
        self.level_dropout = cfg.level_dropout
        self.leakyRELU = nn.LeakyReLU(0.1, inplace=True)
        self.feature_norm = cfg.feature_norm
        self.align_corners = cfg.align_corners
        self.warp_pad = cfg.warp_pad
        self.feature_pyramid_extractor = FeatureExtractor(self.num_chs, convs_per_level=3, rescale_input=True,
                                                          moments=bool(cfg.feature_norm))
        self.n_frames = cfg.n_frames
        self.reduce_dense = cfg.reduce_dense
        # opt-in: cfg.feature_storage = 'bf16' keeps the correlation / warp inputs as bf16 in HBM (fp32 arithmetic);
        # absent or 'fp32' = the reference's fp32 path
        fs = cfg.get('feature_storage', 'fp32') if hasattr(cfg, 'get') else 'fp32'
        self.feature_storage = torch.bfloat16 if fs == 'bf16' else None
        self.corr = Correlation(pad_size=self.search_range, kernel_size=1, max_displacement=self.search_range,
                                stride1=1, stride2=1, corr_multiply=1, storage_dtype=self.feature_storage)
        self.dim_corr = (self.search_range * 2 + 1) ** 2
        est = FlowEstimatorReduce if self.reduce_dense else FlowEstimatorDense
        self.flow_estimators = nn.ModuleList()
        for l, num in enumerate(self.num_chs[::-1][0:self.output_level + 1]):
            ch_in = num + (self.dim_corr + 2) * (self.n_frames - 1) + (self.deconv_chs if l > 0 else 0)
            self.flow_estimators.append(est(ch_in))
        self.context_networks = ContextNetwork(
            (self.flow_estimators[self.output_level].feat_dim + 2) * (self.n_frames - 1))
        self.deconv_networks = nn.ModuleList(
            [deconv(e.feat_dim, self.deconv_chs) for e in self.flow_estimators[0:self.output_level]])

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def init_weights(self):
        init_conv_weights(self, 'xavier')

    def _drops(self, n_passes, batch_per_pass, device):
        """Level-dropout multipliers, drawn from the CPU RNG in the reference's order (per pass:
        one per level, then one for the context net; models/pwclite_uflow.py:226-229,240-242)."""
        n = self.output_level + 2
        if not (self.training and self.level_dropout > 0):
            return None
        vals = [[float(torch.rand(1) > self.level_dropout) for _ in range(n)] for _ in range(n_passes)]
        t = torch.tensor(vals, dtype=torch.float32)                      # [passes, n]
        t = t.repeat_interleave(batch_per_pass, dim=0).t().contiguous()  # [n, passes*B]
        return t.to(device, non_blocking=True).view(n, -1, 1, 1, 1)

    def forward_2_frames(self, x1_pyramid, x2_pyramid, drops=None, moments=None):
        """models/pwclite_uflow.py:193-252.  moments: optional (x1 rows, x2 rows) lists per pyramid level, the feature maps'
        partial moments from the extractor's conv epilogues (FeatureExtractor(moments=True))."""
        flows = []
        b, _, h, w = x1_pyramid[0].shape
        flow = torch.zeros(b, 2, h, w, dtype=torch.float32, device=x1_pyramid[0].device)
        act = None
        levels = list(zip(x1_pyramid[0:self.output_level + 1], x2_pyramid[0:self.output_level + 1]))
        for l, (x1, x2) in enumerate(levels):
            if (self.feature_norm and self.feature_storage is None
                    and AF.level_supported(x1, None if l == 0 else flow, True, self.search_range)):
                # the whole level in front of the estimator as two launches: flow upsample +
                # warp + moments, then the cost volume of the normalised pair straight from the raw maps
                r1 = moments[0][l] if moments is not None else None
                r2 = moments[1][l] if moments is not None else None
                if l == 0:
                    cfg = AF.LevelCfg(['vol', 'x1n', 0], 'joint', 0.1, self.search_range)
                    est_in = AF.level(x1, x2, None, cfg, flow, x1_rows=r1, x2_rows=r2)
                else:
                    cfg = AF.LevelCfg(['vol', 'x1n', 'flow', 0], 'joint', 0.1, self.search_range, True, self.align_corners,
                                      self.warp_pad, self.align_corners)
                    est_in, flow = AF.level(x1, x2, flow, cfg, self.deconv_networks[l - 1](act), x1_rows=r1)
            else:
                if l == 0:
                    x2_warp = x2
                else:
                    flow = F.interpolate(flow * 2, scale_factor=2, mode='bilinear', align_corners=self.align_corners)
                    x2_warp = flow_warp(x2, flow, align_corners=self.align_corners, pad=self.warp_pad,
                                        **({'storage_dtype': self.feature_storage} if self.feature_storage is not None else {}))
                if self.feature_norm:
                    x1, x2_warp = normalize_features([x1, x2_warp])
                # corr + LeakyReLU(0.1) in one kernel, written straight into the estimator's concatenated input
                after = (x1, flow) if l == 0 else (x1, flow, self.deconv_networks[l - 1](act))
                est_in = self.corr.concat(x1, x2_warp, after=after, negative_slope=0.1)
            act, flow_res = self.flow_estimators[l](est_in)
            if drops is not None:
                flow_res = flow_res * drops[l]
                act = act * drops[l]
            flow = flow + flow_res
            flows.append(flow)
        flow_fine = self.context_networks(torch.cat([act, flow], dim=1))
        if drops is not None:
            flow_fine = flow_fine * drops[len(levels)]
        flow = flow + flow_fine
        flows[-1] = flow
        for _ in range(2):
            flow = F.interpolate(flow * 2, scale_factor=2, mode='bilinear', align_corners=self.align_corners)
            flows.append(flow)
        return flows[::-1]

    def forward(self, x, with_bk=False):
        n_frames = x.size(1) // 3
        if n_frames != 2:
            raise NotImplementedError
        B = x.size(0)
        imgs = [x[:, 0:3], x[:, 3:6]]
        pyr_all = self.feature_pyramid_extractor(torch.cat(imgs, 0))
        moms = self.feature_pyramid_extractor.pyramid_moments
        have_m = moms is not None and all(t is not None for t in moms)
        p1 = [p[:B] for p in pyr_all] + [imgs[0]]
        p2 = [p[B:] for p in pyr_all] + [imgs[1]]
        res = {}
        if with_bk:
            # first maps of the 2B (fw; bw) samples = the extractor's batch as it stands (no copy); second maps = its
            # two halves swapped; the same holds for the rows of partial moments
            a = list(pyr_all)  # (the image level the reference appends to its pyramids is never read, pwclite_uflow.py:203)
            b = [torch.cat([p[B:], p[:B]], 0) for p in pyr_all]
            m = ([t for t in moms], [torch.roll(t, B, 0) for t in moms]) if have_m else None
            flows = self.forward_2_frames(a, b, self._drops(2, B, x.device), m)
            res['flows_fw'] = [f[:B] for f in flows]
            res['flows_bw'] = [f[B:] for f in flows]
        else:
            m = ([t[:B] for t in moms], [t[B:] for t in moms]) if have_m else None
            res['flows_fw'] = self.forward_2_frames(p1, p2, self._drops(1, B, x.device), m)
        return res
