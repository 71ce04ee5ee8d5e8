"""PWCLite host model (ARFlow) on the gfx950 ops.  Same constructor cfg keys, forward contract and
state_dict layout as models/pwclite.py:109-283; correlation and warp are the HIP autograd ops."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as AF
from ..correlation import Correlation
from ..warp_utils import flow_warp
from .blocks import (ContextNetwork, FeatureExtractor, FlowEstimatorDense, FlowEstimatorReduce, conv,
                     init_conv_weights, pair_batches)


class PWCLite(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.search_range = 4
        self.num_chs = [3, 16, 32, 64, 96, 128, 192]
        self.output_level = 4
        self.num_levels = 7
        self.leakyRELU = nn.LeakyReLU(0.1, inplace=True)
        self.feature_pyramid_extractor = FeatureExtractor(self.num_chs)
        self.upsample = cfg.upsample
        self.n_frames = cfg.n_frames
        self.reduce_dense = cfg.reduce_dense
        self.corr = Correlation(pad_size=self.search_range, kernel_size=1, max_displacement=self.search_range,
                                stride1=1, stride2=1, corr_multiply=1)
        self.dim_corr = (self.search_range * 2 + 1) ** 2
        self.num_ch_in = 32 + (self.dim_corr + 2) * (self.n_frames - 1)
        est = FlowEstimatorReduce if self.reduce_dense else FlowEstimatorDense
        self.flow_estimators = est(self.num_ch_in)
        self.context_networks = ContextNetwork((self.flow_estimators.feat_dim + 2) * (self.n_frames - 1))
        self.conv_1x1 = nn.ModuleList([conv(c, 32, kernel_size=1, stride=1, dilation=1)
                                       for c in (192, 128, 96, 64, 32)])

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def init_weights(self):
        init_conv_weights(self, 'kaiming')

    def forward_2_frames(self, x1_pyramid, x2_pyramid):
        """models/pwclite.py:161-204."""
        flows = []
        b, _, h, w = x1_pyramid[0].shape
        flow = torch.zeros(b, 2, h, w, dtype=torch.float32, device=x1_pyramid[0].device)
        for l, (x1, x2) in enumerate(zip(x1_pyramid, x2_pyramid)):
            if l == 0:
                x2_warp = x2
            elif AF.warp_up2_supported(x2, flow):
                # flow x2 upsample (models/pwclite.py:178-179) folded into the warp launch
                x2_warp, flow = AF.warp_up2(x2, flow, up_align=True)
            else:
                flow = F.interpolate(flow * 2, scale_factor=2, mode='bilinear', align_corners=True)
                x2_warp = flow_warp(x2, flow)
            x1_1by1 = self.conv_1x1[l](x1)
            # corr + LeakyReLU(0.1) in one kernel, written straight into the estimator's concatenated input
            x_intm, flow_res = self.flow_estimators(self.corr.concat(x1, x2_warp, after=(x1_1by1, flow), negative_slope=0.1))
            flow = flow + flow_res
            flow = flow + self.context_networks(torch.cat([x_intm, flow], dim=1))
            flows.append(flow)
            if l == self.output_level:
                break
        if self.upsample:
            # only the finest flow is upsampled (models/pwclite.py:200-203) -> 6 outputs
            flows.append(F.interpolate(flow * 4, scale_factor=4, mode='bilinear', align_corners=True))
        return flows[::-1]

    def forward_3_frames(self, x0_pyramid, x1_pyramid, x2_pyramid):
        """Multi-view pass of models/pwclite.py:206-258: the centre frame is matched against BOTH neighbours and
        each view's decoder also sees the other view's cost volume and (negated) flow.

        Here the two views (centre->0, centre->2) are stacked on the batch axis, like the two directions of the
        2-frame pass: one warp, one cost volume, one estimator pass and one context pass per level at batch 2B
        instead of two of each at batch B.  "The other view" of a stacked tensor is its two halves swapped.
        Returns (flows centre->0, flows centre->2), finest first."""
        B = x1_pyramid[0].shape[0]

        def other(t):  # [view a; view b] -> [view b; view a]
            return torch.cat([t[B:], t[:B]], 0)

        flows = []
        flow = None  # [2B,2,h,w]: rows 0..B-1 view (1->0), rows B..2B-1 view (1->2)
        for l, (x0, x1, x2) in enumerate(zip(x0_pyramid, x1_pyramid, x2_pyramid)):
            centre = torch.cat([x1, x1], 0)
            neighbours = torch.cat([x0, x2], 0)
            if flow is None:
                flow = torch.zeros(2 * B, 2, *x1.shape[2:], dtype=torch.float32, device=x1.device)
            elif AF.warp_up2_supported(neighbours, flow):
                neighbours, flow = AF.warp_up2(neighbours, flow, up_align=True)
            else:
                flow = F.interpolate(flow * 2, scale_factor=2, mode='bilinear', align_corners=True)
                neighbours = flow_warp(neighbours, flow)
            cost = self.corr(centre, neighbours, negative_slope=0.1)
            squeezed = self.conv_1x1[l](x1)
            feat, residual = self.flow_estimators(
                torch.cat([torch.cat([squeezed, squeezed], 0), cost, other(cost), flow, -other(flow)], 1))
            flow = flow + residual
            flow = flow + self.context_networks(torch.cat([feat, other(feat), flow, -other(flow)], 1))
            flows.append(flow)
            if l == self.output_level:
                break
        if self.upsample:
            flows = [F.interpolate(f * 4, scale_factor=4, mode='bilinear', align_corners=True) for f in flows]
        return [f[:B] for f in flows[::-1]], [f[B:] for f in flows[::-1]]

    def forward(self, x, with_bk=False):
        """models/pwclite.py:260-283 -> {'flows_fw': [...], 'flows_bw': [...]} finest first."""
        n_frames = x.size(1) // 3
        imgs = [x[:, 3 * i: 3 * i + 3] for i in range(n_frames)]
        B = x.size(0)
        # one extractor pass over all frames (batch n*B) instead of n passes
        pyr_all = self.feature_pyramid_extractor(torch.cat(imgs, 0))
        pyrs = [[p[i * B:(i + 1) * B] for p in pyr_all] + [imgs[i]] for i in range(n_frames)]
        res = {}
        if n_frames == 2:
            if with_bk:
                a, b = pair_batches(pyrs[0], pyrs[1])
                flows = self.forward_2_frames(a, b)
                res['flows_fw'] = [f[:B] for f in flows]
                res['flows_bw'] = [f[B:] for f in flows]
            else:
                res['flows_fw'] = self.forward_2_frames(pyrs[0], pyrs[1])
        elif n_frames == 3:
            flows_10, flows_12 = self.forward_3_frames(pyrs[0], pyrs[1], pyrs[2])
            res['flows_fw'], res['flows_bw'] = flows_12, flows_10
        elif n_frames == 5:
            # sliding multi-view windows (models/pwclite.py:274-281): window k is frames (k-1, k, k+1) around
            # centre k; forward flows come from the windows centred on 1 and 2, backward ones on 2 and 3
            def window(k):
                return self.forward_3_frames(pyrs[k - 1], pyrs[k], pyrs[k + 1])  # (k -> k-1, k -> k+1)
            w1, w2 = window(1), window(2)
            res['flows_fw'] = [w1[1], w2[1]]
            if with_bk:
                res['flows_bw'] = [w2[0], window(3)[0]]
        else:
            raise NotImplementedError
        return res
