"""Run the product's HOST model classes (arflow_amd/models: conv pyramids, decoders, wiring) with the
ORACLE ops substituted for the HIP ops, on the CPU.  Used by tests (host-logic parity against the
reference's golden flows) and by bench.py's ``cpu_baseline`` leg.  TEST INFRASTRUCTURE ONLY: the
substitution is made from here, the product never selects it."""
import contextlib

import torch
import torch.nn as nn

from . import ops as O


class OracleCorrelation(nn.Module):
    def __init__(self, d=4):
        super().__init__()
        self.d = d

    def forward(self, x1, x2, negative_slope=1.0):
        out = O.correlation(x1, x2, self.d)
        return out if negative_slope == 1.0 else torch.nn.functional.leaky_relu(out, negative_slope)

    def concat(self, x1, x2, before=(), after=(), negative_slope=1.0):
        return torch.cat(list(before) + [self.forward(x1, x2, negative_slope)] + list(after), 1)


@contextlib.contextmanager
def oracle_ops(model):
    import arflow_amd.models.pwclite as mp
    import arflow_amd.models.pwclite_uflow as mpu
    import arflow_amd.models.uflow_model as mum
    import arflow_amd.models.blocks as mb
    import arflow_amd.functional as AF
    saved = [(AF, 'level_supported', AF.level_supported), (AF, 'warp_up2_supported', AF.warp_up2_supported),
             (mp, 'flow_warp', mp.flow_warp), (mpu, 'flow_warp', mpu.flow_warp),
             (mum, 'cost_volume_concat', mum.cost_volume_concat),
             (mum.uflow_utils, 'resample_flow', mum.uflow_utils.resample_flow),
             (mpu, 'normalize_features', mpu.normalize_features), (mum, 'normalize_features', mum.normalize_features),
             (mb, 'bias_act', mb.bias_act)]
    old_corr = getattr(model, 'corr', None)
    try:
        # the fused level / upsample-fused warp launches are HIP ops too: off, so that a twin on the GPU runs the op-by-op
        # wiring with the oracle ops below (on the CPU they are off anyway)
        AF.level_supported = lambda *a, **k: False
        AF.warp_up2_supported = lambda *a, **k: False
        mp.flow_warp = O.flow_warp
        mpu.flow_warp = O.flow_warp
        mum.cost_volume_concat = lambda a, b, before, after, max_displacement, negative_slope=1.0: OracleCorrelation(
            max_displacement).concat(a, b, before, after, negative_slope)
        mum.uflow_utils.resample_flow = lambda src, flow: O.resample(src, O.flow_to_warp(flow))
        mpu.normalize_features = O.normalize_features_joint
        mb.bias_act = lambda y, b, s: torch.nn.functional.leaky_relu(y + b.view(1, -1, 1, 1), s)
        mum.normalize_features = O.normalize_features_uflow
        if old_corr is not None:
            model.corr = OracleCorrelation(4)
        yield model
    finally:
        for mod, name, val in saved:
            setattr(mod, name, val)
        if old_corr is not None:
            model.corr = old_corr
