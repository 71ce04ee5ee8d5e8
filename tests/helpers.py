"""Test helpers: run the PRODUCT host models with the ORACLE ops patched in (CPU), so that the host
logic -- pyramid wiring, state_dict layout, fw/bw batching -- is checked against the reference's
golden flows without a GPU.  The patching happens here, in tests/, never in arflow_amd/."""
import torch
import torch.nn as nn

from oracle import ops as O


from oracle.host_models import OracleCorrelation, oracle_ops  # noqa: F401,E402


def model_cases():
    """(tag, model-class name, cfg, frames, with_bk) exactly as oracle/make_golden.py::gen_models."""
    from arflow_amd.config import AttrDict as C
    return [
        ('pwclite2', 'PWCLite', C(upsample=True, n_frames=2, reduce_dense=True), 2, True),
        ('pwclite2_dense', 'PWCLite', C(upsample=True, n_frames=2, reduce_dense=False), 2, False),
        ('pwclite3', 'PWCLite', C(upsample=True, n_frames=3, reduce_dense=True), 3, True),
        ('pwclite_uflow_1', 'PWCLiteUflow', C(level_dropout=0.0, feature_norm=True, align_corners=True,
                                             warp_pad='zeros', n_frames=2, reduce_dense=False), 2, True),
        ('pwclite_uflow_0', 'PWCLiteUflow', C(level_dropout=0.0, feature_norm=False, align_corners=False,
                                             warp_pad='border', n_frames=2, reduce_dense=False), 2, True),
        ('pwcflow', 'PWCFlow', C(level_dropout=0.0, feature_norm=True), 2, True),
    ]


def epe(a, b):
    return float(torch.sqrt(((a.double().cpu() - b.double().cpu()) ** 2).sum(1)).mean())
