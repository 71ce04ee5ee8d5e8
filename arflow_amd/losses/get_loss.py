"""losses/get_loss.py:9-24 for the loss types on the hot path (ELBO / MSE research losses are out
of scope)."""
from .flow_loss import unFlowLoss
from .fullres_loss import FullResLoss
from .mv_loss import MvLoss
from .uflow_loss import UFlowLoss


def get_loss(cfg):
    if cfg.type == 'unflow':
        return unFlowLoss(cfg)
    if cfg.type == 'fullres':
        return FullResLoss(cfg)
    if cfg.type == 'uflow':
        return UFlowLoss(cfg)
    if cfg.type == 'mv':  # build-defined 3-frame objective; not a reference type
        return MvLoss(cfg)
    raise NotImplementedError(cfg.type)
