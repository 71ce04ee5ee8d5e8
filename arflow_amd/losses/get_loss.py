"""losses/get_loss.py:9-24 for the loss types on the hot path (ELBO / MSE research losses are out
of scope)."""
from .flow_loss import unFlowLoss
from .fullres_loss import FullResLoss
from .uflow_loss import UFlowLoss


def get_loss(cfg):
    if cfg.type == 'unflow':
        return unFlowLoss(cfg)
    if cfg.type == 'fullres':
        return FullResLoss(cfg)
    if cfg.type == 'uflow':
        return UFlowLoss(cfg)
    raise NotImplementedError(cfg.type)
