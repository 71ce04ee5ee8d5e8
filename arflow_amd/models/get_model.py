"""models/get_model.py:8-25 for the deterministic flow models (the probabilistic research variants
are out of scope)."""
from .pwclite import PWCLite
from .pwclite_uflow import PWCLiteUflow
from .uflow_model import PWCFlow


def get_model(cfg):
    if cfg.type == 'pwclite':
        return PWCLite(cfg)
    if cfg.type == 'pwclite_uflow':
        return PWCLiteUflow(cfg)
    if cfg.type == 'uflow':
        return PWCFlow(cfg)
    raise NotImplementedError(cfg.type)
