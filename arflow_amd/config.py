"""Attribute-access config dict.

The reference turns its JSON configs into ``easydict.EasyDict`` (train.py:20-21) and reads them
with attribute access plus ``'key' in cfg`` (losses/flow_loss.py:30).  easydict is not a
dependency here; this is the ~20-line equivalent.
"""
import json


class AttrDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        if isinstance(value, dict) and not isinstance(value, AttrDict):
            value = AttrDict(value)
        elif isinstance(value, list):
            value = [AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v
                     for v in value]
        super().__setitem__(key, value)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def load_json(path):
    with open(path) as f:
        return AttrDict(json.load(f))
