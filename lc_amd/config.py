"""Tiny attribute-dict config (stand-in for `mmcv.Config`, which the reference uses and this image lacks):
attribute access + `.get`, nested dicts wrapped on the fly, YAML loader for the reference's `configs/*.yaml`."""
from __future__ import annotations


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        v = super().get(k, default)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


def load_yaml(path: str) -> AttrDict:
    import yaml

    with open(path) as f:
        return AttrDict(yaml.safe_load(f))
