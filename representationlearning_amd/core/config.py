"""Attribute-dict configs and the config-carrying nn.Module base used by the model classes
(stands in for `ever.ERModule`: user dict merged over `set_default_config`, hrnet_aux.py:112-134)."""
import torch.nn as nn


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @classmethod
    def wrap(cls, d):
        out = cls()
        for k, v in d.items():
            out[k] = cls.wrap(v) if isinstance(v, dict) else v
        return out

    def merge(self, d):
        """Recursive update (nested dicts are merged, leaves replaced)."""
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                if not isinstance(self[k], AttrDict):
                    self[k] = AttrDict.wrap(self[k])
                self[k].merge(v)
            else:
                self[k] = AttrDict.wrap(v) if isinstance(v, dict) else v
        return self

    def update(self, d=None, **kw):           # set_default_config() calls config.update(dict(...))
        return self.merge(dict(d or {}, **kw))


class ConfigModule(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.config = AttrDict()
        self.set_default_config()
        self.config.merge(dict(config or {}))

    def set_default_config(self):
        pass


def apply_overrides(cfg, pairs):
    """CLI `a.b.c value` overrides (scripts/train.sh:12-14 of the reference)."""
    import ast
    for key, val in zip(pairs[0::2], pairs[1::2]):
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        try:
            val = ast.literal_eval(val)
        except (ValueError, SyntaxError):
            pass
        node[parts[-1]] = val
    return cfg
