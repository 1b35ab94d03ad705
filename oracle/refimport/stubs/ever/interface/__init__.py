import torch.nn as nn


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(d):
        out = AttrDict()
        for k, v in d.items():
            out[k] = AttrDict.wrap(v) if isinstance(v, dict) else v
        return out

    def update_nested(self, d):
        for k, v in d.items():
            if isinstance(v, dict) and isinstance(self.get(k), dict):
                self[k].update_nested(v)
            else:
                self[k] = AttrDict.wrap(v) if isinstance(v, dict) else v


class _Cfg(AttrDict):
    # ERModule.set_default_config calls self.config.update(dict(...)) with nested dicts
    def update(self, d=None, **kw):
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = AttrDict.wrap(v) if isinstance(v, dict) else v


class ERModule(nn.Module):
    def __init__(self, config=None, **kw):
        super().__init__()
        self.config = _Cfg()
        self.set_default_config()
        user = dict(config or {})
        AttrDict.update_nested(self.config, user)

    def set_default_config(self):
        pass
