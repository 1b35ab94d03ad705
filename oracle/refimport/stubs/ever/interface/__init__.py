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


# ---- only for the eval/TTA golden case: `er.MultiTransform` and `ever.interface.transform_base.Transform` are used by
# the reference's module/tta.py:12-24,52-136.  ever is un-vendored: semantics restated from the call sites in tta():
# transform(image) -> one image per transform; inv_transform(outs) -> each output mapped back by ITS transform.
class Transform:
    def transform(self, inputs):
        raise NotImplementedError

    def inv_transform(self, transformed_inputs):
        raise NotImplementedError


class MultiTransform:
    def __init__(self, *transforms):
        self.transforms = list(transforms)

    def transform(self, inputs):
        return [t.transform(inputs) for t in self.transforms]

    def inv_transform(self, outs):
        return [t.inv_transform(o) for t, o in zip(self.transforms, outs)]
