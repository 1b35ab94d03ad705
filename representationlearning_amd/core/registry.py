"""Name -> constructor registries (the slice of `ever.core.registry` the RSSFormer path uses:
train.py:11, hrnet_aux.py:70, hrnet_encoder.py:14-17,28 of the reference)."""


class Registry(dict):
    def register(self, name, obj=None):
        """Both call forms of the reference: @MODEL.register('x') and MODEL.register('x', fn)."""
        if obj is not None:
            self[name] = obj
            return obj

        def deco(o):
            self[name] = o
            return o
        return deco


MODEL = Registry()
DATALOADER = Registry()


def register_all():
    """Import every module that registers something (reference: er.registry.register_all, train.py:11)."""
    from ..module.baseline import hrnet_aux  # noqa: F401
    from ..module.baseline.base_hrnet import hrnet_encoder  # noqa: F401
