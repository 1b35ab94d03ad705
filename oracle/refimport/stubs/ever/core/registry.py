class _Registry(dict):
    def register(self, name, obj=None):
        if obj is not None:
            self[name] = obj
            return obj

        def deco(o):
            self[name] = o
            return o
        return deco


MODEL = _Registry()
DATALOADER = _Registry()
OPT = _Registry()
LR = _Registry()
