"""HRNetEncoder (reference: module/baseline/base_hrnet/hrnet_encoder.py:28-105)."""
from ....core import registry
from ....core.config import ConfigModule
from ._hrnet_rssformer import hrnetv2_w18, hrnetv2_w32, hrnetv2_w40, hrnetv2_w48

for _n, _f in (("hrnetv2_w18", hrnetv2_w18), ("hrnetv2_w32", hrnetv2_w32), ("hrnetv2_w40", hrnetv2_w40),
               ("hrnetv2_w48", hrnetv2_w48)):
    registry.MODEL.register(_n, _f)

default_config = dict(hrnet_type="hrnetv2_w18", pretrained=False, weight_path=None, norm_eval=False, frozen_stages=-1,
                      with_cp=False)
_WIDTHS = {"hrnetv2_w18": (18, 36, 72, 144), "hrnetv2_w32": (32, 64, 128, 256), "hrnetv2_w40": (40, 80, 160, 320),
           "hrnetv2_w48": (48, 96, 192, 384)}


@registry.MODEL.register("HRNetEncoder")
class HRNetEncoder(ConfigModule):
    def __init__(self, config=default_config):
        super().__init__(config)
        c = self.config
        self.hrnet = registry.MODEL[c.hrnet_type](pretrained=c.pretrained, weight_path=c.weight_path,
                                                  norm_eval=c.norm_eval, frozen_stages=c.frozen_stages)

    def forward(self, x):
        return self.hrnet(x)

    def set_default_config(self):
        self.config.update(default_config)

    def output_channels(self):
        try:
            return _WIDTHS[self.config.hrnet_type]
        except KeyError:
            raise NotImplementedError("{} is not implemented.".format(self.config.hrnet_type))
