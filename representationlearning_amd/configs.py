"""Model configurations (dict schema of the reference's configs/baseline/hrnetw32.py:5-34)."""


def rssformer_config(variant="base", classes=6, pretrained=False):
    ht, neck = {"tiny": ("hrnetv2_w18", 270), "base": ("hrnetv2_w32", 480), "large": ("hrnetv2_w48", 720)}[variant]
    return dict(backbone=dict(hrnet_type=ht, pretrained=pretrained, norm_eval=False, frozen_stages=-1, with_cp=False,
                              with_gc=False),
                neck=dict(in_channels=neck), classes=classes, head=dict(in_channels=neck, upsample_scale=4.0),
                loss=dict(ignore_index=-1, ce=dict()))


def synthetic_batch(B, S, classes=6, seed=2333, device="cuda"):
    """SURVEY §8d inputs: N(0,1) tiles, labels randint(-1, classes) constant over 16x16 blocks (~14 % ignore)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(B, 3, S, S, generator=g)
    lab = torch.randint(-1, classes, (B, (S + 15) // 16, (S + 15) // 16), generator=g)
    lab = lab.repeat_interleave(16, 1).repeat_interleave(16, 2)[:, :S, :S].contiguous()
    return img.to(device), lab.to(device)
