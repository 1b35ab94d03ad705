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


def config_by_name(name):
    """Dotted config names of the reference (configs/baseline/hrnetw32.py + configs/base/loveda.py): model params +
    the optimizer / LR / train sections the trainer reads.  `classes` = 7 as in the reference's LoveDA config
    (configs/baseline/hrnetw32.py:19: LoveDA masks reach 6 after the `mask - 1` shift), so a reference checkpoint loads;
    BASELINE's synthetic 6-class tiles (bench.py) use rssformer_config() directly — or override `model.params.classes 6`."""
    variants = {"baseline.hrnetw32": "base", "baseline.hrnetw18": "tiny", "baseline.hrnetw48": "large"}
    if name not in variants:
        raise KeyError("unknown config '%s' (built: %s)" % (name, ", ".join(sorted(variants))))
    return dict(
        model=dict(type="RSSFormer", params=rssformer_config(variants[name], classes=7)),
        optimizer=dict(type="sgd", params=dict(momentum=0.9, weight_decay=0.0001), grad_clip=dict(max_norm=35, norm_type=2)),
        learning_rate=dict(type="poly", params=dict(base_lr=0.01, power=0.9, max_iters=30000)),
        train=dict(forward_times=1, num_iters=30000, eval_per_epoch=True, summary_grads=False, summary_weights=False,
                   distributed=True, apex_sync_bn=True, sync_bn=True, eval_after_train=True, log_interval_step=50,
                   save_ckpt_interval_epoch=1000, eval_interval_epoch=20),
        # data sections of configs/base/loveda.py:6-66 (paths and loader parameters; the transforms run on the GPU: data/loveda.py)
        data=dict(
            train=dict(type="LoveDALoader", params=dict(
                image_dir=["./LoveDA/Train/Urban/images_png/", "./LoveDA/Train/Rural/images_png/"],
                mask_dir=["./LoveDA/Train/Urban/masks_png/", "./LoveDA/Train/Rural/masks_png/"],
                crop=512, p_oneof=0.75, shift_scale_rotate=dict(shift_limit=0.0625, scale_limit=0.2, rotate_limit=45, p=0.2),
                CV=dict(k=10, i=-1), training=True, batch_size=8, num_workers=2)),
            test=dict(type="LoveDALoader", params=dict(
                image_dir=["./LoveDA/Val/Urban/images_png/", "./LoveDA/Val/Rural/images_png/"],
                mask_dir=["./LoveDA/Val/Urban/masks_png/", "./LoveDA/Val/Rural/masks_png/"],
                CV=dict(k=10, i=-1), training=False, batch_size=4, num_workers=0))),
        test=dict())
