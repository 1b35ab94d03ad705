#!/usr/bin/env python3
"""eval.py — surface of the reference's RSSFormer-TIP2023/eval.py (:17-24, :32-81): `evaluate(ckpt_path, config_path,
use_tta)`: load a checkpoint (DDP `module.` prefix stripped, :37-38), softmax -> argmax -> ignore(-1) mask -> confusion
matrix -> mIoU, optionally with the six-scale test-time augmentation (module/tta.py).  Tiles: the LoveDA validation folders named by the config
(data/loveda.py reader, normalised on the GPU) when they exist, else the synthetic validation tiles."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def remove_module_prefix(state):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}


def evaluate(ckpt_path, config_path="baseline.hrnetw32", use_tta=False, batches=None, vis_dir=None, overrides=()):
    from representationlearning_amd import _lib
    from representationlearning_amd.configs import config_by_name, synthetic_batch
    from representationlearning_amd.core import registry
    from representationlearning_amd.core.config import AttrDict, apply_overrides
    from train import evaluate_cls_fn
    _lib.load()
    registry.register_all()
    cfg = apply_overrides(AttrDict.wrap(config_by_name(config_path)), list(overrides))       # e.g. model.params.classes 6
    model = registry.MODEL[cfg.model.type](cfg.model.params)
    if ckpt_path:
        model.load_state_dict(remove_module_prefix(torch.load(ckpt_path, map_location="cpu")))
    model = model.cuda().eval()
    if batches is None:
        test = cfg.data.test.params
        dirs = [test.image_dir] if isinstance(test.image_dir, str) else list(test.image_dir)
        if all(os.path.isdir(d) for d in dirs):         # the LoveDA validation folders of configs/base/loveda.py:47-56
            from representationlearning_amd.data.loveda import DeviceLoader, LoveDA
            batches = ((img, gt["cls"]) for img, gt in DeviceLoader(LoveDA(dirs, test.mask_dir), batch_size=test.batch_size))
        else:
            batches = [synthetic_batch(4, 512, classes=cfg.model.params.classes, seed=7)]
    # eval.py:57-64 of the reference: six bilinear scales when --tta
    scales = (0.5, 0.75, 1.0, 1.25, 1.5, 1.75) if use_tta else None
    # eval.py:48-50: palette PNGs next to the checkpoint (vis-<ckpt name>); without a checkpoint only when vis_dir is given
    if vis_dir is None and ckpt_path:
        vis_dir = os.path.join(os.path.dirname(ckpt_path), "vis-{}".format(os.path.basename(ckpt_path)))
    viz_op = None
    if vis_dir:
        from representationlearning_amd.module.viz import LOVEDA_PALETTE, VisualizeSegmm
        viz_op = VisualizeSegmm(vis_dir, LOVEDA_PALETTE)
    return evaluate_cls_fn(model, batches, cfg.model.params.classes, tta_scales=scales, viz_op=viz_op)


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Eval methods")
    ap.add_argument("--ckpt_path", type=str, default=None)
    ap.add_argument("--config_path", type=str, default="baseline.hrnetw32")
    ap.add_argument("--tta", type=bool, default=False)
    ap.add_argument("--vis_dir", type=str, default=None, help="where the palette PNGs go (default: vis-<ckpt> next to the checkpoint)")
    ap.add_argument("overrides", nargs="*", help="`a.b.c value` config overrides, as train.py takes them")
    a = ap.parse_args()
    evaluate(a.ckpt_path, a.config_path, a.tta, vis_dir=a.vis_dir, overrides=a.overrides)
