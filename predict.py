#!/usr/bin/env python3
"""predict.py — surface of the reference's RSSFormer-TIP2023/predict.py (:14-48): `predict_test(ckpt_path, config_path, save_dir)`
loads a checkpoint (DDP `module.` prefix stripped), runs the model over the test tiles and writes one class-id PNG per tile, named
like the tile.  The x4 bilinear head and the argmax are one HIP launch (`HRNetFusion.predict`, rssf_head_upsample_softmax): the
full-resolution probabilities the reference materialises only to take their argmax never reach HBM.
Tiles: the folders of the config's `data.test.params.image_dir` (LoveDA layout, reference configs/base/loveda.py:47-53) or
--image_dir; without either, the synthetic LoveDA-shaped tiles of SURVEY.md §8d (named 0000.png, 0001.png, ...)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def predict_test(ckpt_path, config_path="baseline.hrnetw32", save_dir="", image_dir=None, overrides=(), synthetic_tiles=8):
    from PIL import Image
    from eval import remove_module_prefix
    from train import seed_torch
    from representationlearning_amd import _lib
    from representationlearning_amd.configs import config_by_name, synthetic_batch
    from representationlearning_amd.core import registry
    from representationlearning_amd.core.config import AttrDict, apply_overrides
    from representationlearning_amd.data.loveda import DeviceLoader, LoveDA
    _lib.load()
    seed_torch(2333)
    registry.register_all()
    os.makedirs(save_dir, exist_ok=True)
    cfg = apply_overrides(AttrDict.wrap(config_by_name(config_path)), list(overrides))
    model = registry.MODEL[cfg.model.type](cfg.model.params)
    if ckpt_path:
        model.load_state_dict(remove_module_prefix(torch.load(ckpt_path, map_location=lambda storage, loc: storage)))
        print("Load model!")
    model = model.cuda().eval()
    test = cfg.data.test.params
    dirs = image_dir if image_dir is not None else test.image_dir
    dirs = [dirs] if isinstance(dirs, str) else list(dirs)
    if all(os.path.isdir(d) for d in dirs):
        loader = DeviceLoader(LoveDA(dirs, None), batch_size=test.batch_size)
    else:       # no dataset on this machine: synthetic tiles, so that the path can still be exercised end to end
        bs = test.batch_size
        loader = []
        for k in range(0, synthetic_tiles, bs):
            img, _ = synthetic_batch(min(bs, synthetic_tiles - k), 512, classes=cfg.model.params.classes, seed=2333 + k)
            loader.append((img, dict(fname=["%04d.png" % (k + i) for i in range(img.shape[0])])))
    written = []
    with torch.no_grad():
        for img, gt in loader:
            pred = model.predict(img).cpu().numpy()
            for clsmap, imname in zip(pred, gt["fname"]):
                path = os.path.join(save_dir, imname)
                Image.fromarray(clsmap.astype(np.uint8)).save(path)          # reference: skimage.io.imsave of the uint8 class map
                written.append(path)
    torch.cuda.empty_cache()
    return written


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Eval methods")
    parser.add_argument("--ckpt_path", type=str, help="ckpt path", default="./log/hrnetw32.pth")
    parser.add_argument("--config_path", type=str, help="config path", default="baseline.hrnetw32")
    parser.add_argument("--out_dir", type=str, help="out dir", default="./out")
    parser.add_argument("--image_dir", type=str, nargs="*", default=None, help="tile folders (default: the config's data.test)")
    parser.add_argument("overrides", nargs="*", help="`a.b.c value` config overrides, as train.py takes them")
    args = parser.parse_args()
    ckpt = args.ckpt_path if os.path.exists(args.ckpt_path) else None
    if ckpt is None:
        print("checkpoint %s not found: predicting with random-init weights" % args.ckpt_path)
    files = predict_test(ckpt, args.config_path, args.out_dir, args.image_dir, args.overrides)
    print("wrote %d class maps to %s" % (len(files), args.out_dir))
