#!/usr/bin/env python3
"""train.py — same surface as the reference's RSSFormer-TIP2023/train.py (:64-80): `seed_torch`, the evaluate callback
and `python train.py --config_path=baseline.hrnetw32 --model_dir=... [key value ...]` launched one process per GPU
(`python -m torch.distributed.run --nproc-per-node N train.py ...`).  The external `ever` trainer is replaced by
representationlearning_amd.trainer (RCCL data parallel, bf16, fused clip+SGD).  No dataset ships with this build:
without --data_dir it trains on the synthetic LoveDA-shaped tiles of SURVEY.md §8d."""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def seed_torch(seed=2333):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def evaluate_cls_fn(model, batches, classes, logger=print, tta_scales=None, viz_op=None):
    """train.py:14-57 / eval.py:55-77 of the reference: (optional test-time augmentation,) argmax, drop ignore(-1),
    confusion-matrix metrics; with a `viz_op` (module/viz.py) the label and prediction maps are also written as palette
    PNGs, `gt<name>.png` / `<name>.png` (eval.py:73-77; names: batch index and position, the synthetic tiles have no file)."""
    from representationlearning_amd import ops
    from representationlearning_amd.metric import PixelMetric
    from representationlearning_amd.module.tta import tta, Scale
    metric = PixelMetric(classes)
    model.eval()
    with torch.no_grad():
        for bi, (img, lab) in enumerate(batches):
            scores = model(img) if tta_scales is None else tta(model, img, [Scale(scale_factor=s) for s in tta_scales])
            metric.forward_scores(lab, scores)          # argmax + ignore(-1) mask + confusion matrix: one HIP kernel
            if viz_op is not None:
                pred = ops.argmax_confusion(scores, want_pred=True).cpu().numpy()
                for i in range(pred.shape[0]):
                    viz_op(lab[i].cpu().numpy(), "gt%04d_%02d.png" % (bi, i))
                    viz_op(pred[i], "%04d_%02d.png" % (bi, i))
    out = metric.summary_all()
    logger("mIoU %.4f  OA %.4f" % (out["miou"], out["overall_accuracy"]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", default="baseline.hrnetw32")
    ap.add_argument("--model_dir", default="./log/rssformer")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (reference: 8, configs/base/loveda.py:39)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("overrides", nargs="*", help="`a.b.c value` pairs (scripts/train.sh:12-14)")
    args = ap.parse_args()

    from representationlearning_amd import _lib, nnf
    from representationlearning_amd.configs import config_by_name, synthetic_batch
    from representationlearning_amd.core import registry
    from representationlearning_amd.core.config import AttrDict, apply_overrides
    from representationlearning_amd.trainer import Trainer, init_distributed, shutdown_distributed
    _lib.load()
    rank, local, world = init_distributed()
    seed_torch(2333)
    registry.register_all()
    cfg = apply_overrides(AttrDict.wrap(config_by_name(args.config_path)), args.overrides)
    model = registry.MODEL[cfg.model.type](cfg.model.params).cuda()
    tr = cfg.train
    trainer = Trainer(model, base_lr=cfg.learning_rate.params.base_lr, momentum=cfg.optimizer.params.momentum,
                      weight_decay=cfg.optimizer.params.weight_decay, max_norm=cfg.optimizer.grad_clip.max_norm,
                      power=cfg.learning_rate.params.power, max_iters=cfg.learning_rate.params.max_iters, bf16=not args.fp32,
                      sync_bn=tr.sync_bn)
    img, lab = synthetic_batch(args.batch, args.size, classes=cfg.model.params.classes, seed=2333 + rank)
    for it in range(args.iters):
        loss = trainer.step(img, dict(cls=lab))
        if rank == 0 and (it % tr.log_interval_step == 0 or it == args.iters - 1):
            print("iter %d  fc_loss %.5f" % (it, float(loss)), flush=True)
    if rank == 0:
        os.makedirs(args.model_dir, exist_ok=True)
        from representationlearning_amd.trainer import flush_bn_counters
        flush_bn_counters(trainer)
        path = os.path.join(args.model_dir, "model-%d.pth" % trainer.it)
        torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)   # reference-compatible keys
        print("saved", path)
        evaluate_cls_fn(model, [(img, lab)], cfg.model.params.classes)
    shutdown_distributed(trainer)


if __name__ == "__main__":
    main()
