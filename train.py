#!/usr/bin/env python3
"""train.py — same surface as the reference's RSSFormer-TIP2023/train.py (:64-80): `seed_torch`, the evaluate callback
and `python train.py --config_path=baseline.hrnetw32 --model_dir=... [key value ...]` launched one process per GPU
(`python -m torch.distributed.run --nproc-per-node N train.py ...`).  The external `ever` trainer is replaced by
representationlearning_amd.trainer (RCCL data parallel, bf16, fused clip+SGD).  The loop iterates `LoveDALoader`
(data/loveda.py:97-117 semantics: one permutation per epoch shared by the ranks, every rank its 1/world share, drop_last) over
HBM-resident uint8 tiles augmented on the GPU, evaluates every `train.eval_interval_epoch` epochs, saves every
`train.save_ckpt_interval_epoch` epochs and at the end, and resumes from the newest checkpoint in --model_dir
(configs/base/loveda.py:102-111).  No dataset ships with this build: when the config's LoveDA folders do not exist the same
loaders run over procedurally drawn uint8 tiles."""
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


def build_loaders(cfg, args, rank, world, dtype):
    """(train loader, evaluation loader factory, description).  `--data_dir D` re-roots the config's relative LoveDA folders
    (configs/base/loveda.py:6-66: ./LoveDA/Train/{Urban,Rural}/images_png ...) under D; when the training folders do not exist - no
    dataset ships with this build - the same loaders run over procedurally drawn uint8 tiles (data/loveda.py::SyntheticTiles)."""
    from representationlearning_amd.data.loveda import DeviceLoader, LoveDA, LoveDALoader, SyntheticTiles
    tp, ep = cfg.data.train.params, cfg.data.test.params
    root = args.data_dir

    def rooted(dirs):
        dirs = dirs if isinstance(dirs, (list, tuple)) else [dirs]
        return [os.path.join(root, d[2:] if d.startswith("./") else d) if root else d for d in dirs]

    timg, tmask = rooted(tp.image_dir), rooted(tp.mask_dir)
    classes = cfg.model.params.classes
    if all(os.path.isdir(d) for d in timg):
        train_ds, what = LoveDA(timg, tmask), "LoveDA tiles under %s" % ", ".join(timg)
        eimg, emask = rooted(ep.image_dir), rooted(ep.mask_dir)
        eval_ds = LoveDA(eimg, emask) if all(os.path.isdir(d) for d in eimg) else None
    else:
        if root:
            raise SystemExit("train.py: --data_dir %s holds none of %s" % (root, ", ".join(tp.image_dir)))
        n = max(args.synthetic_tiles, args.batch * world)
        train_ds = SyntheticTiles(n, 2 * args.size, classes=classes - 1 if classes == 7 else classes, seed=2333)
        eval_ds = SyntheticTiles(max(4, ep.batch_size), args.size, classes=classes - 1 if classes == 7 else classes, seed=4666)
        what = "%d synthetic %dx%d uint8 tiles (no LoveDA folders at %s)" % (n, 2 * args.size, 2 * args.size, ", ".join(timg))
    loader = LoveDALoader(train_ds, batch_size=args.batch, rank=rank, world=world, crop=args.size, p_oneof=tp.get("p_oneof", 0.75),
                          shift_scale_rotate=tp.get("shift_scale_rotate"), dtype=dtype, seed=2333)

    def eval_batches():
        if eval_ds is None:
            return []
        return ((img, tgt["cls"]) for img, tgt in DeviceLoader(eval_ds, batch_size=ep.batch_size, dtype=dtype))
    return loader, eval_batches, what


def latest_checkpoint(model_dir):
    """(step, model path, trainer-state path | None) of the newest `model-<step>.pth` in model_dir, or None."""
    c = checkpoints(model_dir)
    return c[0] if c else None


def checkpoints(model_dir):
    """Every `model-<step>.pth` of model_dir, newest first, as (step, model path, trainer-state path | None); None when there is none."""
    import glob
    import re
    found = []
    for f in glob.glob(os.path.join(model_dir, "model-*.pth")):
        m = re.search(r"model-(\d+)\.pth$", f)
        if m:
            found.append((int(m.group(1)), f))
    # newest first (the caller falls back to the next one when a file does not load: `resume()` below)
    return [(step, f, os.path.join(model_dir, "trainer-%d.pth" % step) if os.path.exists(os.path.join(model_dir, "trainer-%d.pth" % step)) else None)
            for step, f in sorted(found, reverse=True)] or None


def save_checkpoint(model_dir, model, trainer):
    """`model-<step>.pth` = the model's state_dict with the reference's keys (what eval.py / predict.py load, eval.py:37-38);
    `trainer-<step>.pth` = momentum + step counter, so that a resumed run continues the poly schedule and the momentum."""
    from representationlearning_amd.trainer import flush_bn_counters
    os.makedirs(model_dir, exist_ok=True)
    flush_bn_counters(trainer)
    path = os.path.join(model_dir, "model-%d.pth" % trainer.it)

    def atomic_save(obj, dst):       # a reader (the auto-resume of the next start) sees the old file or the complete new one
        tmp = dst + ".tmp"
        torch.save(obj, tmp)
        os.replace(tmp, dst)
    # trainer state first, model last: the model file's appearance is what makes the step resumable
    atomic_save(trainer.state_dict(), os.path.join(model_dir, "trainer-%d.pth" % trainer.it))
    atomic_save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", default="baseline.hrnetw32")
    ap.add_argument("--model_dir", default="./log/rssformer")
    ap.add_argument("--data_dir", default=None, help="root of the LoveDA folders the config names (./LoveDA/Train/...); "
                                                     "default: the config's paths as they are, synthetic tiles if they do not exist")
    ap.add_argument("--iters", type=int, default=None, help="stop after this many iterations (default: train.num_iters of the config)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: data.train.params.batch_size = 8)")
    ap.add_argument("--size", type=int, default=None, help="crop size (default: data.train.params.crop = 512)")
    ap.add_argument("--synthetic_tiles", type=int, default=32, help="number of synthetic tiles when no dataset is on disk")
    ap.add_argument("--no_resume", action="store_true", help="ignore checkpoints already in --model_dir")
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("overrides", nargs="*", help="`a.b.c value` pairs (scripts/train.sh:12-14)")
    args = ap.parse_args()

    import time
    from representationlearning_amd import _lib
    from representationlearning_amd.configs import config_by_name
    from representationlearning_amd.core import registry
    from representationlearning_amd.core.config import AttrDict, apply_overrides
    from representationlearning_amd.trainer import Trainer, init_distributed, shutdown_distributed
    _lib.load()
    rank, local, world = init_distributed()
    seed_torch(2333)
    registry.register_all()
    cfg = apply_overrides(AttrDict.wrap(config_by_name(args.config_path)), args.overrides)
    tr = cfg.train
    args.batch = args.batch or cfg.data.train.params.batch_size
    args.size = args.size or cfg.data.train.params.get("crop", 512)
    num_iters = args.iters if args.iters is not None else tr.num_iters
    log = print if rank == 0 else (lambda *a, **k: None)
    model = registry.MODEL[cfg.model.type](cfg.model.params).cuda()
    # The resume point is RANK 0's choice, broadcast before anybody loads: ranks that skipped different files would start with
    # different iteration counters - different learning rates, loop lengths that do not match, collectives that hang.
    # Rank 0 takes the newest checkpoint whose model AND trainer state load (a run killed in the middle of a save that predates the
    # atomic rename, a full disk: skipped with a warning); a rank that then cannot load the agreed files stops the run.
    resume, resume_sd, resume_tr = None, None, None
    if rank == 0:
        for cand in ([] if args.no_resume else (checkpoints(args.model_dir) or [])):
            try:
                sd = torch.load(cand[1], map_location="cpu")
                sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
                missing = set(model.state_dict().keys()) ^ set(sd.keys())
                if missing:
                    raise KeyError("state_dict keys differ (%d), e.g. %s" % (len(missing), sorted(missing)[:2]))
                tsd = torch.load(cand[2], map_location="cpu") if cand[2] is not None else None
            except Exception as e:       # noqa: BLE001 - any unreadable / unusable file is skipped
                log("skipping unusable checkpoint %s (%s: %s)" % (cand[1], type(e).__name__, str(e)[:200]))
                continue
            resume, resume_sd, resume_tr = cand, sd, tsd
            break
    if world > 1:
        import torch.distributed as dist
        box = [resume]
        dist.broadcast_object_list(box, src=0)
        resume = box[0]
        if resume is not None and rank != 0:
            resume_sd = {(k[7:] if k.startswith("module.") else k): v for k, v in torch.load(resume[1], map_location="cpu").items()}
            resume_tr = torch.load(resume[2], map_location="cpu") if resume[2] is not None else None
    if resume is not None:
        model.load_state_dict(resume_sd)      # the model BEFORE the trainer re-seats its parameters into the flat buffer
    trainer = Trainer(model, base_lr=cfg.learning_rate.params.base_lr, momentum=cfg.optimizer.params.momentum,
                      weight_decay=cfg.optimizer.params.weight_decay, max_norm=cfg.optimizer.grad_clip.max_norm,
                      power=cfg.learning_rate.params.power, max_iters=cfg.learning_rate.params.max_iters, bf16=not args.fp32,
                      sync_bn=tr.sync_bn)
    if resume is not None:
        if resume_tr is not None:
            trainer.load_state_dict(resume_tr)
        else:
            trainer.it = resume[0]
        log("resumed from %s (iteration %d)" % (resume[1], trainer.it))
    loader, eval_batches, what = build_loaders(cfg, args, rank, world, torch.float32)
    steps_per_epoch = len(loader)
    log("data: %s; %d tiles, %d iterations per epoch on each of %d rank(s), batch %d x %dx%d"
        % (what, len(loader.ds), steps_per_epoch, world, args.batch, args.size, args.size))
    classes = cfg.model.params.classes

    def evaluate():
        if rank == 0:
            evaluate_cls_fn(model, eval_batches(), classes, logger=log)

    loader.epoch = trainer.it // steps_per_epoch
    t_load = t_step = 0.0
    n_timed = 0
    done = trainer.it >= num_iters
    while not done:
        epoch = loader.epoch
        t0 = time.perf_counter()
        for img, target in loader:
            t1 = time.perf_counter()
            loss = trainer.step(img, target)
            it = trainer.it                       # iterations finished
            if it % tr.log_interval_step == 0 or it == num_iters or it == 1:
                lv = float(loss)                  # the only host sync of the loop
                trainer.check_exchange()          # a peer-to-peer SyncBN exchange that hit its (debug) spin bound stops the run here
                t2 = time.perf_counter()
                t_load, t_step, n_timed = t_load + (t1 - t0), t_step + (t2 - t1), n_timed + 1
                log("iter %d  epoch %d  fc_loss %.5f  lr %.6f" % (it, epoch, lv, float(trainer.lr_dev)), flush=True)
            if it >= num_iters:
                done = True
                break
            t0 = time.perf_counter()
        if done:
            break
        ep_done = loader.epoch
        if tr.get("eval_per_epoch", True) and tr.eval_interval_epoch and ep_done % tr.eval_interval_epoch == 0:
            evaluate()
        if rank == 0 and tr.save_ckpt_interval_epoch and ep_done % tr.save_ckpt_interval_epoch == 0:
            trainer.check_exchange()          # never a checkpoint of replicas that may have diverged
            log("saved", save_checkpoint(args.model_dir, model, trainer))
    if n_timed:
        log("host time per logged iteration: loader (draw + decode-if-new + one rssf_input_pipeline launch) %.2f ms, step enqueue+sync %.2f ms; "
            "resident tiles %.0f %%" % (1e3 * t_load / n_timed, 1e3 * t_step / n_timed, 100 * loader.resident_fraction()))
    if rank == 0:
        trainer.check_exchange()
        log("saved", save_checkpoint(args.model_dir, model, trainer))
    if tr.get("eval_after_train", True):
        evaluate()
    shutdown_distributed(trainer)


if __name__ == "__main__":
    main()
