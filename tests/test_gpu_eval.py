"""GPU parity of the evaluation path: eval-mode forward, six-transform TTA machinery with the
bilinear `Scale`, argmax + ignore mask + confusion matrix kernel - against the golden vectors produced by the reference's
own eval code (oracle/make_golden.py::case_eval) and against plain torch on the same inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.procedural import seeded_input
from tests.helpers import golden, rel_err
from tests.test_gpu_model import build

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,K,H,W", [(2, 6, 33, 47), (1, 7, 8, 8), (3, 19, 16, 5)])
def test_argmax_confusion_kernel(B, K, H, W, dtype):
    from representationlearning_amd import ops
    torch.manual_seed(B * 100 + K)
    s = torch.randn(B, K, H, W).to(dtype)
    s[0, :, 0, 0] = 1.0                                           # a tie: the first maximum wins, as torch.argmax
    y = torch.randint(-1, K, (B, H, W))
    ref_pred = s.float().argmax(1)
    keep = y != -1
    ref_cm = torch.bincount(y[keep] * K + ref_pred[keep], minlength=K * K).reshape(K, K)
    cm = torch.zeros(K, K, dtype=torch.int64, device=DEV)
    sd = s.to(DEV).contiguous(memory_format=torch.channels_last)
    pred = ops.argmax_confusion(sd, y.to(DEV), cm)
    ops.argmax_confusion(sd, y.to(DEV), cm, want_pred=False)      # accumulates
    assert torch.equal(pred.cpu().long(), ref_pred)
    assert torch.equal(cm.cpu(), 2 * ref_cm)


@pytest.mark.parametrize("factor", [0.5, 0.75, 1.25, 1.75])
def test_scale_transform_matches_interpolate(factor):
    from representationlearning_amd.module.tta import Scale
    torch.manual_seed(3)
    x = torch.randn(2, 5, 24, 40)
    t = Scale(scale_factor=factor)
    ref = F.interpolate(x, scale_factor=factor, mode="bilinear", align_corners=True)
    got = t.transform(x.to(DEV))
    assert got.shape == ref.shape and rel_err(got.cpu(), ref) < 1e-5
    back_ref = F.interpolate(ref, size=(24, 40), mode="bilinear", align_corners=True)
    assert rel_err(t.inv_transform(got).cpu(), back_ref) < 1e-5


def test_flip_rotate_transforms_round_trip():
    from representationlearning_amd.module.tta import Rotate90k, HorizontalFlip, VerticalFlip, Transpose, Identity, tta
    x = torch.randn(1, 3, 6, 6, device=DEV)
    for t in (Identity(), Rotate90k(1), Rotate90k(3), HorizontalFlip(), VerticalFlip(), Transpose()):
        assert torch.equal(t.inv_transform(t.transform(x)), x)
    # an equivariant "model" (identity) is a fixed point of the averaged TTA
    assert rel_err(tta(lambda im: im, x, [Identity(), HorizontalFlip(), Rotate90k(2)]).cpu(), x.cpu()) < 1e-6


def test_eval_probs_tta_confusion_vs_reference():
    """Tiny 2x3x64x64, eval mode: probabilities, 3-scale TTA and confusion matrices of the reference's eval.py path."""
    from representationlearning_amd.module.tta import tta, Scale
    from representationlearning_amd.metric import PixelMetric
    g = golden("eval_tiny_2x64")
    m = build("tiny").eval()
    x = seeded_input((2, 3, 64, 64), 11).to(DEV)
    y = torch.from_numpy(g["y"])
    with torch.no_grad():
        probs = m(x)
        out = tta(m, x, [Scale(scale_factor=float(s)) for s in g["scales"]])
    assert rel_err(probs.cpu(), g["probs"]) < 1e-3
    assert rel_err(out.cpu(), g["tta"]) < 1e-3
    for scores, key in ((probs, "cm"), (out, "cm_tta")):
        metric = PixelMetric(6)
        metric.forward_scores(y.to(DEV), scores)
        metric._sync()
        # a handful of near-tie pixels may flip class between fp32 implementations: total count exact, cells within 0.2 %
        assert int(metric.cm.sum()) == int(g[key].sum())
        assert np.abs(metric.cm.numpy() - g[key]).sum() <= max(4, 0.002 * g[key].sum()), (metric.cm, g[key])
    s = PixelMetric(6)
    s.forward_scores(y.to(DEV), out)
    assert 0.0 <= s.summary_all()["miou"] <= 1.0


def test_evaluate_entry_point_with_tta():
    """eval.evaluate(): checkpoint-less model, synthetic tile, with and without the six-scale TTA."""
    import eval as ev
    from representationlearning_amd.configs import synthetic_batch
    batches = [synthetic_batch(1, 128, classes=6, seed=3)]
    for use_tta in (False, True):
        out = ev.evaluate(None, "baseline.hrnetw32", use_tta, batches=batches)
        assert set(out) >= {"miou", "iou", "overall_accuracy"} and 0.0 <= out["overall_accuracy"] <= 1.0


def test_evaluate_writes_palette_pngs(tmp_path):
    """eval.py:73-77 of the reference: label and prediction maps as palette PNGs; the prediction file equals the argmax of the
    model's own scores."""
    import numpy as np
    from PIL import Image
    import eval as ev
    from representationlearning_amd.configs import synthetic_batch
    img, lab = synthetic_batch(2, 64, classes=6, seed=5)
    ev.evaluate(None, "baseline.hrnetw32", False, batches=[(img, lab)], vis_dir=str(tmp_path))
    names = sorted(p.name for p in tmp_path.iterdir())
    assert names == ["0000_00.png", "0000_01.png", "gt0000_00.png", "gt0000_01.png"]
    gt = np.asarray(Image.open(tmp_path / "gt0000_01.png"))
    assert np.array_equal(gt, lab[1].cpu().numpy().astype(np.uint8) & 15)      # 4-bit palette file: ignore (-1 -> 255) reads as 15
    pred = np.asarray(Image.open(tmp_path / "0000_00.png"))
    assert pred.shape == (64, 64) and pred.max() < 6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,K,IH,IW,s", [(2, 6, 16, 16, 4), (1, 7, 9, 13, 4), (1, 19, 5, 7, 3)])
def test_head_upsample_softmax_kernel(B, K, IH, IW, s, dtype):
    """x s bilinear (align_corners=True) + softmax + argmax in one launch == F.interpolate + softmax + argmax."""
    from representationlearning_amd import nnf
    torch.manual_seed(K)
    lg = torch.randn(B, K, IH, IW).to(dtype)
    ref_l = F.interpolate(lg.float(), size=(IH * s, IW * s), mode="bilinear", align_corners=True)
    ref_p = ref_l.softmax(1)
    probs, pred = nnf.head_upsample_softmax(lg.to(DEV).contiguous(memory_format=torch.channels_last), (IH * s, IW * s), want_pred=True)
    assert probs.shape == ref_p.shape and probs.dtype == torch.float32
    assert rel_err(probs.cpu(), ref_p) < 2e-6
    assert float((probs.sum(1) - 1).abs().max()) < 1e-5
    # argmax: identical wherever the reference's top-2 margin is above fp32 rounding of the interpolation
    top2 = ref_l.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5
    assert torch.equal(pred.cpu().long()[clear], ref_l.argmax(1)[clear])


def test_predict_writes_class_maps(tmp_path):
    """predict.py surface (reference predict.py:29-48): one uint8 class-id PNG per tile; the fused head's class map equals the
    argmax of the eval-mode probabilities."""
    import predict
    from PIL import Image
    files = predict.predict_test(None, "baseline.hrnetw18", str(tmp_path), image_dir=str(tmp_path / "no_such_dir"), synthetic_tiles=2)
    assert len(files) == 2
    arr = np.asarray(Image.open(files[0]))
    assert arr.dtype == np.uint8 and arr.shape == (512, 512) and arr.max() < 7
    m = build("tiny", classes=7).eval()
    x = seeded_input((1, 3, 64, 64), 5).to(DEV)
    with torch.no_grad():
        pr = m(x)
    assert torch.equal(m.predict(x).long().cpu(), pr.argmax(1).cpu())


def test_loveda_folder_reader_and_device_loader(tmp_path):
    """data/loveda.py: the reference's folder layout (images_png/*.png + masks_png/<same name>), `mask - 1`, Normalize with
    max_pixel_value=1 on the GPU; batches keep the file names (predict.py / eval.py name their outputs after them)."""
    from PIL import Image
    from representationlearning_amd.data.loveda import DeviceLoader, LoveDA, LOVEDA_MEAN, LOVEDA_STD
    rng = np.random.default_rng(0)
    (tmp_path / "images_png").mkdir(); (tmp_path / "masks_png").mkdir()
    imgs, masks = {}, {}
    for name in ("10.png", "7.png", "3.png"):
        imgs[name] = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
        masks[name] = rng.integers(0, 8, (32, 32), dtype=np.uint8)
        Image.fromarray(imgs[name]).save(tmp_path / "images_png" / name)
        Image.fromarray(masks[name]).save(tmp_path / "masks_png" / name)
    ds = LoveDA([str(tmp_path / "images_png")], [str(tmp_path / "masks_png")])
    assert len(ds) == 3
    seen = []
    for img, gt in DeviceLoader(ds, batch_size=2):
        for i, name in enumerate(gt["fname"]):
            want = (imgs[name].astype(np.float32) - np.array(LOVEDA_MEAN, np.float32)) * (1.0 / np.array(LOVEDA_STD, np.float32))
            assert np.allclose(img[i].permute(1, 2, 0).cpu().numpy(), want, atol=1e-5)
            assert np.array_equal(gt["cls"][i].cpu().numpy(), masks[name].astype(np.int64) - 1)
            seen.append(name)
    assert sorted(seen) == sorted(imgs)
