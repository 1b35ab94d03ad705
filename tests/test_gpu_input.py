"""Device input pipeline against the numpy restatement of the albumentations chain (oracle/input_cpu.py,
parity unpinned: see its header).  Bit-exact: labels are integers, the fp32 image is the same two float32 roundings, the bf16
image is that value rounded to nearest-even."""
import numpy as np
import pytest
import torch

from oracle import input_cpu as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tiles(n, H, W, seed):
    rng = np.random.RandomState(seed)
    return rng.randint(0, 256, size=(n, H, W, 3)).astype(np.uint8), rng.randint(0, 8, size=(n, H, W)).astype(np.uint8)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_every_op_matches_oracle_bit_exact(dtype):
    from representationlearning_amd.data import DeviceAugment, LOVEDA_MEAN, LOVEDA_STD
    img, msk = _tiles(3, 40, 52, 1)
    aug = DeviceAugment(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV), crop=32, dtype=dtype, seed=0)
    params = np.array([[0, 0, 0, 0], [1, 8, 20, 1], [2, 3, 5, 2], [0, 8, 0, 3], [1, 0, 20, 4], [2, 7, 19, 5], [1, 4, 4, 6]], dtype=np.int32)
    x, y = aug.apply(params)
    ri, rl = O.pipeline(img, msk, params, 32, LOVEDA_MEAN, LOVEDA_STD)
    assert x.shape == (7, 3, 32, 32) and x.is_contiguous(memory_format=torch.channels_last) and y.dtype == torch.int64
    assert np.array_equal(y.cpu().numpy(), rl) and rl.min() == -1
    ref = torch.from_numpy(ri).permute(0, 3, 1, 2)
    if dtype == torch.bfloat16:
        ref = ref.bfloat16()
    assert torch.equal(x.cpu(), ref)


def test_random_draws_in_range_and_reference_size_properties():
    """1024 x 1024 tiles, 512 crops (the reference's sizes): draws stay inside the tile, all ops occur, flipping twice and four
    quarter turns are the identity, and a crop of a crop equals the direct crop."""
    from representationlearning_amd.data import DeviceAugment
    img, msk = _tiles(2, 1024, 1024, 2)
    aug = DeviceAugment(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV), crop=512, seed=3)
    p = aug.draw(256)
    assert p[:, 1].min() >= 0 and p[:, 1].max() <= 512 and p[:, 2].min() >= 0 and p[:, 2].max() <= 512 and set(p[:, 3]) == set(range(7))
    assert 0.15 < (p[:, 3] == 0).mean() < 0.35          # OneOf(p=0.75): a quarter untouched
    base = np.array([[1, 100, 200, 0]], dtype=np.int32)
    x0, y0 = aug.apply(base)
    for op, inv in ((1, lambda t: t.flip(-1)), (2, lambda t: t.flip(-2)), (4, lambda t: torch.rot90(t, -1, (-2, -1))),
                    (5, lambda t: torch.rot90(t, -2, (-2, -1))), (6, lambda t: torch.rot90(t, -3, (-2, -1)))):
        q = base.copy(); q[0, 3] = op
        x, y = aug.apply(q)
        assert torch.equal(inv(x), x0) and torch.equal(inv(y), y0)
    xb, yb = aug(4)      # the __call__ path
    assert xb.shape == (4, 3, 512, 512) and yb.shape == (4, 512, 512) and int(yb.min()) >= -1 and int(yb.max()) <= 6


def test_bad_parameters_are_rejected():
    from representationlearning_amd.data import DeviceAugment
    img, msk = _tiles(1, 16, 16, 0)
    aug = DeviceAugment(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV), crop=8)
    for bad in ([[1, 0, 0, 0]], [[0, 9, 0, 0]], [[0, 0, 0, 7]], [[0, -1, 0, 0]]):
        with pytest.raises(ValueError):
            aug.apply(np.array(bad, dtype=np.int32))
    with pytest.raises(ValueError):
        DeviceAugment(torch.from_numpy(img).to(DEV), crop=32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_shift_scale_rotate_matches_oracle_bit_exact(dtype):
    """ShiftScaleRotate after crop + flip (configs/base/loveda.py:31): the fixed-point cv2.warpAffine restatement of the kernel and
    of the numpy oracle agree bit for bit on image and mask, for rotations, scales and shifts incl. the reflected border."""
    from representationlearning_amd.data import DeviceAugment, LOVEDA_MEAN, LOVEDA_STD
    from representationlearning_amd.data.loveda import ssr_inverse_matrix
    img, msk = _tiles(2, 48, 56, 5)
    S = 32
    aug = DeviceAugment(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV), crop=S, dtype=dtype, seed=0)
    params = np.array([[0, 0, 0, 0], [1, 8, 20, 1], [0, 3, 5, 2], [1, 8, 0, 4], [0, 0, 20, 5], [1, 7, 19, 6], [0, 4, 4, 0]], dtype=np.int32)
    warps = [(30.0, 1.1, 0.05, -0.03), (-44.9, 0.8, -0.0625, 0.0625), (0.0, 1.0, 0.0, 0.0), (12.5, 1.2, 0.0, 0.01), (90.0, 1.0, 0.0, 0.0),
             (-7.0, 0.93, 0.02, 0.02), None]
    aff = np.full((7, 6), np.nan)
    for b, w in enumerate(warps):
        if w is not None:
            aff[b] = ssr_inverse_matrix(S, S, *w)
    x, y = aug.apply(params, aff)
    ri, rl = O.pipeline(img, msk, params, S, LOVEDA_MEAN, LOVEDA_STD, affine=aff)
    assert np.array_equal(y.cpu().numpy(), rl)
    ref = torch.from_numpy(ri).permute(0, 3, 1, 2)
    if dtype == torch.bfloat16:
        ref = ref.bfloat16()
    assert torch.equal(x.cpu(), ref)
    # the identity warp (angle 0, scale 1, no shift) is the un-warped crop; a NaN row is "no warp"
    plain, yp = aug.apply(params)
    assert torch.equal(x[2], plain[2]) and torch.equal(y[2], yp[2]) and torch.equal(x[6], plain[6])
    assert not torch.equal(x[0], plain[0])


def test_shift_scale_rotate_draws():
    from representationlearning_amd.data import DeviceAugment
    img, msk = _tiles(1, 64, 64, 6)
    aug = DeviceAugment(torch.from_numpy(img).to(DEV), torch.from_numpy(msk).to(DEV), crop=32, seed=1,
                        shift_scale_rotate=dict(shift_limit=0.0625, scale_limit=0.2, rotate_limit=45, p=0.2))
    a = aug.draw_affine(2000)
    frac = np.isfinite(a[:, 0]).mean()
    assert 0.16 < frac < 0.24                                  # p = 0.2
    m = a[np.isfinite(a[:, 0])]
    scale_inv = np.sqrt(m[:, 0] ** 2 + m[:, 1] ** 2)           # inverse matrix: 1 / scale
    assert scale_inv.min() > 1 / 1.2 - 1e-9 and scale_inv.max() < 1 / 0.8 + 1e-9
    xb, yb = aug(8)
    assert xb.shape == (8, 3, 32, 32) and int(yb.min()) >= -1


def test_training_loader_feeds_resident_tiles_through_the_device_pipeline():
    """LoveDALoader (train.py's data path): the sampler's tiles - not random ones - are cropped / flipped / normalised by ONE
    rssf_input_pipeline launch per batch out of the HBM-resident store; tiles are uploaded once; an epoch has n/world/batch batches;
    the result equals the oracle's transform chain on the same tile with the same drawn parameters (bit-exact, as above)."""
    from representationlearning_amd.data.loveda import LoveDALoader, SyntheticTiles, epoch_shard
    ds = SyntheticTiles(12, 96, classes=6)
    ld = LoveDALoader(ds, batch_size=4, rank=1, world=2, crop=64, dtype=torch.float32, seed=5, decode_threads=2)
    assert len(ld) == 1
    seen = []
    rng_state = ld.aug.rng.bit_generator.state
    for img, tgt in ld:
        assert img.shape == (4, 3, 64, 64) and tgt["cls"].shape == (4, 64, 64) and tgt["cls"].dtype == torch.int64
        seen.append((img.clone(), tgt["cls"].clone()))
    assert ld.epoch == 1 and 0 < ld.resident_fraction() <= 5 / 12 + 1e-9
    tiles = epoch_shard(12, 0, 1, 2, 4, 5)
    ld.aug.rng.bit_generator.state = rng_state            # redraw the same parameters
    params = ld.aug.draw(4)
    params[:, 0] = tiles
    aff = ld.aug.draw_affine(4)
    from representationlearning_amd.data import LOVEDA_MEAN, LOVEDA_STD
    imgs = np.stack([ds[i][0] for i in range(12)])
    msks = np.stack([ds[i][1]["raw_mask"] for i in range(12)])
    ri, rl = O.pipeline(imgs, msks, params, 64, LOVEDA_MEAN, LOVEDA_STD, affine=aff)
    assert torch.equal(seen[0][0].cpu(), torch.from_numpy(ri).permute(0, 3, 1, 2))
    assert np.array_equal(seen[0][1].cpu().numpy(), rl)


def test_train_py_runs_saves_and_resumes(tmp_path):
    """train.py end to end on synthetic tiles: iterates the loader, logs, evaluates, writes model-<it>.pth (reference keys) +
    trainer-<it>.pth, and a second invocation resumes at that iteration with the momentum restored."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train.py"), "--model_dir", str(tmp_path), "--batch", "2", "--size", "64", "--synthetic_tiles", "4",
           "--config_path", "baseline.hrnetw18", "train.log_interval_step", "1", "train.eval_interval_epoch", "1",
           "train.save_ckpt_interval_epoch", "2"]
    r = subprocess.run(cmd + ["--iters", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "iter 5" in r.stdout and "mIoU" in r.stdout and os.path.exists(tmp_path / "model-5.pth")
    # 4 tiles / batch 2 = 2 iterations per epoch: the interval of 2 epochs saves after iteration 4 and not after iteration 2 (ADVICE r5:
    # the save had slipped out of its rank-0 / interval guard)
    assert os.path.exists(tmp_path / "model-4.pth") and not os.path.exists(tmp_path / "model-2.pth"), sorted(os.listdir(tmp_path))
    sd = torch.load(tmp_path / "model-5.pth")
    assert "backbone.hrnet.stage2.0.transformer.attn.attn.q_proj.weight" in sd
    ts = torch.load(tmp_path / "trainer-5.pth")
    assert ts["it"] == 5 and float(ts["momentum"]["head.0.weight"].abs().sum()) > 0
    r = subprocess.run(cmd + ["--iters", "7"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "resumed from" in r.stdout and "iter 6" in r.stdout and "iter 5 " not in r.stdout and os.path.exists(tmp_path / "model-7.pth")
