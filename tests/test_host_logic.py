"""CPU: host-side logic of the product package (no kernels run): config merge, registry, state_dict compatibility
with the reference, tap lists, LR schedule, loud failure without a GPU, and oracle isolation."""
import os
import re

import pytest
import numpy as np
import torch
import torch.nn as nn

from tests.helpers import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_merge_and_overrides():
    from representationlearning_amd.core.config import AttrDict, apply_overrides
    c = AttrDict.wrap(dict(a=dict(b=1, c=dict(d=2)), e=3))
    c.merge(dict(a=dict(c=dict(d=5, f=6))))
    assert c.a.b == 1 and c.a.c.d == 5 and c.a.c.f == 6 and c.e == 3
    apply_overrides(c, ["a.c.d", "7", "e", "x"])          # scripts/train.sh-style `key value` pairs
    assert c.a.c.d == 7 and c.e == "x"


@pytest.mark.parametrize("variant", ["tiny", "base", "large"])
def test_state_dict_identical_to_reference(variant):
    """Key ORDER, names and shapes equal the reference model's (fixture captured from the reference itself)."""
    from representationlearning_amd.configs import rssformer_config
    from representationlearning_amd.core import registry
    registry.register_all()
    m = registry.MODEL["RSSFormer"](rssformer_config(variant))
    g = golden(f"keys_{variant}")
    sd = m.state_dict()
    assert list(sd.keys()) == g["names"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == g["shapes"].tolist()
    assert sum(p.numel() for p in m.parameters()) == int(g["nparams"])


def test_registry_call_forms_and_errors():
    from representationlearning_amd.core import registry
    from representationlearning_amd.module.baseline.base_hrnet._hrnet_rssformer import HighResolutionModule, BasicBlock
    registry.register_all()
    assert "RSSFormer" in registry.MODEL and "hrnetv2_w32" in registry.MODEL and "HRNetEncoder" in registry.MODEL
    with pytest.raises(ValueError):       # _hrnet_rssformer.py:313-326 of the reference
        HighResolutionModule(2, BasicBlock, (4,), [32, 64], (32, 64), "SUM")
    from representationlearning_amd.module.baseline.base_hrnet.modules.DAL import Mhca
    with pytest.raises(AssertionError):   # DAL.py:700-702
        Mhca(30, 4)


def test_conv_spec_taps():
    from representationlearning_amd.nnf import ConvSpec
    s = ConvSpec([nn.Conv2d(8, 8, 3, 2, 1)])
    assert s.ntaps == 9 and s.dy[0] == -1 and s.dx[8] == 1 and s.out_hw(16, 15) == (8, 8)
    f = ConvSpec([nn.Conv2d(8, 8, 1), nn.Conv2d(8, 8, 3, 1, 6, 6), nn.Conv2d(8, 8, 3, 1, 12, 12)])
    # the three centre positions sample the same pixel: ONE tap (the 1x1's) with the two 3x3 centres as weight aliases
    assert f.ntaps == 17 and sorted(set(f.dy)) == [-12, -6, 0, 6, 12] and f.src.count(0) == 1 and f.out_hw(10, 10) == (10, 10)
    assert f.alias[0] == [1, 4, 2, 4] and all(a == [-1] * 4 for a in f.alias[1:]) and len(set(zip(f.dy, f.dx))) == 17
    with pytest.raises(NotImplementedError):
        ConvSpec([nn.Conv2d(8, 8, 3, groups=2)])


def test_poly_lr_and_synthetic_batch():
    from representationlearning_amd.trainer import poly_lr
    from representationlearning_amd.configs import synthetic_batch
    assert poly_lr(0.01, 0.9, 30000, 0) == pytest.approx(0.01)
    assert 0 < poly_lr(0.01, 0.9, 30000, 29999) < 1e-5
    img, lab = synthetic_batch(2, 64, device="cpu")
    assert img.shape == (2, 3, 64, 64) and lab.shape == (2, 64, 64) and lab.min() >= -1 and lab.max() <= 5
    assert (lab[:, :16, :16] == lab[:, :1, :1]).all()        # 16x16 constant blocks


def test_ops_fail_loudly_without_gpu():
    """No CPU fallback: a CPU tensor is an error, not a silent eager path."""
    from representationlearning_amd import ops, nnf
    x = torch.randn(2, 10, 32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(x, torch.ones(32), torch.zeros(32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        nnf.conv_bias(torch.randn(1, 8, 4, 4), nn.Conv2d(8, 8, 1))


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "representationlearning_amd")):
        for f in files:
            if f.endswith(".py") and pat.search(open(os.path.join(root, f)).read()):
                bad.append(f)
    assert not bad, bad
    for f in ("train.py", "eval.py"):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert not pat.search(open(p).read()), f


def test_miou_metric():
    from representationlearning_amd.metric import PixelMetric
    m = PixelMetric(3)
    y_true = torch.tensor([0, 0, 1, 1, 2, 2, -1])
    y_pred = torch.tensor([0, 1, 1, 1, 2, 0, 2])
    keep = y_true != -1
    m.forward(y_true[keep], y_pred[keep])
    iou = m.iou()
    assert iou[0] == pytest.approx(1 / 3) and iou[1] == pytest.approx(2 / 3) and iou[2] == pytest.approx(1 / 2)
    assert m.miou() == pytest.approx((1 / 3 + 2 / 3 + 1 / 2) / 3)


def test_palette_png_writer(tmp_path):
    """module/viz.py (mirror of the reference's VisualizeSegmm, viz.py:6-23; byte-identical files were checked against the
    imported reference class in the build container): palette PNG with the LoveDA colours; PIL stores the 7-colour image at
    4 bits per pixel, so the ignore label (-1 -> uint8 255) reads back as 15 - the reference's files do the same."""
    import numpy as np
    from PIL import Image
    from representationlearning_amd.module.viz import LOVEDA_PALETTE, VisualizeSegmm
    y = np.array([[[0, 1, 2, 3], [4, 5, 6, -1]]], dtype=np.int64)
    VisualizeSegmm(str(tmp_path / "vis"), LOVEDA_PALETTE)(y, "a.png")
    im = Image.open(tmp_path / "vis" / "a.png")
    assert im.mode == "P" and im.size == (4, 2)
    assert np.array_equal(np.asarray(im), np.array([[0, 1, 2, 3], [4, 5, 6, 15]], dtype=np.uint8))
    assert im.getpalette()[:21] == LOVEDA_PALETTE and im.convert("RGB").getpixel((1, 0)) == (255, 0, 0)


def test_input_oracle_warp_affine_properties():
    """oracle/input_cpu.py::warp_affine_u8 (the cv2.warpAffine restatement, parity unpinned): identity, integer shifts with
    BORDER_REFLECT_101, nearest for masks, and the 180-degree rotation about the crop centre."""
    import numpy as np
    from oracle import input_cpu as O
    from representationlearning_amd.data.loveda import ssr_inverse_matrix
    rng = np.random.RandomState(0)
    im = rng.randint(0, 256, (16, 20, 3)).astype(np.uint8)
    mk = rng.randint(0, 8, (16, 20)).astype(np.uint8)
    ident = ssr_inverse_matrix(20, 16, 0, 1, 0, 0)
    assert np.array_equal(O.warp_affine_u8(im, ident), im) and np.array_equal(O.warp_affine_u8(mk, ident, nearest=True), mk)
    right3 = ssr_inverse_matrix(20, 16, 0, 1, 3 / 20, 0)
    w = O.warp_affine_u8(im, right3)
    assert np.array_equal(w[:, 3:], im[:, :-3])
    assert np.array_equal(w[:, 0], im[:, 3]) and np.array_equal(w[:, 1], im[:, 2]) and np.array_equal(w[:, 2], im[:, 1])   # ..3 2 1 | 0 1 2..
    down2 = ssr_inverse_matrix(20, 16, 0, 1, 0, 2 / 16)
    assert np.array_equal(O.warp_affine_u8(mk, down2, nearest=True)[2:], mk[:-2])
    # rotation by 180 degrees about (w/2, h/2): dst(x, y) = src(w - x, h - y) -> one pixel of reflected border on the top / left
    half = ssr_inverse_matrix(20, 16, 180, 1, 0, 0)
    r = O.warp_affine_u8(mk, half, nearest=True)
    assert np.array_equal(r[1:, 1:], mk[::-1, ::-1][:-1, :-1])


def test_fuse_lockstep_schedule_equals_reference_order(monkeypatch):
    """HighResolutionModule._fuse_lockstep (data-parallel path: one SyncBN exchange per depth group) runs every fuse convolution
    exactly once and reproduces the per-output sums of the plain path bit for bit - checked with CPU stand-ins for the fused
    ops, for the 2-, 3- and 4-branch modules of stages 2 / 3 / 4.  Groups (= SyncBN exchanges) per module: the outputs 1.. depth by depth
    (1 / 2 / 3 groups, on the side stream of nnf.fork_side) and ONE group for output 0's 1x1 convolutions in front of the transformer."""
    import torch.nn as nn
    import torch.nn.functional as F
    from representationlearning_amd import nnf
    from representationlearning_amd.module.baseline.base_hrnet import _hrnet_rssformer as H
    calls = []

    def fake_group(items):
        calls.append([id(d["conv"]) for d in items])
        outs = []
        for d in items:
            c = d["conv"]
            y = F.conv2d(d["x"], c.weight, None, c.stride, c.padding)
            if d.get("res_pre") is not None:
                y = y + d["res_pre"]
            outs.append(torch.relu(y) if d.get("act") == nnf.ACT_RELU else y)
        return outs

    def fake_up(acc, x, scale):
        u = F.interpolate(x, scale_factor=scale, mode="nearest")
        return u if acc is None else acc + u

    monkeypatch.setattr(nnf, "conv_bn_act_group", fake_group)
    monkeypatch.setattr(nnf, "upsample_nearest_add", fake_up)

    class PassLow(nn.Module):
        def forward(self, low, x0):
            return low

    def chain(sq, x, res_pre=None, act_last=False):
        mods = list(sq)
        for k, st in enumerate(mods):
            x = F.conv2d(x, st[0].weight, None, st[0].stride, st[0].padding)
            last = k == len(mods) - 1
            if res_pre is not None and last:
                x = x + res_pre
            if len(st) > 2 or (last and act_last):
                x = torch.relu(x)
        return x

    torch.manual_seed(0)
    for nb, ch, groups in ((2, [8, 16], [1, 1]), (3, [8, 16, 24], [3, 2, 2]), (4, [8, 16, 24, 32], [8, 4, 1, 3])):
        m = H.HighResolutionModule(nb, H.BasicBlock, [1] * nb, list(ch), list(ch), "SUM")
        m.transformer = PassLow()
        xs = [torch.randn(1, ch[i], 16 >> i, 16 >> i) for i in range(nb)]
        calls.clear()
        outs = m._fuse_lockstep(xs)
        assert [len(g) for g in calls] == groups
        assert sorted(c for g in calls for c in g) == sorted(id(mod) for mod in m.fuse_layers.modules() if isinstance(mod, nn.Conv2d))
        for i in range(nb):
            low = None
            for j in range(1, nb):
                if j == i:
                    low = xs[j] if low is None else low + xs[j]
                elif j > i:
                    fl = m.fuse_layers[i][j]
                    low = fake_up(low, F.conv2d(xs[j], fl[0].weight), int(fl[2].scale_factor))
                else:
                    t = chain(m.fuse_layers[i][j], xs[j])
                    low = t if low is None else low + t
            ref = torch.relu(low) if i == 0 else chain(m.fuse_layers[i][0], xs[0], res_pre=low, act_last=True)
            assert torch.equal(ref, outs[i]), (nb, i)


def test_padblock_and_local_permute_methods_match_oracle_partition():
    """The reference's PadBlock.pad_if_needed / depad_if_needed and LocalPermuteModule.permute / rev_permute
    (multihead_isa_attention.py:373-426) on the mirror classes: same tensors as the oracle's window_partition / window_merge."""
    from oracle import rssformer_cpu as O
    from representationlearning_amd.module.baseline.base_hrnet.modules.multihead_isa_attention import LocalPermuteModule, PadBlock
    pad, perm = PadBlock(7), LocalPermuteModule(7)
    for (B, H, W, C) in ((2, 10, 10, 5), (1, 14, 21, 3), (1, 9, 11, 4), (3, 7, 7, 2)):
        t = torch.randn(B, H, W, C)
        p = pad.pad_if_needed(t, t.shape)
        Hp, Wp = pad.padded_size(H, W)
        assert p.shape == (B, Hp, Wp, C)
        w = perm.permute(p, (B, Hp, Wp, C))
        ref, geom = O.window_partition(t)
        assert torch.equal(w, ref)
        back = pad.depad_if_needed(perm.rev_permute(w, (B, Hp, Wp, C)), t.shape)
        assert torch.equal(back, t) and torch.equal(O.window_merge(ref, geom), t)
        n_, slot = perm.window_of(B - 1, Hp - 1, Wp - 1, Hp, Wp)
        assert n_ == w.shape[1] - 1 and slot == 48


def test_training_loader_epoch_shards_are_disjoint_and_reshuffled():
    """data/loveda.py:97-117 semantics (StepDistributedSampler + drop_last): per epoch ONE permutation, every rank a disjoint share of
    equal length in whole batches; another epoch, another order; the same (epoch, seed) on every rank."""
    from representationlearning_amd.data.loveda import epoch_shard
    n, world, bs = 103, 4, 8
    for epoch in (0, 1):
        shards = [epoch_shard(n, epoch, r, world, bs) for r in range(world)]
        assert all(len(s) == (n // world // bs) * bs for s in shards)
        flat = np.concatenate(shards)
        assert len(set(flat.tolist())) == len(flat)                       # disjoint
        assert flat.min() >= 0 and flat.max() < n
    assert not np.array_equal(epoch_shard(n, 0, 0, world, bs), epoch_shard(n, 1, 0, world, bs))
    assert np.array_equal(epoch_shard(n, 3, 2, world, bs), epoch_shard(n, 3, 2, world, bs))
    assert len(epoch_shard(16, 0, 0, 1, 16)) == 16


def test_synthetic_tiles_look_like_loveda_items():
    from representationlearning_amd.data.loveda import SyntheticTiles
    ds = SyntheticTiles(3, 64, classes=6)
    img, tgt = ds[2]
    img2, tgt2 = ds[2]
    assert img.dtype == np.uint8 and img.shape == (64, 64, 3) and np.array_equal(img, img2)
    assert tgt["raw_mask"].dtype == np.uint8 and tgt["raw_mask"].max() <= 6 and tgt["cls"].min() >= -1
    assert np.array_equal(tgt["cls"], tgt["raw_mask"].astype(np.int64) - 1)


def test_train_checkpoint_discovery(tmp_path):
    import importlib
    train = importlib.import_module("train")
    assert train.latest_checkpoint(str(tmp_path)) is None
    for k in (5, 40, 12):
        (tmp_path / ("model-%d.pth" % k)).write_bytes(b"x")
    (tmp_path / "trainer-40.pth").write_bytes(b"x")
    step, mp, tp = train.latest_checkpoint(str(tmp_path))
    assert step == 40 and mp.endswith("model-40.pth") and tp.endswith("trainer-40.pth")


def test_train_py_saves_checkpoints_on_rank_zero_only():
    """Every save_checkpoint call of train.main sits under an `if` that tests `rank == 0` (the per-epoch save had been
    dedented out of its guard - every rank of a data-parallel run then wrote and renamed the same files)."""
    import ast, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "train.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    parents = {}
    for node in ast.walk(main):
        for ch in ast.iter_child_nodes(node):
            parents[ch] = node
    calls = [n for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", None) == "save_checkpoint"]
    assert len(calls) >= 2
    for c in calls:
        node, guarded = c, False
        while node in parents:
            prev, node = node, parents[node]
            if isinstance(node, ast.If) and prev in node.body and "rank == 0" in ast.unparse(node.test):
                guarded = True
        assert guarded, "save_checkpoint at line %d is not under a rank-0 guard" % c.lineno
    # the per-epoch save also honours its interval
    src = open(os.path.join(root, "train.py")).read()
    assert src.count("trainer.check_exchange()          # never a checkpoint") == 1
    guard = [n for n in ast.walk(main) if isinstance(n, ast.If) and "save_ckpt_interval_epoch" in ast.unparse(n.test)]
    assert guard and any(isinstance(x, ast.Call) and getattr(x.func, "id", None) == "save_checkpoint" for g in guard for x in ast.walk(g))
