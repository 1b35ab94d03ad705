"""SCD class-activation maps (BASELINE config 5 as worded): multi_scale_cam of TSCD(mit_b1, stride [4, 2, 2, 1]) on synthetic VOC-sized
3 x 321 x 321 images, scales [1, 0.5, 1.5] (configs/voc_attn_reg.yaml) - images per second (one "image" = its six forwards: three
scales x {image, flip}), the FLOPs of the GEMM-shaped work counted from the launches themselves, and the stand-alone rate of the
attention kernel at the stage-4 geometry of the largest scale.  One JSON line per (dtype, batch).
  python tools/scd_cam_bench.py [B ...]      default: 2 (the reference's samples_per_gpu) and 16"""
import json, sys, time, torch
sys.path.insert(0, ".")
from representationlearning_amd import nnf, ops
from representationlearning_amd.scd.network.TSCD_model import TSCD
from representationlearning_amd.scd.utils.camutils import GraphedMultiScaleCam, multi_scale_cam
torch.manual_seed(0)
m = TSCD("mit_b1", num_classes=21, embedding_dim=256, stride=[4, 2, 2, 1], pretrained=False, pooling="gmp").eval().cuda()
SCALES = [1, 0.5, 1.5]
MFMA_BF16_PEAK, MFMA_F32_PEAK = 2.5e15, 157.3e12     # MI355X_MICROARCH.md: dense bf16 MFMA; fp32 matrix peak

flops = [0.0]
_fwd, _mha = nnf._conv_forward, ops.mha_fwd


def counted(spec, xh, weights, bias, stats, rt=None, addend=None, preact=None, cache_pack=False):
    out = _fwd(spec, xh, weights, bias, stats, rt, addend=addend, preact=preact, cache_pack=cache_pack)
    if spec.parts is None:
        flops[0] += 2.0 * out.numel() * spec.cin * spec.ntaps
    return out


def counted_mha(q, kv, heads, scale, want_logits=False):
    flops[0] += 4.0 * q.shape[0] * q.shape[1] * kv.shape[1] * q.shape[2]          # q k^T and p v
    return _mha(q, kv, heads, scale, want_logits)


def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


for B in ([int(a) for a in sys.argv[1:]] or [2, 16]):
    x = torch.randn(B, 3, 321, 321, device="cuda")
    for dt in (torch.float32, torch.bfloat16):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
            nnf._conv_forward, ops.mha_fwd, flops[0] = counted, counted_mha, 0.0
            multi_scale_cam(m, x, SCALES)
            nnf._conv_forward, ops.mha_fwd = _fwd, _mha
            dtm = timed(lambda: multi_scale_cam(m, x, SCALES), 10)
        fn = GraphedMultiScaleCam(m, x, SCALES, autocast_dtype=torch.bfloat16 if dt == torch.bfloat16 else None)
        dtg = timed(lambda: fn(x), 20)
        # the attention kernels alone: stage 4 of the 1.5 x scale (31 x 31 = 961 tokens attend to all 961, 8 heads of 64)
        q = torch.randn(2 * B, 961, 512, device="cuda", dtype=dt); kv = torch.randn(2 * B, 961, 1024, device="cuda", dtype=dt)
        wp, bp = m.attn_proj.weight.detach(), m.attn_proj.bias.detach()
        ta = timed(lambda: _mha(q, kv, 8, 0.125), 20)
        tp = timed(lambda: ops.attn_pred(q, kv, q, kv, wp, bp, 8), 20)
        fa = 4.0 * 2 * B * 961 * 961 * 512
        peak = MFMA_BF16_PEAK if dt == torch.bfloat16 else MFMA_F32_PEAK
        print(json.dumps({"metric": "multi-scale CAM images/sec, SCD TSCD(mit_b1, stride 4-2-2-1, 21 classes), 3x321x321, scales 1/0.5/1.5 x flip",
                          "value": round(B / dtm, 1), "unit": "images/s", "dtype": "bf16" if dt == torch.bfloat16 else "f32", "batch": B,
                          "ms_per_batch": round(dtm * 1e3, 3), "step_launch": "eager",
                          "graph_replay": {"value": round(B / dtg, 1), "ms_per_batch": round(dtg * 1e3, 3)}, "gemm_gflop_per_image": round(flops[0] / B / 1e9, 3),
                          "roofline": {"bound": "mfma", "achieved": round(flops[0] / dtm / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                                       "frac": round(flops[0] / dtm / peak, 4), "what": "Linear / convolution / attention FLOPs of one multi_scale_cam / wall time"},
                          "mha_kernel": {"shape": "B=%d N=M=961 heads=8 d=64" % (2 * B), "us": round(ta * 1e6, 1), "tflops": round(fa / ta / 1e12, 2)},
                          "attn_pred_kernel": {"shape": "same, two blocks", "us": round(tp * 1e6, 1), "tflops": round(fa / tp / 1e12, 2),
                                               "write_GBps": round(2 * B * 961 * 961 * 4 / tp / 1e9, 1)}}), flush=True)
