"""WaveCAM ResNet-50 CAM inference (BASELINE config 5's conv-only relative): images per second at 321 x 321, with the whole-network
MFMA fraction (convolution FLOPs counted from the launches themselves) - one JSON line per (dtype, batch).
  python tools/cam_bench.py [B ...]      default: 2 (the reference's image + flip pair) and 32"""
import json, sys, time, torch
sys.path.insert(0, ".")
from representationlearning_amd import nnf
from representationlearning_amd.wavecam.net.resnet50_cam import CAM
torch.manual_seed(0)
m = CAM(stride=16, n_classes=20); m.eval(); m = m.cuda()
MFMA_BF16_PEAK, MFMA_F32_PEAK = 2.5e15, 157.3e12     # MI355X_MICROARCH.md: dense bf16 MFMA; fp32 (xf32-free) matrix peak

flops = [0.0]
_fwd = nnf._conv_forward


def counted(spec, xh, weights, bias, stats, rt=None, addend=None, preact=None):
    out = _fwd(spec, xh, weights, bias, stats, rt, addend=addend, preact=preact)
    if spec.parts is None:
        flops[0] += 2.0 * out.numel() * spec.cin * spec.ntaps
    return out


for B in ([int(a) for a in sys.argv[1:]] or [2, 32]):
    x = torch.randn(B, 3, 321, 321, device="cuda")
    for dt in (torch.float32, torch.bfloat16):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
            nnf._conv_forward, flops[0] = counted, 0.0
            m(x, separate=True)
            nnf._conv_forward = _fwd
            for _ in range(5): m(x, separate=True)
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(30): m(x, separate=True)
            torch.cuda.synchronize()
        dtm = (time.perf_counter() - t) / 30
        peak = MFMA_BF16_PEAK if dt == torch.bfloat16 else MFMA_F32_PEAK
        print(json.dumps({"metric": "CAM inference images/sec, WaveCAM ResNet-50 CAM (stride 16, 20 classes), 3x321x321", "value": round(B / dtm, 1),
                          "unit": "images/s", "dtype": "bf16" if dt == torch.bfloat16 else "f32", "batch": B, "ms_per_forward": round(dtm * 1e3, 3),
                          "step_launch": "eager", "conv_gflop_per_image": round(flops[0] / B / 1e9, 3),
                          "roofline": {"bound": "mfma", "achieved": round(flops[0] / dtm / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                                       "frac": round(flops[0] / dtm / peak, 4), "what": "all convolutions of one forward / wall time"}}), flush=True)
