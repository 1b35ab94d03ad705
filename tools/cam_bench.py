"""WaveCAM ResNet-50 CAM inference (BASELINE config 5's conv-only relative): image pairs (image + flip) per second at 321 x 321."""
import sys, time, torch
sys.path.insert(0, ".")
from representationlearning_amd.wavecam.net.resnet50_cam import CAM
torch.manual_seed(0)
m = CAM(stride=16, n_classes=20); m.eval(); m = m.cuda()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
x = torch.randn(B, 3, 321, 321, device="cuda")
for dt in (torch.float32, torch.bfloat16):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
        for _ in range(5): m(x, separate=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(30): m(x, separate=True)
        torch.cuda.synchronize()
    dtm = (time.perf_counter() - t) / 30
    print("%s  B=%d  %.2f ms per forward  %.0f images/s" % (str(dt).split(".")[-1], B, dtm * 1e3, B / dtm), flush=True)
