"""Training loader (data/loveda.py::LoveDALoader, what train.py iterates) over HBM-resident tiles: batches of 16 x 512^2 crops out of
64 synthetic 1024^2 uint8 tiles - first epoch (tiles decoded / drawn on the host and uploaded once) and steady state (resident).
Run on the GPU box from the repository root."""
import sys, time, torch
sys.path.insert(0, ".")
from representationlearning_amd.data.loveda import LoveDALoader, SyntheticTiles
ds = SyntheticTiles(64, 1024, classes=7)
t = time.perf_counter()
ld = LoveDALoader(ds, batch_size=16, crop=512, shift_scale_rotate=dict(shift_limit=0.0625, scale_limit=0.2, rotate_limit=45, p=0.2), dtype=torch.float32)
for ep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for img, tgt in ld:
        n += img.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("epoch %d: %d images in %.1f ms = %.0f images/s (resident tiles %.0f %%)" % (ep, n, dt * 1e3, n / dt, 100 * ld.resident_fraction()), flush=True)
