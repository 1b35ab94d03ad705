"""LayerNorm backward over [16, 128 x 128, 32] bf16 tokens (the three launches per transformer block): HIP-event time per call."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from representationlearning_amd import ops
B, N, C = 16, 128 * 128, 32
sets = []
for r in range(6):
    x = torch.randn(B, N, C, device="cuda").bfloat16(); dy = torch.randn(B, N, C, device="cuda").bfloat16()
    st = torch.rand(B * N, 2, device="cuda") + 0.5
    sets.append((x, dy, st))
g = torch.rand(C, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
for k in range(5): ops.layernorm_bwd(sets[k][1], sets[k][0], sets[k][2], g, dg, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(60):
    x, dy, st = sets[k % 6]
    ops.layernorm_bwd(dy, x, st, g, dg, db)
e1.record(); torch.cuda.synchronize()
print("layernorm_bwd %.1f us per call (%s)" % (e0.elapsed_time(e1) * 1000 / 60, "no atomics tail" if os.environ.get("RSSF_LN_DBG_NOATOM") else "as shipped"))
