"""Loss sequences of eager vs graph-replayed deterministic training (debug aid for tests/test_gpu_trainer.py)."""
import os, sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
os.environ.setdefault("RSSF_BRANCH_STREAMS", "1")
from representationlearning_amd.trainer import Trainer
from representationlearning_amd.configs import synthetic_batch
from test_gpu_trainer import _mk
img, lab = synthetic_batch(2, 128, seed=5)
for mode in sys.argv[1:] or ["e", "e", "g", "g"]:
    t = Trainer(_mk(6), bf16=True, base_lr=0.002, use_graph=(mode[0] == "g"), deterministic=True)
    ls = []
    for i in range(9):
        if mode == "gs" and i >= 4:
            torch.cuda.synchronize()
        ls.append(float(t.step(img, dict(cls=lab))))
    print(mode, " ".join("%.5f" % l for l in ls), flush=True)
