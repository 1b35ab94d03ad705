import sys, torch
sys.path.insert(0, ".")
from tests.helpers import golden, rel_err
from oracle.procedural import procedural_state, proc_input
from representationlearning_amd.module.baseline.base_hrnet.modules.DAL import Mhca
DEV = "cuda"
for rep in range(3):
    C, nw, tag = 32, 3, "c32"
    g = golden(f"mhca_{tag}")
    m = Mhca(C, 2, dropout=0.0); m.load_state_dict(procedural_state(m.state_dict())); m = m.to(DEV).train()
    x = proc_input((49, nw, C), 0.3).to(DEV).requires_grad_()
    y = proc_input((49, nw, C), 1.1).to(DEV).requires_grad_()
    out = m(x, y, y)
    (out * proc_input(out.shape, 2.0).to(DEV)).sum().backward()
    print("out", rel_err(out.detach().cpu(), g["out"]), "gx", rel_err(x.grad.cpu(), g["gx"]), "gy", rel_err(y.grad.cpu(), g["gy"]))
    d = (y.grad.cpu() - torch.from_numpy(g["gy"])).abs()
    print(" gy err by window", d.amax(dim=(0, 2)).tolist(), "by token max idx", d.amax(dim=(1, 2)).argmax().item(), "by channel", [round(v, 5) for v in d.amax(dim=(0, 1)).tolist()][:32])
    for k, p in m.named_parameters():
        print("  ", k, rel_err(p.grad.cpu(), g["g_" + k.replace(".", "_")]))
