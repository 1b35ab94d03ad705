import sys; sys.path.insert(0,'.')
import torch
from representationlearning_amd import _lib as L
v=torch.arange(64,device='cuda').float()+1; out=torch.zeros(10,64,device='cuda')
L.check(L.load().rssf_debug_lane_reduce(L.ptr(v),L.ptr(out),L.stream()),'x')
o=out.cpu()
for k,name in ((0,'xor16 sum'),(1,'xor32 sum'),(2,'rows sum'),(3,'rows max'),(4,'wave sum'),(5,'wave max'),(6,'p16 r0'),(7,'p16 r1'),(8,'p32 r0'),(9,'p32 r1')):
    print(name, o[k].int().tolist())
