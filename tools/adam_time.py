#!/usr/bin/env python3
"""GPU time of the optimiser step of the training loop: torch's fused Adam over the 4K per-factor Parameters against the
same torch._fused_adam_ over the module's 4 stacked buffers.  usage (GPU box): python tools/adam_time.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd.model import Disentangle
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Disentangle(128, 512, 64, nfactor=8, beta=0.5, t=1).to(dev)
for p in model.parameters():
    p.grad = torch.randn_like(p) * 1e-3

def gpu_time(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for fused in (True, False):
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, fused=fused)
    print(f"torch Adam fused={fused} over {len(list(model.parameters()))} parameters: {gpu_time(opt.step):.1f} us per step")
opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=5e-4, fused=True, capturable=True)
print(f"torch Adam fused, capturable: {gpu_time(opt.step):.1f} us per step")
bufs = list(model._stacked.values())
grads = [torch.randn_like(b) * 1e-3 for b in bufs]
m = [torch.zeros_like(b) for b in bufs]
v = [torch.zeros_like(b) for b in bufs]
steps = [torch.zeros((), dtype=torch.float32, device=dev) for _ in bufs]
def stacked():
    torch._foreach_add_(steps, 1)
    torch._fused_adam_(bufs, grads, m, v, [], steps, lr=1e-4, beta1=0.9, beta2=0.999, weight_decay=5e-4, eps=1e-8,
                       amsgrad=False, maximize=False)
print(f"torch._fused_adam_ over the {len(bufs)} stacked buffers: {gpu_time(stacked):.1f} us per step")
