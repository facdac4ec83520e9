import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from disenlink_amd.model import Disentangle
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2277
m = Disentangle(128, 512, 64, nfactor=8, beta=0.5, t=1, projection="mfma").cuda()
x = torch.randn(N, 128, device="cuda")
with torch.no_grad():
    for _ in range(5): m.project(x)
torch.cuda.synchronize()
