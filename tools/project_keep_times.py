"""Projection forward+backward with the hidden layer recomputed vs kept (and the library GEMM form), kernel time.
usage: python tools/project_keep_times.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for N, F, K, nhid, d in [(5201, 128, 8, 512, 64), (5201, 269, 8, 512, 64), (5201, 512, 8, 512, 64), (5201, 2089, 8, 512, 64),
                         (41554, 128, 16, 512, 128), (500000, 269, 8, 512, 64)]:
    x, dZ = torch.randn(N, F, device=dev), torch.randn(N, K, d, device=dev)
    W1 = torch.randn(K, nhid, F, device=dev) / F ** 0.5; b1 = torch.randn(K, nhid, device=dev) * 0.1
    W2 = torch.randn(K, d, nhid, device=dev) / nhid ** 0.5; b2 = torch.zeros(K, d, device=dev)
    def recompute():
        ops.project_fwd(x, W1, b1, W2, b2); ops.project_bwd(x, W1, b1, W2, dZ)
    def keep():
        _Z, hid = ops.project_fwd(x, W1, b1, W2, b2, keep_hid=True); ops.project_bwd(x, W1, b1, W2, dZ, hid=hid)
    W1r = W1.clone().requires_grad_(True); b1r = b1.clone().requires_grad_(True); W2r = W2.clone().requires_grad_(True); b2r = b2.clone().requires_grad_(True)
    def library():
        hid = torch.relu(torch.nn.functional.linear(x, W1r.reshape(K * nhid, F), b1r.reshape(-1))).view(N, K, nhid)
        Z = torch.einsum("nkh,kdh->nkd", hid, W2r) + b2r
        torch.autograd.grad(Z, (W1r, b1r, W2r, b2r), dZ)
    tr, tk = timeit(recompute), timeit(keep)
    tl = timeit(library) if N * K * nhid * 4 < (6 << 30) else float("nan")
    print(f"N={N} F={F} K={K} nhid={nhid} d={d}: fwd+bwd recompute {tr*1e3:8.3f} ms  kept {tk*1e3:8.3f} ms  library autograd {tl*1e3:8.3f} ms", flush=True)
