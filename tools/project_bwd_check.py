"""Per-output error of dl_project_bwd against fp64 for a list of shapes, and where the masked hidden gradient
in the workspace differs (debugging aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disenlink_amd import ops
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(5201, 128, 8, 512, 64)]
for N, F, K, nhid, d in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, F, generator=g); dZ = torch.randn(N, K, d, generator=g)
    W1 = torch.randn(K, nhid, F, generator=g) / F ** 0.5; b1 = torch.randn(K, nhid, generator=g) * 0.1
    W2 = torch.randn(K, d, nhid, generator=g) / nhid ** 0.5
    X, G = x.double(), dZ.double()
    pre = torch.einsum("nf,khf->nkh", X, W1.double()) + b1.double()
    dh = torch.einsum("nkd,kdh->nkh", G, W2.double()) * (pre > 0)
    ref = (torch.einsum("nkh,nf->khf", dh, X), dh.sum(0), torch.einsum("nkd,nkh->kdh", G, pre.clamp_min(0)), G.sum(0))
    out = ops.project_bwd(*[v.cuda() for v in (x, W1, b1, W2, dZ)])
    torch.cuda.synchronize()
    ws = next(iter(ops._ws.buf.values()))
    got = ws[: N * K * nhid * 4].view(torch.float32).view(N, K, nhid).cpu().double()
    bad = ((got - dh).abs() > 1e-4 * dh.abs().max()) & (pre.abs() > 1e-4)
    msg = ""
    if bad.any():
        idx = bad.nonzero()
        msg = (f" dhid bad {int(bad.sum())}: n {int(idx[:,0].min())}..{int(idx[:,0].max())} k {sorted(set(idx[:,1].tolist()))} "
               f"h {int(idx[:,2].min())}..{int(idx[:,2].max())} nodes%128 {sorted(set((idx[:,0] % 128).tolist()))[:12]} first {idx[0].tolist()} "
               f"got {float(got[tuple(idx[0])]):.4f} want {float(dh[tuple(idx[0])]):.4f}")
    print((N, F, K, nhid, d), " ".join(f"{n}:{float((o.cpu().double() - r).abs().max() / r.abs().max()):.1e}"
                                       for n, o, r in zip(("dW1", "db1", "dW2", "db2"), out, ref)) + msg, flush=True)
