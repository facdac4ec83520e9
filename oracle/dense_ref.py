"""Dense CPU restatement of the reference forward (torch, autograd-capable).  TEST INFRASTRUCTURE.

Same arithmetic as the reference, which is dense ``[K,N,N]`` throughout; written here over
one stacked ``Z [K,N,d]`` tensor with batched ops.  Follows:

  project      model.py:13-15 (Factor), :24-27 (Factor2), fan-out :106
  route_aggregate   model.py:55-77 (Disentangle_layer.forward)
  score_allpairs    model.py:109-113
  forward      model.py:105-114
  bce_pair_loss     main_disentangled.py:195

This is also the timed "port" CPU baseline of bench.py (baseline A).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def n_factors(sd) -> int:
    k = 0
    while any(key.startswith(f"factor_{k}.") for key in sd):
        k += 1
    return k


def project(x: torch.Tensor, sd: dict) -> torch.Tensor:
    """Z[k] = MLP_k(x); returns [K,N,d].  model.py:13-15 / :24-27 / :106."""
    zs = []
    for k in range(n_factors(sd)):
        pre = f"factor_{k}."
        if pre + "mlp.weight" in sd:                      # nhid == 1: single Linear (model.py:94-95)
            z = F.linear(x, sd[pre + "mlp.weight"], sd[pre + "mlp.bias"])
        else:                                             # Linear -> ReLU -> Linear (model.py:96-97)
            hid = torch.relu(F.linear(x, sd[pre + "mlp1.weight"], sd[pre + "mlp1.bias"]))
            z = F.linear(hid, sd[pre + "mlp2.weight"], sd[pre + "mlp2.bias"])
        zs.append(z)
    return torch.stack(zs, dim=0)


def route_aggregate(Z: torch.Tensor, adj: torch.Tensor, beta: float, t: float):
    """model.py:55-77.  Z [K,N,d], adj [N,N] in {0,1}.

    Returns H [K,N,d], e [K,N,N] (un-normalised exp, model.py:57,77), att [K,N,N],
    p [N,N] (arg-max factor id, 0-based), s [N,K] (normaliser after zero->1).
    """
    K = Z.shape[0]
    e = torch.exp(torch.bmm(Z, Z.transpose(1, 2)) / t)            # :56-57
    alpha = e / e.sum(dim=0)                                      # :59-60 (no max-subtraction)
    p = torch.argmax(alpha, dim=0)                                # :61 first max wins
    routed = (p + 1) * adj                                        # :62
    H, att, s_all = [], [], []
    for k in range(K):
        a_k = (routed == k + 1).float() * alpha[k]                # :64-66, :70
        s_k = a_k.sum(dim=1)                                      # :71
        s_k = torch.where(s_k == 0, torch.ones_like(s_k), s_k)    # :72
        w_k = a_k / s_k                                           # :73 broadcasts over COLUMNS: a[i,j]/s[j]
        att.append(w_k)
        s_all.append(s_k)
        H.append(beta * Z[k] + (1 - beta) * (w_k @ Z[k]))         # :75
    return torch.stack(H, 0), e, torch.stack(att, 0), p, torch.stack(s_all, 1)


def score_allpairs(H: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """P[i,j] = sigmoid(sum_k (h_k[i].h_k[j]) * e[k,i,j]).  model.py:109-113."""
    q = torch.bmm(H, H.transpose(1, 2))
    return torch.sigmoid((q * e).sum(dim=0))


def forward(x, adj, sd, beta: float, t: float):
    """Disentangle.forward, model.py:105-114 -> (emb [N,K*d], link_pred [N,N])."""
    Z = project(x, sd)
    H, e, _att, _p, _s = route_aggregate(Z, adj, beta, t)
    P = score_allpairs(H, e)
    emb = torch.cat(list(H), dim=1)                               # :114, factor-major along dim 1
    return emb, P


def bce_pair_loss(P, ori_adj, pos_mask, neg_mask, m: int):
    """main_disentangled.py:195: BCE over masked entries (mean; log clamped at -100) + BCE(neg)/m."""
    lp = F.binary_cross_entropy(P[pos_mask == 1].unsqueeze(0), ori_adj[pos_mask == 1].unsqueeze(0))
    ln = F.binary_cross_entropy(P[neg_mask == 1].unsqueeze(0), ori_adj[neg_mask == 1].unsqueeze(0))
    return lp + ln / m
