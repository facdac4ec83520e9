#!/usr/bin/env python3
"""Generate golden vectors by importing the reference model (CPU, this container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)

The reference has no tests or fixtures of its own, so the oracle in
``oracle/`` is pinned by outputs of the reference itself: this script imports
``/root/reference/model.py`` (``Disentangle``, model.py:91-114), runs it on tiny seeded
inputs and stores inputs + outputs as ``tests/golden/case_*.npz``.  Only data is stored:
inputs, the state_dict, and the tensors the reference returned / autograd produced.

Per case:
  x, adj                       inputs of Disentangle.forward (model.py:105)
  sd__<key>                    the full state_dict (key names are part of the boundary)
  emb, link_pred               forward outputs (model.py:114)
  alpha0, att                  Disentangle_layer internals returned at model.py:77
  p, a, s                      derived per-pair factor id / weight and per-node normaliser,
                               computed with the same expressions as model.py:59-72
  ori_adj, pos_mask, neg_mask  a fixed train-positive / train-negative mask pair + labels; the masks are SUMMED index
                               lists like the caller's (duplicates give 2, 3, ...): the loss takes the entries == 1
  loss, grad__<key>            loss of main_disentangled.py:195 and all parameter gradients
Plus auc_*.npz: scores/labels/roc_auc_score value (sklearn) incl. heavy ties at 1.0.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def make_graph(rng, n, p_edge, isolated=(), self_loops=(), hub=None):
    """Directed edge rows (may contain duplicates), like edge_index in the reference."""
    rows = []
    for i in range(n):
        for j in range(n):
            if i != j and rng.random() < p_edge:
                rows.append((i, j))
    if hub is not None:
        rows += [(hub, j) for j in range(n) if j != hub]
    rows += [(v, v) for v in self_loops]
    rows = [(i, j) for (i, j) in rows if i not in isolated and j not in isolated]
    # a few duplicate rows, as in chameleon.npz
    dup = [rows[k] for k in rng.integers(0, len(rows), size=max(1, len(rows) // 10))]
    rows = rows + dup
    rows = np.array(rows, dtype=np.int64)
    rng.shuffle(rows, axis=0)
    return rows


def dense01(rows, n):
    a = np.zeros((n, n), dtype=np.float32)
    a[rows[:, 0], rows[:, 1]] = 1.0
    return a


def summed(rows, n):
    """torch.sparse_coo_tensor(idx, ones).to_dense() as main_disentangled.py:176-179 builds the TRAIN masks pos_train_adj /
    neg_train_adj: duplicate index pairs ADD UP and the masks are never binarised, so `a_pred[mask == 1]` (:195) leaves
    out every pair that occurs more than once."""
    a = np.zeros((n, n), dtype=np.float32)
    np.add.at(a, (rows[:, 0], rows[:, 1]), 1.0)
    return a


CASES = [
    # name,      N,  F,  K, d, nhid, beta, t, x_scale, p_edge, isolated, self_loops, hub, m
    ("tiny_k1",   7,  5, 1,  3,   1, 0.5, 1, 1.0, 0.4, (3,), (), None, 2),
    ("tiny_k3",  12,  6, 3,  3,   1, 0.7, 1, 1.5, 0.3, (0,), (5,), None, 2),
    ("k4_d8",    30, 10, 4,  8,   6, 0.9, 1, 1.0, 0.15, (7,), (1, 2), 4, 3),
    ("k4_d32",   40, 24, 4, 32,  16, 0.6, 1, 0.5, 0.12, (), (3,), None, 5),
    ("k8_d8_t2", 48, 12, 8,  8,   1, 0.5, 2, 2.0, 0.10, (11, 12), (), 0, 5),
    ("k8_d32",   64, 16, 8, 32,  32, 0.7, 1, 0.45, 0.08, (9,), (10, 20), 5, 5),
    ("k3_d8_hub", 80, 8, 3,  8,   4, 0.9, 2, 1.0, 0.05, (1,), (2,), 3, 1),
    ("k5_d64",   56, 20, 5, 64,  24, 0.8, 1, 0.3, 0.10, (), (0,), None, 5),
    ("k8_d64",   72, 32, 8, 64,  48, 0.5, 1, 0.25, 0.09, (4,), (8, 9), 6, 5),
    # round 5: the factor shape of BASELINE configs[4] (Penn94: K = 16, d = 128) — the wide kernels against the reference itself
    ("k16_d128", 44, 24, 16, 128, 128, 0.6, 1, 0.3, 0.12, (5,), (7,), 2, 5),
]


def run_case(model_mod, spec, seed):
    (name, n, f, k, d, nhid, beta, t, xs, p_edge, iso, loops, hub, m) = spec
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    rows = make_graph(rng, n, p_edge, iso, loops, hub)
    e_rows = rows.shape[0]
    perm = rng.permutation(e_rows)
    n_tr = int(round(0.85 * e_rows))
    tr = rows[perm[:n_tr]]
    ori_adj = dense01(rows, n)
    adj = dense01(tr, n)
    adj_sym = ((adj + adj.T) != 0).astype(np.float32)
    # negatives: for each train row (i, j) a node q with (i, q) not an edge of ori_adj, m draws
    negs = []
    for _ in range(m):
        for (i, _j) in tr:
            cand = np.flatnonzero(ori_adj[i] == 0)
            cand = cand[cand != i] if (cand != i).any() else cand
            negs.append((i, int(rng.choice(cand))))
    negs = np.array(negs, dtype=np.int64)
    pos_mask = summed(tr, n)
    neg_mask = summed(negs, n)

    x = (rng.standard_normal((n, f)) * xs).astype(np.float32)
    model = model_mod.Disentangle(f, nhid, d, nfactor=k, beta=beta, t=t)
    xt = torch.from_numpy(x)
    at = torch.from_numpy(adj_sym)
    Z = [fac(xt) for fac in model.factors]                      # model.py:106
    h_list, alpha0, att = model.disentangle_layer1(Z, at)       # model.py:107
    emb, link_pred = model(xt, at)                              # model.py:105-114
    assert torch.equal(emb, torch.cat(h_list, dim=1))

    # derived routing quantities, same expressions as model.py:59-72
    alpha = alpha0 / torch.sum(alpha0, dim=0)
    p = torch.argmax(alpha, dim=0)
    a = torch.gather(alpha, 0, p.unsqueeze(0)).squeeze(0) * at
    s = torch.stack([((p == kk).float() * at * alpha[kk]).sum(dim=1) for kk in range(k)], dim=1)
    s[s == 0] = 1

    ori_t = torch.from_numpy(ori_adj)
    pm = torch.from_numpy(pos_mask)
    nm = torch.from_numpy(neg_mask)
    loss = (F.binary_cross_entropy(link_pred[pm == 1].unsqueeze(0), ori_t[pm == 1].unsqueeze(0))
            + F.binary_cross_entropy(link_pred[nm == 1].unsqueeze(0), ori_t[nm == 1].unsqueeze(0)) / m)
    model.zero_grad()
    loss.backward()                                             # main_disentangled.py:195-198

    out = dict(
        x=x, adj=adj_sym, emb=emb.detach().numpy(), link_pred=link_pred.detach().numpy(),
        alpha0=alpha0.detach().numpy(), att=torch.stack(att, 0).detach().numpy(),
        p=p.numpy().astype(np.int32), a=a.detach().numpy(), s=s.detach().numpy(),
        ori_adj=ori_adj, pos_mask=pos_mask, neg_mask=neg_mask,
        loss=np.float32(loss.item()),
        meta=np.array(json.dumps(dict(name=name, N=n, F=f, K=k, d=d, nhid=nhid, beta=beta, t=t, m=m, seed=seed))),
    )
    for key, v in model.state_dict().items():
        out["sd__" + key] = v.detach().numpy().copy()
    for key, prm in model.named_parameters():
        out["grad__" + key] = prm.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"case_{name}.npz"), **out)
    nsat = int((link_pred == 1).sum())
    gmax = max(float(prm.grad.abs().max()) for prm in model.parameters())
    print(f"{name}: N={n} K={k} d={d} nnz={int(adj_sym.sum())} loss={loss.item():.6f} "
          f"saturated={nsat}/{n * n} max|grad|={gmax:.3e}")


TRAJ = [
    # name,       N,  F, K,  d, nhid, beta, t, x_scale, p_edge, m, epochs, lr
    ("traj_k4",  90, 12, 4, 32,  16, 0.7, 1, 0.4, 0.06, 3, 8, 1e-2),
    ("traj_k8", 120, 16, 8,  8,   1, 0.5, 1, 0.8, 0.05, 5, 8, 5e-3),
    # round 5: the benchmark's factor shape (K=8, d=64) at temperature 2 — the `/ t` of model.py:56 inside the one-pass
    # training scorer (its T1 = false instantiation) pinned by a training trajectory of the reference
    ("traj_k8_d64_t2", 140, 16, 8, 64, 32, 0.6, 2, 0.5, 0.05, 5, 8, 5e-3),
    ("traj_k16_d128", 100, 16, 16, 128, 128, 0.6, 1, 0.3, 0.06, 5, 6, 5e-3),
]


def run_trajectory(model_mod, spec, seed):
    """A short run of the reference MODEL under the schedule of main_disentangled.py:150,191-219 (Adam with
    weight decay 5e-4, full-batch epochs, validation AUC from the pre-step forward, best weights kept, test AUC
    with them): pins loss / AUC trajectories, i.e. forward + backward + optimiser together."""
    from copy import deepcopy
    from sklearn.metrics import roc_auc_score
    (name, n, f, k, d, nhid, beta, t, xs, p_edge, m, epochs, lr) = spec
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    rows = make_graph(rng, n, p_edge)
    perm = rng.permutation(rows.shape[0])
    n_tr, n_va = int(round(0.85 * rows.shape[0])), int(round(0.05 * rows.shape[0]))
    tr, va, te = rows[perm[:n_tr]], rows[perm[n_tr:n_tr + n_va]], rows[perm[n_tr + n_va:]]
    ori = dense01(rows, n)
    adj_sym = ((dense01(tr, n) + dense01(tr, n).T) != 0).astype(np.float32)

    def negatives(part):
        out = []
        for _ in range(m):
            for (i, _j) in part:
                cand = np.flatnonzero(ori[i] == 0)
                cand = cand[cand != i] if (cand != i).any() else cand
                out.append((i, int(rng.choice(cand))))
        return np.array(out, dtype=np.int64)

    masks = {"pos_train": summed(tr, n), "neg_train": summed(negatives(tr), n),
             "val": np.minimum(dense01(va, n) + dense01(negatives(va), n), 1),
             "test": np.minimum(dense01(te, n) + dense01(negatives(te), n), 1)}
    x = (rng.standard_normal((n, f)) * xs).astype(np.float32)
    model = model_mod.Disentangle(f, nhid, d, nfactor=k, beta=beta, t=t)
    init = {key: v.detach().numpy().copy() for key, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=5e-4)
    xt, at, ot = torch.from_numpy(x), torch.from_numpy(adj_sym), torch.from_numpy(ori)
    mk = {key: torch.from_numpy(v) == 1 for key, v in masks.items()}
    losses, aucs, best, kept = [], [], 0.0, None
    for _ep in range(epochs):
        _emb, pred = model(xt, at)
        loss = (F.binary_cross_entropy(pred[mk["pos_train"]].unsqueeze(0), ot[mk["pos_train"]].unsqueeze(0))
                + F.binary_cross_entropy(pred[mk["neg_train"]].unsqueeze(0), ot[mk["neg_train"]].unsqueeze(0)) / m)
        opt.zero_grad()
        loss.backward()
        opt.step()
        auc = roc_auc_score(ot[mk["val"]].numpy(), pred[mk["val"]].detach().numpy())
        losses.append(loss.item())
        aucs.append(auc)
        if auc > best:
            best, kept = auc, deepcopy(model.state_dict())
    model.load_state_dict(kept)
    _emb, pred = model(xt, at)
    test_auc = roc_auc_score(ot[mk["test"]].numpy(), pred[mk["test"]].detach().numpy(), average="weighted")
    out = dict(x=x, adj=adj_sym, ori_adj=ori, losses=np.array(losses, dtype=np.float64),
               val_aucs=np.array(aucs, dtype=np.float64), test_auc=np.float64(test_auc),
               meta=np.array(json.dumps(dict(name=name, N=n, F=f, K=k, d=d, nhid=nhid, beta=beta, t=t, m=m, seed=seed,
                                             epochs=epochs, lr=lr))))
    for key, v in masks.items():
        out["mask__" + key] = v.astype(np.float32)
    for key, v in init.items():
        out["sd__" + key] = v
    for key, v in kept.items():
        out["best__" + key] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}: losses {losses[0]:.5f} -> {losses[-1]:.5f}, val auc {aucs[0]:.4f} -> {max(aucs):.4f}, test auc {test_auc:.4f}")


ADAM = [("k4_d8", 12, 1e-2), ("k5_d64", 12, 1e-2)]      # (committed case, Adam steps, lr)


def run_adam_steps(model_mod, case, steps, lr):
    """Fixed masks, several optimiser steps (main_disentangled.py:150,194-199 around the reference model, from a
    committed case's inputs and weights).  Pins what one forward / backward cannot: the set of entries of
    d loss / d link_pred that are non-zero MOVES from step to step under fixed masks — a saturated positive (p == 1.0 in
    fp32, y == 1) has exactly zero BCE gradient until the weights move it out of saturation (SURVEY.md §0 finding 4) — so
    a backward keyed on the gradient's non-zero set instead of the masks goes wrong from the second step on.  Stored:
    per-step loss, per-step count of non-zero gradient entries (nnz_grad) and of masked entries (n_masked), the
    gradients of the LAST step and the weights after it."""
    g = dict(np.load(os.path.join(OUT, f"case_{case}.npz"), allow_pickle=False))
    m = json.loads(str(g["meta"]))
    model = model_mod.Disentangle(m["F"], m["nhid"], m["d"], nfactor=m["K"], beta=m["beta"], t=m["t"])
    model.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")})
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=5e-4)
    x, adj, ori = (torch.from_numpy(g[k]) for k in ("x", "adj", "ori_adj"))
    pm, nm = torch.from_numpy(g["pos_mask"]) == 1, torch.from_numpy(g["neg_mask"]) == 1
    losses, nnz = [], []
    for _ in range(steps):
        _emb, a_pred = model(x, adj)
        a_pred.retain_grad()
        loss = (F.binary_cross_entropy(a_pred[pm].unsqueeze(0), ori[pm].unsqueeze(0))
                + F.binary_cross_entropy(a_pred[nm].unsqueeze(0), ori[nm].unsqueeze(0)) / m["m"])
        opt.zero_grad()
        loss.backward()
        losses.append(loss.item())
        nnz.append(int(torch.count_nonzero(a_pred.grad)))
        last = {k: prm.grad.detach().numpy().copy() for k, prm in model.named_parameters()}
        opt.step()
    out = dict(losses=np.array(losses, np.float64), nnz_grad=np.array(nnz, np.int64),
               n_masked=np.int64(int((pm | nm).sum())),
               meta=np.array(json.dumps(dict(case=case, steps=steps, lr=lr, weight_decay=5e-4))))
    for k, v in last.items():
        out["grad__" + k] = v
    for k, v in model.state_dict().items():
        out["sd__" + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, f"adam_{case}.npz"), **out)
    print(f"adam_{case}: losses {losses[0]:.6f} -> {losses[-1]:.6f}; non-zero gradient entries per step {nnz} of {int(out['n_masked'])} masked")


def auc_cases():
    from sklearn.metrics import roc_auc_score
    rng = np.random.default_rng(7)
    specs = [("ties_at_one", 500, 0.6), ("no_ties", 300, 0.0), ("all_ties_half", 200, 0.95)]
    for name, n, tie_frac in specs:
        y = (rng.random(n) < 0.3).astype(np.float32)
        sc = rng.random(n).astype(np.float32) * 0.5 + y * 0.3
        tie = rng.random(n) < tie_frac
        sc[tie] = 1.0
        if name == "all_ties_half":
            sc[~tie] = np.float32(0.25)
        auc = roc_auc_score(y, sc)
        np.savez_compressed(os.path.join(OUT, f"auc_{name}.npz"), y=y, score=sc, auc=np.float64(auc))
        print(f"auc_{name}: {auc:.10f}")


def main():
    sys.path.insert(0, REF)
    import model as model_mod  # the reference's model.py
    torch.set_num_threads(1)
    only_traj = "--trajectories-only" in sys.argv or "--adam-only" in sys.argv   # the case_* / auc_* files are already committed
    only_case = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only-case=")]      # --only-case=k16_d128: one case
    if only_case:
        for idx, spec in enumerate(CASES):
            if spec[0] in only_case:
                run_case(model_mod, spec, seed=100 + idx)
        return
    if not only_traj:
        for idx, spec in enumerate(CASES):
            run_case(model_mod, spec, seed=100 + idx)
        auc_cases()
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]     # --only=traj_k8_d64_t2: one trajectory
    if "--adam-only" not in sys.argv:
        for idx, spec in enumerate(TRAJ):
            if not only or spec[0] in only:
                run_trajectory(model_mod, spec, seed=300 + idx)
    if only:
        return
    for case, steps, lr in ADAM:
        run_adam_steps(model_mod, case, steps, lr)


if __name__ == "__main__":
    main()
