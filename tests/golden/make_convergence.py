#!/usr/bin/env python3
"""Convergence-length parity fixture: the reference MODEL under the reference's whole per-run protocol
(main_disentangled.py:131-224: `for run in range(args.run)` -> fresh split, fresh model, Adam, validation AUC every
epoch from the pre-step forward, best weights after the step, patience, test AUC with the best weights, mean / std over
the runs) at the chameleon recipe of /root/reference/hyperparameters_setting:2 (beta 0.7, t 1, K 5, nhid 512, d 32,
lr 1e-4, m 5) — with an epoch cap and a patience small enough for the early stop to FIRE inside the cap (the
reference's 2000 / 200 would be hours of dense CPU epochs per seed).

Run (this container only; needs /root/reference):  python tests/golden/make_convergence.py [seed ...]

The dataset arrays are those of tests/golden/real_chameleon.npz (make_real_data.py); DL_CONV_TAG=fast: the same at lr 1e-3;
DL_CONV_TAG=cora: Cora (real_cora.npz) at hyperparameters_setting:11 -> conv_cora.npz.  Writes
tests/golden/conv_chameleon.npz: per seed the per-epoch loss and validation AUC, the epoch the loop stopped at, the
epoch of the best validation AUC, the test AUC — outputs of the reference's model.py on CPU, nothing else.
"""
import json
import os
import sys
import time
from copy import deepcopy

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
RECIPE = dict(K=5, d=32, nhid=512, beta=0.7, t=1, m=5, lr=1e-4, weight_decay=5e-4)     # hyperparameters_setting:2
EPOCHS, PATIENCE = 400, 20
SEEDS = (21, 22, 23)
# At that recipe the validation AUC of the reference still improves at (nearly) every one of 400 epochs — the early stop
# never fires (it would take the reference's 2000 epochs).  A second set, tag "fast": the same recipe at the learning rate
# of hyperparameters_setting:10-13 (1e-3), where the validation AUC peaks and decays inside the cap: the patience FIRES.
# A third set, tag "cora": the Planetoid graph of BASELINE.json's configs[0] at the recipe of hyperparameters_setting:11
# (beta 0.6, K 10, nhid 256, d 64, lr 1e-3, m 5; binary features, NOT standardised — main_disentangled.py:117-123).
TAG = os.environ.get("DL_CONV_TAG", "")
DATASET = "chameleon"
if TAG == "fast":
    RECIPE = dict(RECIPE, lr=1e-3)
    EPOCHS, PATIENCE = 400, 20
elif TAG == "cora":
    DATASET = "cora"
    RECIPE = dict(K=10, d=64, nhid=256, beta=0.6, t=1, m=5, lr=1e-3, weight_decay=5e-4)
    EPOCHS, PATIENCE = 400, 20


def dense(u, v, n):
    a = np.zeros((n, n), dtype=np.float32)
    a[u, v] = 1.0
    return a


def summed(u, v, n):
    a = np.zeros((n, n), dtype=np.float32)
    np.add.at(a, (u, v), 1.0)
    return a


def one_run(seed, ref_model, feats, edges, log):
    """One iteration of the reference's `for run in range(args.run)` body (main_disentangled.py:131-221); the split and
    the initial weights come from `seed` (the reference is unseeded)."""
    from sklearn.metrics import roc_auc_score
    from disenlink_amd.datasets import standardise_rows
    from disenlink_amd.splits import make_link_split
    r = RECIPE
    n = feats.shape[0]
    x = standardise_rows(feats) if DATASET == "chameleon" else feats     # main_disentangled.py:97-101 / :117-123
    split = make_link_split(edges[:, 0], edges[:, 1], n, m=r["m"], seed=seed, keep_raw=True)
    ori = dense(edges[:, 0], edges[:, 1], n)
    adj = dense(split.train_src, split.train_dst, n)
    adj_sym = ((adj + adj.T) != 0).astype(np.float32)
    masks = {"pos": summed(split.train_src, split.train_dst, n) == 1, "neg": summed(*split.raw["neg_train"], n) == 1,
             "val": dense(*split.raw["val"], n) == 1, "test": dense(*split.raw["test"], n) == 1}
    for key, ps in (("pos", split.pos_train), ("neg", split.neg_train), ("val", split.val), ("test", split.test)):
        assert np.array_equal(np.stack(np.nonzero(masks[key])), np.stack([ps.u, ps.v])), key
    torch.manual_seed(seed)
    model = ref_model.Disentangle(feats.shape[1], r["nhid"], r["d"], nfactor=r["K"], beta=r["beta"], t=r["t"])
    opt = torch.optim.Adam(model.parameters(), lr=r["lr"], weight_decay=r["weight_decay"])
    xt, at, ot = torch.from_numpy(x), torch.from_numpy(adj_sym), torch.from_numpy(ori)
    mk = {k: torch.from_numpy(v) for k, v in masks.items()}
    losses, aucs, best, kept, stale, best_epoch = [], [], 0.0, None, 0, -1
    for ep in range(EPOCHS):
        t0 = time.perf_counter()
        _emb, pred = model(xt, at)
        loss = (F.binary_cross_entropy(pred[mk["pos"]].unsqueeze(0), ot[mk["pos"]].unsqueeze(0))
                + F.binary_cross_entropy(pred[mk["neg"]].unsqueeze(0), ot[mk["neg"]].unsqueeze(0)) / r["m"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        auc = roc_auc_score(ot[mk["val"]].numpy(), pred[mk["val"]].detach().numpy())
        losses.append(loss.item())
        aucs.append(auc)
        if auc > best:                                           # :206-213
            stale, best, best_epoch, kept = 0, auc, ep, deepcopy(model.state_dict())
        else:
            stale += 1
        log(f"seed {seed} epoch {ep}: loss {loss.item():.6f} val auc {auc:.6f} best {best:.6f}@{best_epoch} "
            f"({time.perf_counter() - t0:.1f} s)")
        if stale > PATIENCE:
            break
    model.load_state_dict(kept)
    _emb, pred = model(xt, at)
    test_auc = roc_auc_score(ot[mk["test"]].numpy(), pred[mk["test"]].detach().numpy(), average="weighted")
    counts = (int(split.pos_train.u.size), int(split.neg_train.u.size), int(split.val.u.size), int(split.test.u.size))
    return dict(losses=np.array(losses), val_aucs=np.array(aucs), epochs_run=len(losses), best_epoch=best_epoch,
                best_val_auc=best, test_auc=float(test_auc), counts=counts)


def main():
    sys.path.insert(0, REF)
    import model as ref_model                                    # the reference's model.py
    if DATASET == "cora":
        from disenlink_amd.datasets import load_planetoid
        ds = load_planetoid(os.path.join(REF, "data/cora/raw"), "cora")
        feats, edges = ds.x, np.stack([ds.src, ds.dst], axis=1)
    else:
        raw = np.load(os.path.join(REF, "data_pre_false/chameleon/raw/chameleon.npz"), allow_pickle=True)
        feats, edges = np.asarray(raw["features"], np.float32), np.asarray(raw["edges"], np.int64)
    seeds = [int(s) for s in sys.argv[1:]] or list(SEEDS)
    out = {}
    log = lambda s: print(s, flush=True)
    for seed in seeds:
        part = os.path.join(HERE, f"_conv{TAG}_part_{seed}.npz")      # per-seed part files: seeds can be made in parallel processes
        res = one_run(seed, ref_model, feats, edges, log)
        np.savez_compressed(part, **{k: np.asarray(v) for k, v in res.items()})
        log(f"seed {seed}: stopped after {res['epochs_run']} epochs, best {res['best_val_auc']:.6f} at epoch "
            f"{res['best_epoch']}, test auc {res['test_auc']:.6f}")
    # merge every part present
    prefix = f"_conv{TAG}_part_"
    parts = sorted(p for p in os.listdir(HERE) if p.startswith(prefix))
    all_seeds = [int(p[len(prefix):-4]) for p in parts]
    if sorted(all_seeds) == sorted(SEEDS):
        for s in SEEDS:
            with np.load(os.path.join(HERE, f"{prefix}{s}.npz")) as g:
                for k in g.files:
                    out[f"s{s}_{k}"] = g[k]
        meta = dict(RECIPE, epochs=EPOCHS, patience=PATIENCE, seeds=list(SEEDS), dataset=DATASET,
                    recipe="hyperparameters_setting:11" if TAG == "cora" else
                    "hyperparameters_setting:2" + (" at lr 1e-3 (the learning rate of :10-13)" if TAG == "fast" else ""))
        tests = np.array([float(out[f"s{s}_test_auc"]) for s in SEEDS])
        meta["test_auc_mean"], meta["test_auc_std"] = float(tests.mean()), float(tests.std())   # np.std, as :222-223
        path = os.path.join(HERE, "conv_cora.npz" if TAG == "cora" else f"conv_chameleon{'_' + TAG if TAG else ''}.npz")
        np.savez_compressed(path, meta=np.array(json.dumps(meta)), **out)
        for p in parts:
            os.remove(os.path.join(HERE, p))
        log(f"-> {path} {os.path.getsize(path)} bytes; test AUC {tests.mean():.6f} +- {tests.std():.6f}")


if __name__ == "__main__":
    main()
