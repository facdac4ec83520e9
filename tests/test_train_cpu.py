"""Caller-side pieces on CPU: on-device AUC vs the sklearn golden vectors, pair-list loss vs the
dense masked loss of the reference, the split builder's contract, and the train loop's schedule
(run here with an oracle-backed module standing in for the HIP module)."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN_DIR, golden_case_names, load_golden
from oracle import dense_ref, metrics_ref


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN_DIR, "auc_*.npz"))))
def test_device_auc_matches_sklearn_vectors(path):
    from disenlink_amd.metrics import auc_tie_avg
    g = np.load(path)
    got = float(auc_tie_avg(torch.from_numpy(g["y"]), torch.from_numpy(g["score"])))
    assert abs(got - float(g["auc"])) < 1e-12
    assert abs(got - metrics_ref.auc_tie_avg(g["y"], g["score"])) < 1e-12
    with pytest.raises(ValueError):
        auc_tie_avg(torch.ones(4), torch.rand(4))


def test_auc_plan_matches_sklearn_vectors_and_the_rank_form():
    """AucPlan (negatives sorted, positives located by binary search, integer counts) against the sklearn golden
    vectors and against the rank-based auc_tie_avg on random scores with heavy ties."""
    import glob
    from disenlink_amd.metrics import AucPlan, auc_tie_avg
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "auc_*.npz"))):
        g = np.load(path)
        assert abs(float(AucPlan(torch.from_numpy(g["y"])).auc(torch.from_numpy(g["score"]))) - float(g["auc"])) <= 1e-12
    rng = np.random.default_rng(0)
    for n, levels in ((1000, 7), (5000, 100000), (300, 2)):
        y = torch.from_numpy((rng.random(n) < 0.3).astype(np.float32))
        sc = torch.from_numpy((rng.integers(0, levels, n) / levels).astype(np.float32))
        sc[rng.random(n) < 0.2] = 1.0
        plan = AucPlan(y)
        assert abs(float(plan.auc(sc)) - float(auc_tie_avg(y, sc))) <= 1e-12
        assert abs(float(plan.auc(sc * 0 + 0.5)) - 0.5) <= 1e-15                     # all tied
    assert torch.isnan(AucPlan(torch.ones(5)).auc(torch.rand(5)))                   # one class only


@pytest.mark.parametrize("name", golden_case_names())
def test_pair_loss_equals_the_references_masked_loss(name):
    from disenlink_amd.metrics import pair_bce_loss
    g = load_golden(name)
    P = torch.from_numpy(g["link_pred"])
    pu, pv = np.nonzero(g["pos_mask"] == 1)              # the caller takes mask == 1: pairs that occur exactly once
    nu, nv = np.nonzero(g["neg_mask"] == 1)
    ori = g["ori_adj"]
    loss = pair_bce_loss(P[pu, pv], torch.from_numpy(ori[pu, pv]), P[nu, nv], torch.from_numpy(ori[nu, nv]),
                         g["meta"]["m"])
    assert abs(float(loss) - float(g["loss"])) <= 1e-6 * max(1.0, abs(float(g["loss"])))


def test_split_builder_contract():
    from disenlink_amd.data import synthetic_graph
    from disenlink_amd.splits import make_link_split
    sg = synthetic_graph("chameleon", seed=1, scale=0.2)
    N, E = sg.n_nodes, sg.src.size
    sp = make_link_split(sg.src, sg.dst, N, m=3, seed=5)
    n_tr = int(0.85 * E)
    assert sp.train_src.size == n_tr                                     # 85 % of the edge ROWS
    keys = set((sg.src * N + sg.dst).tolist())
    for ps, want in ((sp.pos_train, 1.0), (sp.neg_train, 0.0)):
        k = ps.u * N + ps.v
        assert np.unique(k).size == k.size and (np.diff(k) > 0).all()    # unique, row-major order
        assert (ps.label == want).all()
        assert all(((int(x) in keys) == bool(want)) for x in k.tolist())
    # m draws per train row; pairs drawn more than once drop out (summed masks, == 1): a small dense graph loses many
    assert 1.5 * n_tr < sp.neg_train.u.size <= 3 * n_tr
    # every negative shares its source with a positive edge row
    assert set(sp.neg_train.u.tolist()) <= set(sg.src.tolist())
    # val / test: positives and negatives together, labels from the directed edge rows
    for ps in (sp.val, sp.test):
        k = ps.u * N + ps.v
        assert np.array_equal(ps.label, np.array([float(int(x) in keys) for x in k.tolist()], np.float32))
        assert 0 < ps.label.mean() < 0.5
    sp2 = make_link_split(sg.src, sg.dst, N, m=3, seed=5)
    assert np.array_equal(sp.test.u, sp2.test.u) and np.array_equal(sp.neg_train.v, sp2.neg_train.v)


def test_train_pair_sets_are_the_callers_summed_masks():
    """main_disentangled.py:176-179 builds pos_train_adj / neg_train_adj with sparse_coo(...).to_dense() and never
    binarises them: duplicate index pairs add up, and the loss takes a_pred[mask == 1] (:195) — the pairs that occur
    EXACTLY once.  all_val_adj / all_test_adj are binarised (:187-190): every distinct pair.  The pair lists must be
    those mask selections, in mask (row-major) order, on edge rows with duplicates (as chameleon.npz has) and with
    negatives colliding across the m draws (hub rows)."""
    from disenlink_amd.splits import make_link_split
    rng = np.random.default_rng(3)
    N, E = 40, 400
    src, dst = rng.integers(0, N, E), rng.integers(0, N, E)
    src[:120] = 0                                                         # a hub: many colliding negatives
    src, dst = np.r_[src, src[:60]], np.r_[dst, dst[:60]]                 # 60 repeated rows
    sp = make_link_split(src, dst, N, m=5, seed=1, keep_raw=True)

    def summed(u, v):
        a = np.zeros((N, N))
        np.add.at(a, (u, v), 1.0)
        return a
    pos, neg = summed(sp.train_src, sp.train_dst), summed(*sp.raw["neg_train"])
    assert (pos > 1).any() and (neg > 1).any()                            # the case is there
    for ps, mask in ((sp.pos_train, pos == 1), (sp.neg_train, neg == 1), (sp.val, summed(*sp.raw["val"]) >= 1),
                     (sp.test, summed(*sp.raw["test"]) >= 1)):
        assert np.array_equal(np.stack([ps.u, ps.v]), np.stack(np.nonzero(mask)))
    assert sp.pos_train.u.size < np.unique(sp.train_src * N + sp.train_dst).size      # repeated rows dropped, not kept once


class OraclePairModule(nn.Module):
    """TEST stand-in with the drop-in module's parameters, computing through oracle/dense_ref on CPU."""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def forward_pairs(self, x, graph, pairs):
        N = graph.n_nodes
        adj = torch.zeros(N, N)
        src = torch.repeat_interleave(torch.arange(N), (graph.rowptr[1:] - graph.rowptr[:-1]).long())
        adj[src, graph.col.long()] = 1
        sd = dict(self.inner.named_parameters())
        emb, P = dense_ref.forward(x, adj, sd, self.inner.beta, self.inner.temperature)
        return emb, P[pairs.pu.long(), pairs.pv.long()]

    def state_dict(self, *a, **k):
        return self.inner.state_dict(*a, **k)

    def load_state_dict(self, sd, *a, **k):
        return self.inner.load_state_dict(sd, *a, **k)


def test_train_loop_schedule_on_cpu():
    """Loss falls, validation AUC is taken from the pre-step forward, early stopping and best-weight
    restore follow main_disentangled.py:191-224."""
    from disenlink_amd.data import synthetic_graph
    from disenlink_amd.model import Disentangle
    from disenlink_amd.splits import make_link_split
    from disenlink_amd.train import prepare_run, run_link_prediction
    sg = synthetic_graph("chameleon", seed=2, scale=0.06)
    split = make_link_split(sg.src, sg.dst, sg.n_nodes, m=2, seed=0)
    run = prepare_run(split, "cpu")
    torch.manual_seed(0)
    inner = Disentangle(16, 8, 8, nfactor=3, beta=0.7, t=1)
    model = OraclePairModule(inner)
    x = torch.from_numpy(sg.features()[:, :16].copy())
    # epoch-0 validation AUC must equal the AUC of the untrained weights
    with torch.no_grad():
        _e, p0 = model.forward_pairs(x, run.graph, run.train_val_pairs)
    from disenlink_amd.metrics import auc_tie_avg
    auc0 = float(auc_tie_avg(run.label_val, p0[run.n_pos + run.n_neg:]))
    res = run_link_prediction(model, x, run, epochs=25, lr=5e-3, patience=3)
    assert abs(res.val_aucs[0] - auc0) < 1e-12
    assert res.losses[-1] < res.losses[0]
    assert res.epochs_run <= 25 and 0.0 <= res.test_auc <= 1.0
    assert res.best_val_auc == max(res.val_aucs)
    # patience: stops `patience`+1 epochs after the last improvement
    last_best = int(np.argmax(res.val_aucs))
    if res.epochs_run < 25:
        assert res.epochs_run == last_best + 1 + 3 + 1
