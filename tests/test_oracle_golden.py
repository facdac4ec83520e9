"""Pin the oracle (oracle/) against outputs of the reference itself (tests/golden/*.npz).

The golden files were produced by tests/golden/make_golden.py importing the reference's
model.py; nothing here reads /root/reference.
"""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import dense_ref, metrics_ref, sparse_ref


def _sd(g):
    return {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}


def _Z_nkd(g):
    Z = dense_ref.project(torch.from_numpy(g["x"]), _sd(g))          # [K,N,d]
    return np.ascontiguousarray(Z.permute(1, 0, 2).numpy())          # [N,K,d]


def test_dense_forward_matches_reference(golden):
    g, m = golden, golden["meta"]
    x, adj = torch.from_numpy(g["x"]), torch.from_numpy(g["adj"])
    emb, P = dense_ref.forward(x, adj, _sd(g), m["beta"], m["t"])
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(P.numpy(), g["link_pred"], rtol=1e-6, atol=1e-6)
    Z = dense_ref.project(x, _sd(g))
    H, e, att, p, s = dense_ref.route_aggregate(Z, adj, m["beta"], m["t"])
    np.testing.assert_allclose(e.numpy(), g["alpha0"], rtol=1e-6)
    np.testing.assert_allclose(att.numpy(), g["att"], rtol=1e-6, atol=1e-7)
    on_edge = g["adj"] == 1
    assert (p.numpy()[on_edge] == g["p"][on_edge]).all()
    np.testing.assert_allclose(s.numpy(), g["s"], rtol=1e-6)


def test_dense_loss_and_grads_match_reference(golden):
    g, m = golden, golden["meta"]
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(g).items()}
    x, adj = torch.from_numpy(g["x"]), torch.from_numpy(g["adj"])
    _emb, P = dense_ref.forward(x, adj, sd, m["beta"], m["t"])
    loss = dense_ref.bce_pair_loss(P, torch.from_numpy(g["ori_adj"]), torch.from_numpy(g["pos_mask"]),
                                   torch.from_numpy(g["neg_mask"]), m["m"])
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    for k, v in sd.items():
        ref = g["grad__" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(v.grad.numpy() - ref).max() <= 1e-4 * scale, k


def test_sparse_forward_matches_reference(golden):
    g, m = golden, golden["meta"]
    Z = _Z_nkd(g)
    N, K, d = Z.shape
    rowptr, col, rev = sparse_ref.csr_from_dense(g["adj"])
    H, p, a, s_raw = sparse_ref.forward(Z, rowptr, col, m["beta"], m["t"])
    src = sparse_ref.edge_rows(rowptr)
    assert (p == g["p"][src, col]).all()
    np.testing.assert_allclose(a, g["a"][src, col], rtol=2e-6)
    s = np.where(s_raw == 0, 1, s_raw)
    np.testing.assert_allclose(s, g["s"], rtol=2e-6)
    emb = H.reshape(N, K * d)                       # [N,K,d] row-major == cat(h_k, dim=1)
    np.testing.assert_allclose(emb, g["emb"], rtol=1e-5, atol=2e-6)
    # every ordered pair, incl. non-edges and the diagonal
    uu, vv = np.divmod(np.arange(N * N), N)
    prob = sparse_ref.score_pairs(Z, H, uu, vv, m["t"]).reshape(N, N)
    np.testing.assert_allclose(prob, g["link_pred"], rtol=1e-5, atol=2e-6)
    # rev really is the transpose permutation
    assert (src[rev] == col).all() and (col[rev] == src).all()


def test_sparse_backward_matches_reference_grads(golden):
    """Analytic edge-list backward (Appendix A.3) -> dZ, pushed through the MLP by autograd,
    must reproduce the reference's parameter gradients."""
    g, m = golden, golden["meta"]
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(g).items()}
    x = torch.from_numpy(g["x"])
    Zt = dense_ref.project(x, sd).permute(1, 0, 2).contiguous()      # [N,K,d]
    Z = Zt.detach().numpy()
    N = Z.shape[0]
    rowptr, col, rev = sparse_ref.csr_from_dense(g["adj"])
    H, p, a, s_raw = sparse_ref.forward(Z, rowptr, col, m["beta"], m["t"])
    pu, pv = np.nonzero(g["pos_mask"] == 1)              # the caller takes mask == 1: pairs that occur exactly once
    nu, nv = np.nonzero(g["neg_mask"] == 1)
    pp = sparse_ref.score_pairs(Z, H, pu, pv, m["t"])
    pn = sparse_ref.score_pairs(Z, H, nu, nv, m["t"])
    lab_p, lab_n = g["ori_adj"][pu, pv], g["ori_adj"][nu, nv]
    loss = metrics_ref.pair_bce(pp, lab_p, pn, lab_n, m["m"])
    assert abs(loss - float(g["loss"])) <= 2e-5 * max(1.0, abs(float(g["loss"])))
    gp = metrics_ref.bce_grad(pp, lab_p, 1.0)
    gn = metrics_ref.bce_grad(pn, lab_n, 1.0 / m["m"])
    au, av = np.r_[pu, nu], np.r_[pv, nv]
    dZ_s, dH = sparse_ref.score_pairs_bwd(Z, H, au, av, m["t"], np.r_[gp, gn])
    dZ = dZ_s + sparse_ref.route_aggregate_bwd(Z, rowptr, col, rev, p, a, s_raw, m["beta"], m["t"], dH)
    Zt.backward(torch.from_numpy(dZ))
    for k, v in sd.items():
        ref = g["grad__" + k]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(v.grad.numpy() - ref).max() <= 2e-4 * scale, k


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN_DIR, "auc_*.npz"))))
def test_auc_matches_sklearn_vectors(path):
    g = np.load(path)
    assert abs(metrics_ref.auc_tie_avg(g["y"], g["score"]) - float(g["auc"])) < 1e-12


def test_c_restatement_matches_numpy_oracle_and_reference(golden):
    """oracle/c/sparse_ref.c (OpenMP) against the numpy oracle and the reference's golden vectors."""
    from oracle import c_ref
    g, m = golden, golden["meta"]
    Z = _Z_nkd(g)
    N, K, d = Z.shape
    rowptr, col, _rev = sparse_ref.csr_from_dense(g["adj"])
    p, a, s_raw = c_ref.route(Z, rowptr, col, m["t"])
    p_n, a_n, _alpha, s_n = sparse_ref.route(Z, rowptr, col, m["t"])
    assert (p == p_n).all()
    np.testing.assert_allclose(a, a_n, rtol=2e-6)
    np.testing.assert_allclose(s_raw, s_n, rtol=1e-5, atol=1e-7)
    H = c_ref.aggregate(Z, rowptr, col, p, a, s_raw, m["beta"])
    np.testing.assert_allclose(H.reshape(N, K * d), g["emb"], rtol=1e-5, atol=2e-6)
    uu, vv = np.divmod(np.arange(N * N), N)
    prob = c_ref.score_pairs(Z, H, uu, vv, m["t"]).reshape(N, N)
    np.testing.assert_allclose(prob, g["link_pred"], rtol=1e-5, atol=2e-6)


def test_c_restatement_backward_matches_numpy_oracle(golden):
    """The gather-form backward of oracle/c/sparse_ref.c (used for parity at the full benchmark sizes) against the numpy
    oracle's scatter form, which test_sparse_backward_matches_reference_gradients pins to the reference's autograd."""
    from oracle import c_ref
    g, m = golden, golden["meta"]
    Z = _Z_nkd(g)
    N, K, d = Z.shape
    rng = np.random.default_rng(N * 131 + K)
    rowptr, col, rev = sparse_ref.csr_from_dense(g["adj"])
    H, p, a, s_raw = sparse_ref.forward(Z, rowptr, col, m["beta"], m["t"])
    P = 4 * N
    pu, pv = rng.integers(0, N, P), rng.integers(0, N, P)
    pu[:3], pv[:3] = pv[:3], pv[:3]                                   # a few self pairs
    gp = (rng.standard_normal(P) * 0.3).astype(np.float32)
    prob = sparse_ref.score_pairs(Z, H, pu, pv, m["t"])
    dZs_n, dH_n = sparse_ref.score_pairs_bwd(Z, H, pu, pv, m["t"], gp)
    dZs_c, dH_c = c_ref.score_pairs_bwd(Z, H, pu, pv, m["t"], prob, gp)
    tol = lambda ref: 2e-5 * max(np.abs(ref).max(), 1e-6)
    assert np.abs(dH_c - dH_n).max() <= tol(dH_n)
    assert np.abs(dZs_c - dZs_n).max() <= tol(dZs_n)
    dH_in = (dH_n + rng.standard_normal(dH_n.shape).astype(np.float32) * 0.1).astype(np.float32)
    dZ_n = sparse_ref.route_aggregate_bwd(Z, rowptr, col, rev, p, a, s_raw, m["beta"], m["t"], dH_in)
    dZ_c = c_ref.route_aggregate_bwd(Z, rowptr, col, p, a, s_raw, m["beta"], m["t"], dH_in)
    assert np.abs(dZ_c - dZ_n).max() <= tol(dZ_n)


@pytest.mark.parametrize("name", __import__("conftest").trajectory_names())
def test_oracle_follows_the_reference_training_trajectory(name):
    """tests/golden/traj_*.npz: the reference model under the reference's schedule (Adam, weight decay 5e-4,
    validation AUC from the pre-step forward, best weights, test AUC).  The dense oracle + the build's own BCE
    and AUC restatements must reproduce losses, AUCs and the kept weights."""
    import torch
    from conftest import load_trajectory
    from oracle import dense_ref, metrics_ref
    g = load_trajectory(name)
    m = g["meta"]
    sd = {k[4:]: torch.nn.Parameter(torch.from_numpy(v.copy())) for k, v in g.items() if k.startswith("sd__")}
    opt = torch.optim.Adam(list(sd.values()), lr=m["lr"], weight_decay=5e-4)
    x, adj, ori = (torch.from_numpy(g[k]) for k in ("x", "adj", "ori_adj"))
    mk = {k[6:]: g[k] == 1 for k in g if k.startswith("mask__")}
    best, kept = 0.0, None
    for ep in range(m["epochs"]):
        _emb, P = dense_ref.forward(x, adj, sd, m["beta"], m["t"])
        loss = dense_ref.bce_pair_loss(P, ori, torch.from_numpy(g["mask__pos_train"]), torch.from_numpy(g["mask__neg_train"]), m["m"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        auc = metrics_ref.auc_tie_avg(g["ori_adj"][mk["val"]], P.detach().numpy()[mk["val"]])
        assert abs(loss.item() - g["losses"][ep]) <= 2e-5 * abs(g["losses"][ep]), (ep, loss.item(), g["losses"][ep])
        assert abs(auc - g["val_aucs"][ep]) <= 1e-6, (ep, auc, g["val_aucs"][ep])
        if auc > best:
            best, kept = auc, {k: v.detach().clone() for k, v in sd.items()}
    for k, v in kept.items():
        np.testing.assert_allclose(v.numpy(), g["best__" + k], rtol=1e-4, atol=1e-6)
    _emb, P = dense_ref.forward(x, adj, kept, m["beta"], m["t"])
    assert abs(metrics_ref.auc_tie_avg(g["ori_adj"][mk["test"]], P.detach().numpy()[mk["test"]]) - float(g["test_auc"])) <= 1e-6


@pytest.mark.parametrize("case", ["k4_d8", "k5_d64"])
def test_oracle_follows_the_reference_over_adam_steps_with_fixed_masks(case):
    """tests/golden/adam_*.npz: 12 optimiser steps of the reference model under FIXED masks.  The oracle reproduces the
    losses, the weights after the last step — and the fact the fixture exists for: the number of non-zero entries of
    d loss / d link_pred moves from step to step (saturated positives carry exactly zero gradient until they
    de-saturate), so it says nothing about where the loss is taken."""
    import json
    import torch
    from conftest import load_golden
    from oracle import dense_ref
    c = load_golden(case)
    g = dict(np.load(os.path.join(GOLDEN_DIR, f"adam_{case}.npz"), allow_pickle=False))
    am, m = json.loads(str(g["meta"])), c["meta"]
    sd = {k[4:]: torch.nn.Parameter(torch.from_numpy(v.copy())) for k, v in c.items() if k.startswith("sd__")}
    opt = torch.optim.Adam(list(sd.values()), lr=am["lr"], weight_decay=am["weight_decay"])
    x, adj, ori = (torch.from_numpy(c[k]) for k in ("x", "adj", "ori_adj"))
    pm, nm = torch.from_numpy(c["pos_mask"]), torch.from_numpy(c["neg_mask"])
    assert int(((pm == 1) | (nm == 1)).sum()) == int(g["n_masked"])
    nnz = []
    for step in range(am["steps"]):
        _emb, P = dense_ref.forward(x, adj, sd, m["beta"], m["t"])
        P.retain_grad()
        loss = dense_ref.bce_pair_loss(P, ori, pm, nm, m["m"])
        opt.zero_grad()
        loss.backward()
        nnz.append(int(torch.count_nonzero(P.grad)))
        assert abs(loss.item() - g["losses"][step]) <= 2e-5 * abs(g["losses"][step]), (step, loss.item(), g["losses"][step])
        opt.step()
    assert nnz == g["nnz_grad"].tolist()
    assert nnz[0] < nnz[-1] == int(g["n_masked"])            # the non-zero set grows under fixed masks
    for k, v in sd.items():
        np.testing.assert_allclose(v.detach().numpy(), g["sd__" + k], rtol=2e-4, atol=2e-6)


def test_oracle_on_the_real_chameleon_fixture():
    """tests/golden/real_chameleon.npz (real dataset arrays + the reference model's trajectory): the dense oracle,
    the build's split, row standardisation, seeded init, BCE and AUC restatements reproduce the reference's first
    epochs on the real graph (3 epochs here: the dense [K,N,N] form takes seconds per epoch on CPU)."""
    import json
    import torch
    from disenlink_amd.datasets import standardise_rows
    from disenlink_amd.model import Disentangle
    from disenlink_amd.splits import make_link_split
    from oracle import dense_ref, metrics_ref
    g = np.load(os.path.join(GOLDEN_DIR, "real_chameleon.npz"))
    m = json.loads(str(g["meta"]))
    feats, edges = g["features"], g["edges"].astype(np.int64)
    n = feats.shape[0]
    split = make_link_split(edges[:, 0], edges[:, 1], n, m=m["m"], seed=m["split_seed"])

    def dense(u, v):
        a = np.zeros((n, n), dtype=np.float32)
        a[u, v] = 1.0
        return a
    ori = dense(edges[:, 0], edges[:, 1])
    adj = dense(split.train_src, split.train_dst)
    adj_sym = torch.from_numpy(((adj + adj.T) != 0).astype(np.float32))
    pos, neg = torch.from_numpy(dense(split.pos_train.u, split.pos_train.v)), torch.from_numpy(dense(split.neg_train.u, split.neg_train.v))
    val = dense(split.val.u, split.val.v) == 1
    torch.manual_seed(m["seed"])
    mod = Disentangle(feats.shape[1], m["nhid"], m["d"], nfactor=m["K"], beta=m["beta"], t=m["t"])   # parameters only
    sd = dict(mod.named_parameters())
    opt = torch.optim.Adam(list(sd.values()), lr=m["lr"], weight_decay=5e-4)
    x, ori_t = torch.from_numpy(standardise_rows(feats)), torch.from_numpy(ori)
    for ep in range(3):
        _emb, P = dense_ref.forward(x, adj_sym, sd, m["beta"], m["t"])
        loss = dense_ref.bce_pair_loss(P, ori_t, pos, neg, m["m"])
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert abs(loss.item() - g["losses"][ep]) <= 2e-5 * g["losses"][ep], (ep, loss.item(), g["losses"][ep])
        assert abs(metrics_ref.auc_tie_avg(ori[val], P.detach().numpy()[val]) - g["val_aucs"][ep]) <= 1e-6
