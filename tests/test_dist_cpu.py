"""world_size-2 (and 3) gloo runs of the row-sharded path on CPU: the sharding choreography of
disenlink_amd/dist.py (partition, padding, all-gathers, autograd glue, gradient all-reduce) with
the oracle standing in for the HIP kernels, checked against the unsharded dense oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from oracle import dense_ref


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem(seed=3, N=37, F=9, K=4, d=8, nhid=6):
    rng = np.random.default_rng(seed)
    E = 4 * N
    src, dst = rng.integers(0, N, E), rng.integers(0, N, E)
    src[:N - 1] = 0
    dst[:N - 1] = np.arange(1, N)                      # a hub touching every shard
    iso = N - 2
    keep = (src != iso) & (dst != iso)
    src, dst = src[keep], dst[keep]
    x = (rng.standard_normal((N, F)) * 0.6).astype(np.float32)
    P = 300
    pu, pv = np.sort(rng.integers(0, N, P)), rng.integers(0, N, P)
    label = (rng.random(P) < 0.4).astype(np.float32)
    return dict(N=N, F=F, K=K, d=d, nhid=nhid, src=src, dst=dst, x=x, pu=pu, pv=pv, label=label, beta=0.6, t=1.0)


def _reference(pb, sd):
    """Unsharded dense oracle: loss = mean BCE over the pair list, parameter gradients by autograd."""
    N = pb["N"]
    adj = np.zeros((N, N), np.float32)
    adj[pb["src"], pb["dst"]] = 1
    adj = ((adj + adj.T) != 0).astype(np.float32)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    emb, P = dense_ref.forward(torch.from_numpy(pb["x"]), torch.from_numpy(adj), sd, pb["beta"], pb["t"])
    prob = P[torch.from_numpy(pb["pu"]), torch.from_numpy(pb["pv"])]
    loss = torch.nn.functional.binary_cross_entropy(prob, torch.from_numpy(pb["label"]))
    loss.backward()
    return emb.detach().numpy(), prob.detach().numpy(), float(loss.detach()), {k: v.grad.numpy() for k, v in sd.items()}


def _worker(rank, world, port, pb, sd, out, n_chunks=1, z_by_peer=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disenlink_amd import dist as dd
        from disenlink_amd.model import Disentangle
        from oracle_backend import OracleBackend
        torch.set_num_threads(1)
        model = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"])
        model.load_state_dict(sd)
        shard = dd.Shard.build(rank, world, pb["N"], pb["src"], pb["dst"], pb["pu"], pb["pv"], "cpu", seg_len=4,
                               n_chunks=n_chunks, z_by_peer=z_by_peer)
        assert bool(shard.route_by_peer) == bool(z_by_peer)
        r0, r1 = shard.local_real_rows()
        emb, prob = dd.sharded_forward(model, torch.from_numpy(pb["x"][r0:r1]), shard, backend=OracleBackend())
        lab = torch.from_numpy(pb["label"][shard.pair_lo:shard.pair_hi])
        # local SUM / GLOBAL count, so that summing the replicas' gradients gives the global mean's gradient
        loss = torch.nn.functional.binary_cross_entropy(prob, lab, reduction="sum") / shard.n_pairs_total
        model.zero_grad()
        loss.backward()
        dd.allreduce_gradients(model)
        tot = loss.detach().clone()
        dist.all_reduce(tot)
        out[rank] = dict(emb=emb.detach().numpy()[: r1 - r0], prob=prob.detach().numpy(), loss=float(tot),
                         rows=(r0, r1), pairs=(shard.pair_lo, shard.pair_hi), work=shard.work(),
                         groups=[int(i.numel()) for i, _ in shard.pair_groups],
                         grads={k: v.grad.numpy().copy() for k, v in model.named_parameters()})
    finally:
        dist.destroy_process_group()


def _loss_worker(rank, world, port, pb, sd, out, table, z_by_peer=False, dh_wire=None, one_pass=None):
    """The training step through sharded_forward_loss: one-pass scorer over the local incidence rows, no (prob, g_prob)
    all-gather; `table` = storage type of the gathered tables."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if dh_wire is not None:
        os.environ["DL_DH_GATHER"] = dh_wire
    if one_pass is not None:
        os.environ["DL_ONE_PASS_SCORER"] = one_pass
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disenlink_amd import dist as dd
        from disenlink_amd.model import Disentangle
        from oracle_backend import OracleBackend
        torch.set_num_threads(1)
        model = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"],
                            table_dtype=torch.bfloat16 if table == "bf16" else torch.float32)
        model.load_state_dict(sd)
        shard = dd.Shard.build(rank, world, pb["N"], pb["src"], pb["dst"], pb["pu"], pb["pv"], "cpu", seg_len=4, n_chunks=2,
                               z_by_peer=z_by_peer)
        r0, r1 = shard.local_real_rows()
        P = pb["pu"].size
        label = torch.from_numpy(pb["label"])
        weight = torch.full((P,), 1.0 / P)                          # the GLOBAL mean's weights, replicated
        emb, prob, loss = dd.sharded_forward_loss(model, torch.from_numpy(pb["x"][r0:r1]), shard, label, weight,
                                                  backend=OracleBackend())
        model.zero_grad()
        loss.backward()
        dd.allreduce_gradients(model)
        tot = loss.detach().clone()
        dist.all_reduce(tot)
        out[rank] = dict(emb=emb.detach().numpy()[: r1 - r0], prob=prob.detach().numpy(), loss=float(tot),
                         rows=(r0, r1), pairs=(shard.pair_lo, shard.pair_hi),
                         grads={k: v.grad.numpy().copy() for k, v in model.named_parameters()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,table,z_by_peer", [(2, "f32", False), (4, "f32", False), (4, "bf16", False), (4, "f32", True)])
def test_sharded_training_step_with_the_one_pass_scorer(world, table, z_by_peer):
    """sharded_forward_loss over gloo: every rank runs the scorer's forward + loss gradient + backward in one pass over
    ITS incidence rows (all pairs touching its nodes), so no (prob, g_prob) all-gather exists; the summed replicas'
    gradients equal the unsharded dense oracle's.  bf16: the gathered Z / H tables are bf16 (half the all-gather bytes);
    the result follows the fp32 reference within bf16 rounding."""
    from disenlink_amd.model import Disentangle
    pb = _skewed_problem()
    torch.manual_seed(0)
    sd = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"]).state_dict()
    emb_ref, prob_ref, loss_ref, grads_ref = _reference(pb, sd)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_loss_worker, args=(world, _free_port(), pb, sd, out, table, z_by_peer), nprocs=world, join=True)
    tol = dict(rtol=1e-5, atol=1e-6) if table == "f32" else dict(rtol=5e-2, atol=2e-2)
    gtol = 2e-4 if table == "f32" else 8e-2
    for r in range(world):
        o = out[r]
        (r0, r1), (q0, q1) = o["rows"], o["pairs"]
        np.testing.assert_allclose(o["emb"], emb_ref[r0:r1], **tol)
        np.testing.assert_allclose(o["prob"], prob_ref[q0:q1], **tol)
        assert abs(o["loss"] - loss_ref) < (1e-5 if table == "f32" else 3e-2) * max(1.0, abs(loss_ref))
        for k, gref in grads_ref.items():
            scale = max(np.abs(gref).max(), 1e-6)
            assert np.abs(o["grads"][k] - gref).max() <= gtol * scale, (r, k, np.abs(o["grads"][k] - gref).max() / scale)


def _skewed_problem(seed=5, N=90, F=7, K=4, d=8, nhid=5):
    """Heavy-tailed degrees (a few hubs, most nodes with a handful of edges), hubs at low ids."""
    rng = np.random.default_rng(seed)
    w = 1.0 / (np.arange(N) + 2.0) ** 0.9
    w /= w.sum()
    E = 12 * N
    src, dst = rng.choice(N, E, p=w), rng.integers(0, N, E)
    x = (rng.standard_normal((N, F)) * 0.6).astype(np.float32)
    P = 500
    pu, pv = np.sort(rng.choice(N, P, p=w)), rng.integers(0, N, P)
    label = (rng.random(P) < 0.4).astype(np.float32)
    return dict(N=N, F=F, K=K, d=d, nhid=nhid, src=src, dst=dst, x=x, pu=pu, pv=pv, label=label, beta=0.7, t=1.0)


@pytest.mark.parametrize("world,n_chunks,skewed,z_by_peer", [(2, 1, False, False), (3, 1, False, False), (3, 2, False, True),
                                                              (4, 3, True, False), (4, 3, True, True), (8, 2, True, False)])
def test_sharded_path_matches_unsharded_oracle(world, n_chunks, skewed, z_by_peer):
    """2 / 3 / 4 ranks over gloo, work-balanced blocks (padded to the largest, ids relabelled), with the H all-gather
    blocking (n_chunks = 1) or in asynchronous row chunks with the pairs scored in arrival order; z_by_peer: the Z table
    gathered peer block by peer block with the routing in arrival order (local columns first)."""
    from disenlink_amd.model import Disentangle
    pb = _skewed_problem() if skewed else _problem()
    torch.manual_seed(0)
    sd = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"]).state_dict()
    emb_ref, prob_ref, loss_ref, grads_ref = _reference(pb, sd)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), pb, sd, out, n_chunks, z_by_peer), nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))
    if n_chunks > 1:                                        # every local pair sits in exactly one arrival group
        for r in range(world):
            q0, q1 = out[r]["pairs"]
            assert len(out[r]["groups"]) == n_chunks + 1 and sum(out[r]["groups"]) == q1 - q0
    if skewed:                                              # blocks follow the work, not the node count
        rows = [out[r]["work"]["rows"] for r in range(world)]
        assert max(rows) > 2 * min(rows), rows
    covered_rows, covered_pairs = 0, 0
    for r in range(world):
        o = out[r]
        r0, r1 = o["rows"]
        q0, q1 = o["pairs"]
        np.testing.assert_allclose(o["emb"], emb_ref[r0:r1], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(o["prob"], prob_ref[q0:q1], rtol=1e-5, atol=1e-6)
        assert abs(o["loss"] - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
        for k, gref in grads_ref.items():
            scale = max(np.abs(gref).max(), 1e-6)
            assert np.abs(o["grads"][k] - gref).max() <= 2e-4 * scale, (r, k)
        covered_rows += r1 - r0
        covered_pairs += q1 - q0
    assert covered_rows == pb["N"] and covered_pairs == pb["pu"].size


def test_routing_in_arrival_order_reads_only_blocks_that_have_arrived():
    """route_in_arrival_order without any process group: a stand-in gather hands over one peer block per wait(q), every
    block that has not arrived yet is NaN.  The entries routed before a block's arrival must not have read it (no NaN
    in p / a / s) and the result equals ONE routing pass over the complete table bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from disenlink_amd import dist as dd
    from oracle_backend import OracleBackend
    pb = _skewed_problem(seed=11, N=120)
    world, K, d = 4, pb["K"], pb["d"]
    rng = np.random.default_rng(0)
    for rank in range(world):
        shard = dd.Shard.build(rank, world, pb["N"], pb["src"], pb["dst"], pb["pu"], pb["pv"], "cpu", seg_len=4,
                               with_backward=False, z_by_peer=True)
        assert len(shard.route_by_peer) == world and any(g is not None for g in shard.route_by_peer)
        # every entry belongs to exactly one peer's routing plan
        seen = np.zeros(shard.graph.n_edges, np.int64)
        for g in shard.route_by_peer:
            if g is not None:
                r = g.route
                for rw, b, e in zip(r.seg_row.numpy(), r.seg_beg.numpy(), r.seg_end.numpy()):
                    if rw >= 0:
                        seen[b:e] += 1
        assert np.all(seen == 1)
        Zfull = torch.from_numpy((rng.standard_normal((shard.n_pad, K, d)) * 0.4).astype(np.float32))
        B = shard.part.block
        be = OracleBackend()
        s_ref = torch.zeros((shard.n_pad, K))
        p_ref, a_ref = be.route_fwd(shard.graph, Zfull, 1.0, s_ref)

        class Reveal:                                            # the gather: block q appears at wait(q)
            def __init__(self, Z):
                self.Z, self.seen = Z, []
            def wait(self, q):
                self.Z[q * B:(q + 1) * B] = Zfull[q * B:(q + 1) * B]
                self.seen.append(q)
            def wait_all(self):
                for q in range(world):
                    if q not in self.seen and q != rank:
                        self.wait(q)
        Z = torch.full_like(Zfull, float("nan"))
        Z[shard.lo:shard.hi] = Zfull[shard.lo:shard.hi]
        s = torch.full((shard.n_pad, K), float("nan"))
        rev = Reveal(Z)
        p, a = dd.route_in_arrival_order(be, shard, Z, 1.0, s, rev)
        assert rev.seen == [q for q in range(world) if q != rank]
        assert torch.equal(p, p_ref) and torch.equal(a, a_ref) and not torch.isnan(a).any()
        assert torch.equal(s[shard.lo:shard.hi], s_ref[shard.lo:shard.hi])


def test_work_balanced_partition_on_hub_skewed_graphs():
    """contiguous row blocks balanced by nnz, not by node count.  On the squirrel-shaped graph
    (median degree ~20, hubs beyond 1,000) and on the same graph with its hubs SORTED to the front, the heaviest shard's
    symmetrised nnz stays within 1.15 x the mean for 2, 4 and 8 ranks; equal node blocks do not."""
    from disenlink_amd import dist as dd
    from disenlink_amd.data import synthetic_graph
    sg = synthetic_graph("squirrel", seed=0)
    N = sg.n_nodes
    for relabel in (False, True):
        src, dst = sg.src, sg.dst
        if relabel:                                         # hubs first: the worst case for equal node blocks
            deg = np.bincount(src, minlength=N) + np.bincount(dst, minlength=N)
            rank_of = np.empty(N, dtype=np.int64)
            rank_of[np.argsort(-deg, kind="stable")] = np.arange(N)
            src, dst = rank_of[src], rank_of[dst]
        key = np.unique(np.concatenate([src * N + dst, dst * N + src]))
        nnz_row = np.bincount(key // N, minlength=N)        # the symmetrised, binarised adjacency the kernels walk
        for world in (2, 4, 8):
            part = dd.Partition.build(N, world, src, dst, balance="nnz", n_chunks=4)
            assert part.cuts[0] == 0 and part.cuts[-1] == N and (np.diff(part.cuts) >= 0).all()
            assert part.block % 4 == 0 and part.block >= np.diff(part.cuts).max() and part.n_pad == world * part.block
            nnz = np.array([nnz_row[part.cuts[r]:part.cuts[r + 1]].sum() for r in range(world)], dtype=np.float64)
            assert nnz.max() / nnz.mean() <= 1.15, (relabel, world, nnz)
            ids = np.arange(N)
            pad = part.to_padded(ids)
            assert (np.diff(pad) > 0).all() and (pad // part.block == np.searchsorted(part.cuts, ids, side="right") - 1).all()
            if relabel and world >= 4:
                eq = dd.Partition.build(N, world, balance="nodes")
                nnz_eq = np.array([nnz_row[eq.cuts[r]:eq.cuts[r + 1]].sum() for r in range(world)], dtype=np.float64)
                assert nnz_eq.max() / nnz_eq.mean() > 1.5
    assert dd.balanced_cuts(np.array([1, 1, 100, 1, 1, 1]), 3).tolist() == [0, 2, 3, 6]      # a hub gets its own block


def test_partition_edge_cases():
    """More ranks than nodes, ranks without rows, chunks that do not divide the block, no edges at all."""
    from disenlink_amd import dist as dd
    part = dd.Partition.build(3, 8, np.array([0, 1]), np.array([1, 2]), n_chunks=4)
    assert part.cuts[0] == 0 and part.cuts[-1] == 3 and (np.diff(part.cuts) >= 0).all() and (np.diff(part.cuts) <= 1).all()
    assert part.block == 4 and part.n_pad == 32 and part.chunk_rows == 1                # block padded up to the chunk count
    pad = part.to_padded(np.arange(3))
    assert len(set(pad.tolist())) == 3 and (pad % part.block == 0).all()                # one real row per owning block
    none = dd.Partition.build(10, 4, np.zeros(0, np.int64), np.zeros(0, np.int64), n_chunks=3)
    assert none.cuts.tolist() == [0, 2, 5, 7, 10] and none.block == 3                   # equal weights: near-equal blocks
    sh = dd.Shard.build(5, 8, 3, [0, 1], [1, 2], [0, 1, 2], [2, 0, 1], "cpu", seg_len=4, n_chunks=2)   # a rank that owns nothing
    r0, r1 = sh.local_real_rows()
    assert r1 - r0 == 0 and sh.graph.n_edges == 0 and sh.pairs.n_pairs == 0 and sh.hi - sh.lo == sh.part.block
    assert [int(i.numel()) for i, _ in sh.pair_groups] == [0, 0, 0]


def test_partition_helpers():
    from disenlink_amd import dist as dd
    assert dd.block_size(10, 4) == 3 and dd.padded_nodes(10, 4) == 12
    assert [dd.row_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 12)]
    pu = np.array([0, 0, 2, 3, 3, 3, 8, 9])
    assert dd.pair_slices(pu, 10, 4).tolist() == [0, 3, 6, 7, 8]
    with pytest.raises(ValueError, match="sorted by pu"):
        dd.Shard.build(0, 2, 10, [0], [1], [3, 1], [0, 0], "cpu")


# --------------------------------------------------------------------------- the sharded train / eval loop
def _split_from_trajectory(g):
    """The reference caller's masks of a trajectory fixture as the pair lists of a LinkSplit (mask == 1 for the train
    sets — summed masks, main_disentangled.py:176-179,195 — binarised validation / test masks, labels from ori_adj)."""
    from disenlink_amd.splits import LinkSplit, PairSet
    ori = g["ori_adj"]

    def pairs(mask, only_single):
        u, v = np.nonzero(mask == 1 if only_single else mask != 0)
        return PairSet(u.astype(np.int64), v.astype(np.int64), ori[u, v].astype(np.float32))

    src, dst = np.nonzero(g["adj"])
    return LinkSplit(int(g["meta"]["N"]), src.astype(np.int64), dst.astype(np.int64), pairs(g["mask__pos_train"], True),
                     pairs(g["mask__neg_train"], True), pairs(g["mask__val"], False), pairs(g["mask__test"], False),
                     int(g["meta"]["m"]))


def _train_worker(rank, world, port, name, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_trajectory
        from disenlink_amd.model import Disentangle
        from disenlink_amd.train import prepare_run_sharded, run_link_prediction_sharded
        from oracle_backend import OracleBackend
        torch.set_num_threads(1)
        g = load_trajectory(name)
        m = g["meta"]
        model = Disentangle(m["F"], m["nhid"], m["d"], nfactor=m["K"], beta=m["beta"], t=m["t"])
        model.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")})
        run = prepare_run_sharded(_split_from_trajectory(g), rank, world, "cpu", n_chunks=2)
        r0, r1 = run.shard.local_real_rows()
        res = run_link_prediction_sharded(model, torch.from_numpy(g["x"][r0:r1]), run, epochs=m["epochs"], lr=m["lr"],
                                          patience=200, backend=OracleBackend())
        out[rank] = dict(losses=res.losses, val_aucs=res.val_aucs, test_auc=res.test_auc, best=res.best_val_auc,
                         sd={k: v.numpy().copy() for k, v in model.state_dict().items()},
                         val=(run.val_plan.n_pos, run.val_plan.n_neg), rows=(r0, r1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("traj_k4", 2), ("traj_k8", 2), ("traj_k4", 4)])
def test_sharded_training_loop_reproduces_the_reference_trajectories(name, world):
    """train.run_link_prediction_sharded over gloo (rows, features and pair list sharded; gradient all-reduce; the
    tie-aware validation / test AUC summed across the ranks) against the trajectories recorded from the REFERENCE model
    under the reference's own schedule (tests/golden/traj_*.npz): per-epoch loss and validation AUC, the test AUC with the
    best weights — to the tolerances of the single-process trajectory test.  Every rank must see identical numbers and end
    with identical weights (lock-step early stopping without a broadcast)."""
    from conftest import load_trajectory
    g = load_trajectory(name)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_train_worker, args=(world, _free_port(), name, out), nprocs=world, join=True)
    r = out[0]
    n_val = int((g["mask__val"] != 0).sum())
    assert sum(r["val"]) == n_val
    for ep in range(g["meta"]["epochs"]):
        assert abs(r["losses"][ep] - g["losses"][ep]) <= 2e-4 * abs(g["losses"][ep]), (ep, r["losses"][ep], g["losses"][ep])
        assert abs(r["val_aucs"][ep] - g["val_aucs"][ep]) <= 2e-3, (ep, r["val_aucs"][ep], g["val_aucs"][ep])
    assert abs(r["test_auc"] - float(g["test_auc"])) <= 5e-3
    for k, v in r["sd"].items():                                  # best weights == the reference's best weights
        np.testing.assert_allclose(v, g["best__" + k], rtol=2e-3, atol=2e-5, err_msg=k)
    for q in range(1, world):
        assert out[q]["losses"] == r["losses"] and out[q]["val_aucs"] == r["val_aucs"] and out[q]["test_auc"] == r["test_auc"]
        for k, v in r["sd"].items():
            assert np.array_equal(out[q]["sd"][k], v), k


def _auc_worker(rank, world, port, label, score, cuts, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disenlink_amd.metrics import ShardedAucPlan
        a, b = cuts[rank], cuts[rank + 1]
        plan = ShardedAucPlan(torch.from_numpy(label[a:b]))
        out[rank] = plan.auc(torch.from_numpy(score[a:b]))
    finally:
        dist.destroy_process_group()


def test_cross_rank_auc_equals_sklearn_on_the_concatenated_vectors():
    """metrics.ShardedAucPlan: the Mann-Whitney counts are additive over slices — every rank counts its positives
    against the all-gathered negatives, the counts are all-reduced.  Checked against the sklearn golden vectors (heavy
    ties at 1.0, all ties, no ties) cut into 3 UNEQUAL slices, one of them holding a single class only."""
    import glob
    from conftest import GOLDEN_DIR
    for path in sorted(glob.glob(os.path.join(GOLDEN_DIR, "auc_*.npz"))):
        g = np.load(path)
        y, sc = g["y"].astype(np.float32), g["score"].astype(np.float32)
        order = np.argsort(-y, kind="stable")                     # positives first: the last slice is negatives only
        y, sc = y[order], sc[order]
        n = y.size
        cuts = [0, n // 7, n // 2, n]
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_auc_worker, args=(3, _free_port(), y, sc, cuts, out), nprocs=3, join=True)
        for r in range(3):
            assert abs(out[r] - float(g["auc"])) <= 1e-12, (path, r, out[r], float(g["auc"]))


def test_bf16_wire_of_the_dH_gather_costs_less_than_the_bf16_tables_themselves():
    """With bf16 forward tables the backward's dH all-gather — its largest message — travels as bf16 too
    (dist.gather_grad_rows; a rank's own rows stay exact).  Stated error: against the SAME run with an fp32 wire
    (DL_DH_GATHER=f32) the summed weight gradients move by less than 6e-3 of the largest gradient entry per tensor —
    well inside what the bf16 Z / H tables already cost against the fp32 reference (8e-2 in the test above) — and the
    forward (loss, probabilities) is untouched bit for bit."""
    from disenlink_amd.model import Disentangle
    pb = _skewed_problem()
    torch.manual_seed(0)
    sd = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"]).state_dict()
    runs = {}
    for wire in ("f32", "bf16"):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_loss_worker, args=(4, _free_port(), pb, sd, out, "bf16", False, wire), nprocs=4, join=True)
        runs[wire] = {r: out[r] for r in range(4)}
    worst = 0.0
    for k, g32 in runs["f32"][0]["grads"].items():
        g16 = runs["bf16"][0]["grads"][k]
        worst = max(worst, float(np.abs(g16 - g32).max() / max(np.abs(g32).max(), 1e-30)))
    assert 0.0 < worst < 6e-3, worst                             # it IS a different wire, and it costs this little
    for r in range(4):
        assert runs["f32"][r]["loss"] == runs["bf16"][r]["loss"]
        assert np.array_equal(runs["f32"][r]["prob"], runs["bf16"][r]["prob"])


@pytest.mark.parametrize("world,table", [(2, "f32"), (4, "f32"), (4, "bf16")])
def test_sharded_training_step_from_the_touching_pairs(world, table):
    """The other scorer of sharded_forward_loss (DL_ONE_PASS_SCORER=0; the default for cache-resident bf16 tables, where
    the one-pass kernel of wide shapes runs one wave per SIMD): every rank scores the pairs that TOUCH its nodes itself
    (Shard.touching), forms their BCE gradient and runs the scorer backward over that list — no per-pair data crosses the
    ranks.  Same results as the one-pass form: losses, probabilities of the owned pairs, summed weight gradients against
    the unsharded dense oracle."""
    from disenlink_amd.model import Disentangle
    pb = _skewed_problem()
    torch.manual_seed(0)
    sd = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"]).state_dict()
    emb_ref, prob_ref, loss_ref, grads_ref = _reference(pb, sd)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_loss_worker, args=(world, _free_port(), pb, sd, out, table, False, None, "0"), nprocs=world, join=True)
    tol = dict(rtol=1e-5, atol=1e-6) if table == "f32" else dict(rtol=5e-2, atol=2e-2)
    gtol = 2e-4 if table == "f32" else 8e-2
    for r in range(world):
        o = out[r]
        q0, q1 = o["pairs"]
        np.testing.assert_allclose(o["prob"], prob_ref[q0:q1], **tol)
        r0, r1 = o["rows"]
        np.testing.assert_allclose(o["emb"], emb_ref[r0:r1], **tol)
        assert abs(o["loss"] - loss_ref) <= (1e-5 if table == "f32" else 5e-2) * max(1.0, abs(loss_ref))
        for k, g in o["grads"].items():
            ref = grads_ref[k]
            assert np.abs(g - ref).max() <= gtol * max(np.abs(ref).max(), 1e-6), (k, np.abs(g - ref).max(), np.abs(ref).max())
    # the touching list: every pair with a local endpoint, the owned ones a contiguous range of it
    from disenlink_amd import dist as dd
    sh = dd.Shard.build(1, world, pb["N"], pb["src"], pb["dst"], pb["pu"], pb["pv"], "cpu", seg_len=4)
    idx, touch, a0, a1 = sh.touching()
    ppu, ppv = sh.part.to_padded(pb["pu"]), sh.part.to_padded(pb["pv"])
    want = np.flatnonzero(((ppu >= sh.lo) & (ppu < sh.hi)) | ((ppv >= sh.lo) & (ppv < sh.hi)))
    assert np.array_equal(idx.numpy(), want) and a1 - a0 == sh.pair_hi - sh.pair_lo
    assert np.array_equal(idx.numpy()[a0:a1], np.arange(sh.pair_lo, sh.pair_hi))


def _overlap_worker(rank, world, port, pb, sd, out, overlap, n_chunks):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DL_Z_OVERLAP=overlap)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from disenlink_amd import dist as dd
        from disenlink_amd.model import Disentangle
        from oracle_backend import OracleBackend
        torch.set_num_threads(1)
        model = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"])
        model.load_state_dict(sd)
        shard = dd.Shard.build(rank, world, pb["N"], pb["src"], pb["dst"], pb["pu"], pb["pv"], "cpu", seg_len=4,
                               n_chunks=n_chunks)
        r0, r1 = shard.local_real_rows()
        P = pb["pu"].size
        label, weight = torch.from_numpy(pb["label"]), torch.full((P,), 1.0 / P)
        calls = []
        orig = model.project
        model.project = lambda x: (calls.append(int(x.shape[0])), orig(x))[1]        # rows per projection launch
        dd.reset_message_counts()
        emb, prob, loss = dd.sharded_forward_loss(model, torch.from_numpy(pb["x"][r0:r1]), shard, label, weight,
                                                  backend=OracleBackend())
        fwd_counts = dd.reset_message_counts()
        model.zero_grad()
        loss.backward()
        dd.allreduce_gradients(model)
        out[rank] = dict(emb=emb.detach().numpy(), prob=prob.detach().numpy(), loss=float(loss.detach()), calls=calls,
                         counts=fwd_counts, block=shard.part.block,
                         grads={k: v.grad.numpy().copy() for k, v in model.named_parameters()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chunks", [(2, 2), (3, 4)])
def test_projection_in_row_chunks_with_the_z_exchange_in_flight(world, n_chunks):
    """dist.project_and_gather (the Z gather under the projection): with a chunked partition the
    local rows are projected chunk by chunk and every chunk's direct exchange is started before the next chunk is
    projected; against DL_Z_OVERLAP=0 (one projection, one all-gather) on the same problem: the same embedding rows,
    probabilities and loss, the same weight gradients to rounding (a sum over the chunks' backward passes), n_chunks
    projection launches of block / n_chunks rows, and the forward's Z all-gather replaced by n_chunks batches of
    2 (W - 1) point-to-point messages."""
    from disenlink_amd.model import Disentangle
    pb = _skewed_problem()
    torch.manual_seed(0)
    sd = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"]).state_dict()
    res = {}
    for overlap in ("1", "0"):
        mgr = mp.Manager()
        out = mgr.dict()
        mp.spawn(_overlap_worker, args=(world, _free_port(), pb, sd, out, overlap, n_chunks), nprocs=world, join=True)
        res[overlap] = {r: out[r] for r in range(world)}
    for r in range(world):
        on, off = res["1"][r], res["0"][r]
        np.testing.assert_allclose(on["emb"], off["emb"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(on["prob"], off["prob"], rtol=1e-6, atol=1e-7)
        assert abs(on["loss"] - off["loss"]) <= 1e-6 * max(1.0, abs(off["loss"]))
        for k, g in off["grads"].items():
            scale = max(np.abs(g).max(), 1e-6)
            assert np.abs(on["grads"][k] - g).max() <= 2e-5 * scale, (r, k)
        assert off["calls"] == [off["block"]] and on["calls"] == [on["block"] // n_chunks] * n_chunks
        assert on["counts"]["collectives"] == off["counts"]["collectives"] - 1
        assert on["counts"]["p2p_ops"] == off["counts"]["p2p_ops"] + n_chunks * 2 * (world - 1)
