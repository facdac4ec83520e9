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


def _worker(rank, world, port, pb, sd, out):
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
        shard = dd.Shard.build(rank, world, pb["N"], pb["src"], pb["dst"], pb["pu"], pb["pv"], "cpu", seg_len=4)
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
                         rows=(r0, r1), pairs=(shard.pair_lo, shard.pair_hi),
                         grads={k: v.grad.numpy().copy() for k, v in model.named_parameters()})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_path_matches_unsharded_oracle(world):
    from disenlink_amd.model import Disentangle
    pb = _problem()
    torch.manual_seed(0)
    sd = Disentangle(pb["F"], pb["nhid"], pb["d"], nfactor=pb["K"], beta=pb["beta"], t=pb["t"]).state_dict()
    emb_ref, prob_ref, loss_ref, grads_ref = _reference(pb, sd)
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), pb, sd, out), nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))
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


def test_partition_helpers():
    from disenlink_amd import dist as dd
    assert dd.block_size(10, 4) == 3 and dd.padded_nodes(10, 4) == 12
    assert [dd.row_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 12)]
    pu = np.array([0, 0, 2, 3, 3, 3, 8, 9])
    assert dd.pair_slices(pu, 10, 4).tolist() == [0, 3, 6, 7, 8]
    with pytest.raises(ValueError, match="sorted by pu"):
        dd.Shard.build(0, 2, 10, [0], [1], [3, 1], [0, 0], "cpu")
