"""Dataset front-end vs the facts SURVEY.md Appendix B measured on the reference's data files.
The files live only in the build container (/root/reference); the tests skip where they are absent
(the GPU box) — nothing on the GPU path reads them."""
import os

import numpy as np
import pytest

REF = "/root/reference"


def _need(path):
    if not os.path.exists(path):
        pytest.skip(f"{path} not present (reference data does not travel)")
    return path


def test_standardise_rows_matches_torch():
    import torch
    from disenlink_amd.datasets import standardise_rows
    x = np.random.default_rng(0).standard_normal((7, 13)).astype(np.float32) * 3 + 1
    t = torch.from_numpy(x)
    ref = (t - t.mean(dim=1, keepdim=True)) / t.std(dim=1).unsqueeze(1)          # main_disentangled.py:99
    np.testing.assert_allclose(standardise_rows(x), ref.numpy(), rtol=1e-6, atol=1e-6)


def test_chameleon_npz():
    from disenlink_amd.datasets import load_npz
    ds = load_npz(_need(f"{REF}/data_pre_false/chameleon/raw/chameleon.npz"), "chameleon")
    assert ds.n_nodes == 2277 and ds.x.shape == (2277, 128) and ds.src.size == 72202
    assert np.unique(ds.src * 2277 + ds.dst).size == 62792 and int((ds.src == ds.dst).sum()) == 100
    np.testing.assert_allclose(ds.x.mean(axis=1), 0, atol=1e-5)
    np.testing.assert_allclose(ds.x.std(axis=1, ddof=1), 1, rtol=1e-4)


def test_squirrel_edge_list_and_split():
    from disenlink_amd.datasets import load_geom_gcn
    from disenlink_amd.graph import Graph
    import torch
    ds = load_geom_gcn(_need(f"{REF}/data/squirrel/geom_gcn/raw/out1_graph_edges.txt"), name="squirrel")
    assert ds.n_nodes == 5201 and ds.src.size == 217073 and int((ds.src == ds.dst).sum()) == 140
    g = Graph.from_edge_rows(torch.from_numpy(ds.src), torch.from_numpy(ds.dst), 5201)
    deg = (g.rowptr[1:] - g.rowptr[:-1]).numpy()
    assert g.n_edges == 396846 and int(deg.max()) == 1904 and int(np.median(deg)) == 17


def test_cora_planetoid():
    from disenlink_amd.datasets import load_planetoid
    ds = load_planetoid(_need(f"{REF}/data/cora/raw"), "cora")
    assert ds.n_nodes == 2708 and ds.x.shape[1] == 1433 and ds.src.size == 10556
    assert not (ds.src == ds.dst).any()
    rs = ds.x.sum(axis=1)
    assert rs.min() >= 1 and rs.max() <= 30 and set(np.unique(ds.x)) == {0.0, 1.0}      # binary, not standardised
    key = ds.src * 2708 + ds.dst
    assert np.unique(key).size == key.size and np.isin(ds.dst * 2708 + ds.src, key).all()   # undirected, coalesced


def test_fb100_amherst(tmp_path):
    from disenlink_amd.datasets import load_binary, load_fb100, save_binary
    ds = load_fb100(_need(f"{REF}/data/facebook100/Amherst41.mat"), "Amherst41", standardise=False)
    assert ds.n_nodes == 2235 and ds.src.size == 181908
    assert set(np.unique(ds.x)) <= {0.0, 1.0} and ds.x.shape[1] > 100
    p = str(tmp_path / "amherst.npz")
    save_binary(ds, p)
    back = load_binary(p)
    assert back.name == "Amherst41" and np.array_equal(back.src, ds.src) and np.array_equal(back.x, ds.x)


def test_twitch_de():
    from disenlink_amd.datasets import load_twitch
    ds = load_twitch(_need(f"{REF}/data/twitch/DE"), "DE", standardise=False)
    assert ds.n_nodes == 9498 and ds.src.size == 2 * 153138 and ds.x.shape[1] <= 3170
