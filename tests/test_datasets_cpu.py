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


@pytest.mark.parametrize("school", ["Amherst41", "Reed98", "JohnsHopkins55"])
def test_fb100_equals_the_references_label_binarize_construction(school):
    """f4 guard: load_fb100 against the reference's construction, restated here line by line from
    /root/reference/other_hetero_datasets.py:131-154 (load_fb100_dataset) over load_data.py:11-19 with the installed
    sklearn: edge_index = A.nonzero(), features = hstack over the metadata columns (gender left out) of
    label_binarize(col, classes=np.unique(col)) — which gives ONE column for a two-valued column and a zero column for a
    single-valued one — as float32.  Equality, not shapes."""
    import scipy.io
    from sklearn.preprocessing import label_binarize
    from disenlink_amd.datasets import load_fb100
    path = _need(f"{REF}/data/facebook100/{school}.mat")
    mat = scipy.io.loadmat(path)
    A, metadata = mat["A"], mat["local_info"].astype(np.int64)           # (np.int of the reference: removed from numpy)
    row, col = A.nonzero()
    feature_vals = np.hstack((np.expand_dims(metadata[:, 0], 1), metadata[:, 2:]))
    features = np.empty((A.shape[0], 0))
    for c in range(feature_vals.shape[1]):
        feat_col = feature_vals[:, c]
        features = np.hstack((features, label_binarize(feat_col, classes=np.unique(feat_col))))
    want_x = features.astype(np.float32)                                # torch.tensor(features, dtype=torch.float)
    ds = load_fb100(path, school, standardise=False)
    assert ds.x.dtype == np.float32 and ds.x.shape == want_x.shape and np.array_equal(ds.x, want_x)
    assert np.array_equal(ds.src, row.astype(np.int64)) and np.array_equal(ds.dst, col.astype(np.int64))
    # ... and the standardised form the script trains on (main_disentangled.py:107-109)
    import torch
    t = torch.from_numpy(want_x)
    ref = (t - torch.mul(torch.ones(t.shape), torch.mean(t, dim=1).unsqueeze(dim=1))) / torch.std(t, dim=1).unsqueeze(dim=1)
    np.testing.assert_allclose(load_fb100(path, school).x, ref.numpy(), rtol=1e-6, atol=1e-6)


def test_twitch_de_equals_the_references_arrays():
    """f4 guard: load_twitch against /root/reference/load_data.py:21-70 restated here (np.int -> np.int64, the one edit the
    installed numpy forces) + main_disentangled.py:111-116: features — 3170 one-hot columns, zero columns removed — equal
    exactly; the edge rows are those of A.nonzero() with the reversed rows appended, as a multiset (the loader keeps the
    csv order, scipy's CSR walks them row by row: only the order differs, and the split shuffles it anyway)."""
    import csv
    import json
    import scipy.sparse
    from disenlink_amd.datasets import load_twitch
    lang = "DE"
    filepath = _need(f"{REF}/data/twitch/{lang}")
    label, node_ids, src, targ, uniq_ids = [], [], [], [], set()
    with open(f"{filepath}/musae_{lang}_target.csv", "r") as f:
        reader = csv.reader(f)
        next(reader)
        for row in reader:
            node_id = int(row[5])
            if node_id not in uniq_ids:
                uniq_ids.add(node_id)
                label.append(int(row[2] == "True"))
                node_ids.append(int(row[5]))
    with open(f"{filepath}/musae_{lang}_edges.csv", "r") as f:
        reader = csv.reader(f)
        next(reader)
        for row in reader:
            src.append(int(row[0]))
            targ.append(int(row[1]))
    with open(f"{filepath}/musae_{lang}_features.json", "r") as f:
        j = json.load(f)
    n = len(label)
    A = scipy.sparse.csr_matrix((np.ones(len(src)), (np.array(src), np.array(targ))), shape=(n, n))
    features = np.zeros((n, 3170))
    for node, feats in j.items():
        if int(node) >= n:
            continue
        features[int(node), np.array(feats, dtype=int)] = 1
    features = features[:, np.sum(features, axis=0) != 0]
    r, c = A.nonzero()
    want_src, want_dst = np.concatenate([r, c]).astype(np.int64), np.concatenate([c, r]).astype(np.int64)   # :114-116
    ds = load_twitch(filepath, lang, standardise=False)
    assert ds.n_nodes == n and np.array_equal(ds.x, features.astype(np.float32))
    key = lambda a, b: np.sort(a * n + b)
    assert np.array_equal(key(ds.src, ds.dst), key(want_src, want_dst))


def _processed_tensors(path):
    """The tensors of a torch_geometric ``processed/data.pt`` (datasets.read_pyg_data: no torch_geometric needed)."""
    from disenlink_amd.datasets import read_pyg_data
    return read_pyg_data(path)


@pytest.mark.parametrize("name,n,rows,loops", [("texas", 183, 325, 16), ("cornell", 183, 298, 3), ("wisconsin", 251, 515, 16)])
def test_webkb_equals_the_reference_processed_file(name, n, rows, loops):
    """main_disentangled.py:69-71: WebKB(root, name)[0] — x and edge_index of the reference's own processed file."""
    import torch
    from disenlink_amd.datasets import load_webkb
    raw = _need(f"{REF}/data/{name}/raw")
    ds = load_webkb(raw, name, standardise=False)
    assert ds.n_nodes == n and ds.x.shape == (n, 1703) and ds.src.size == rows and int((ds.src == ds.dst).sum()) == loops
    m = _processed_tensors(_need(f"{REF}/data/{name}/processed/data.pt"))
    assert np.array_equal(np.stack([ds.src, ds.dst]), m["edge_index"].numpy())
    assert np.array_equal(ds.x, m["x"].numpy())
    t = m["x"]
    want = (t - t.mean(dim=1, keepdim=True)) / t.std(dim=1).unsqueeze(1)               # :95
    np.testing.assert_allclose(load_webkb(raw, name).x, want.numpy(), rtol=1e-6, atol=1e-6)


def test_amazon_npz_format(tmp_path):
    """Amazon(root, 'photo') parses CSR triplets; binarised attributes, no self-loops, undirected, coalesced."""
    import scipy.sparse as sp
    from disenlink_amd.datasets import load_amazon_npz
    rng = np.random.default_rng(3)
    n = 12
    a = sp.csr_matrix((rng.random((n, n)) < 0.2).astype(np.float32))
    x = sp.csr_matrix(rng.integers(0, 3, (n, 7)).astype(np.float32))
    p = tmp_path / "amazon_electronics_photo.npz"
    np.savez(p, adj_data=a.data, adj_indices=a.indices, adj_indptr=a.indptr, adj_shape=np.array(a.shape),
             attr_data=x.data, attr_indices=x.indices, attr_indptr=x.indptr, attr_shape=np.array(x.shape),
             labels=np.zeros(n, np.int64))
    ds = load_amazon_npz(str(p), standardise=False)
    d = a.toarray()
    np.fill_diagonal(d, 0)
    want = np.stack(np.nonzero((d + d.T) != 0))
    assert np.array_equal(np.stack([ds.src, ds.dst]), want)
    assert np.array_equal(ds.x, (x.toarray() > 0).astype(np.float32))


def test_deezer_mat_format(tmp_path):
    """other_hetero_datasets.py:160-164: A.nonzero() rows in the matrix's own order, features densified."""
    import scipy.io
    import scipy.sparse as sp
    from disenlink_amd.datasets import load_deezer
    rng = np.random.default_rng(4)
    n = 9
    a = sp.csr_matrix((rng.random((n, n)) < 0.3).astype(np.float64))
    feats = sp.csr_matrix(rng.integers(0, 2, (n, 5)).astype(np.float64))
    p = tmp_path / "deezer-europe.mat"
    scipy.io.savemat(p, {"A": a, "label": rng.integers(0, 2, (1, n)), "features": feats})
    ds = load_deezer(str(p), standardise=False)
    row, col = scipy.io.loadmat(p)["A"].nonzero()
    assert np.array_equal(ds.src, row) and np.array_equal(ds.dst, col) and ds.src.size == a.nnz
    assert np.array_equal(ds.x, feats.toarray().astype(np.float32))


def test_cli_reads_webkb(tmp_path):
    """--dataset texas goes through load_dataset (main_disentangled.py:69-71)."""
    import argparse
    from disenlink_amd.main import load_dataset
    _need(f"{REF}/data/texas/raw")
    args = argparse.Namespace(data_file=None, data_root=f"{REF}/data", synthetic=False, dataset="texas", sub_dataset="")
    ds = load_dataset(args)
    assert ds.n_nodes == 183 and ds.src.size == 325
    np.testing.assert_allclose(ds.x.mean(axis=1), 0, atol=1e-5)


def test_arxiv_year_mini_and_cli():
    """main_disentangled.py:124-129: --dataset year --miniid i reads mini/year<i>.pt (a pickled PyG Data)."""
    import argparse
    from disenlink_amd.datasets import load_arxiv_year_mini, read_pyg_data
    from disenlink_amd.main import load_dataset
    path = _need(f"{REF}/mini/year0.pt")
    m = read_pyg_data(path)
    ds = load_arxiv_year_mini(path, standardise=False)
    assert ds.x.shape == (5443, 128) and ds.src.size == 15497 and int(max(ds.src.max(), ds.dst.max())) == 5442
    assert np.array_equal(ds.x, m["x"].numpy()) and np.array_equal(np.stack([ds.src, ds.dst]), m["edge_index"].numpy())
    args = argparse.Namespace(data_file=None, data_root=f"{REF}/data", synthetic=False, dataset="year", sub_dataset="", miniid=0)
    got = load_dataset(args)
    t = m["x"]
    want = (t - t.mean(dim=1, keepdim=True)) / t.std(dim=1).unsqueeze(1)                  # :127
    np.testing.assert_allclose(got.x, want.numpy(), rtol=1e-6, atol=1e-6)
    assert got.src.size == 15497
