"""Dataset front-end: the raw file formats the reference's link-prediction script
reads, turned into the two things the hot path needs — features ``x [N,F] fp32`` and directed edge
rows ``(src, dst)`` — without torch_geometric / ogb / torch_sparse (absent here and on the GPU box).

Each loader restates the arithmetic of the reference's loader it replaces (cited per function); the
per-row standardisation is the one of main_disentangled.py:99.  Nothing here runs per epoch.
``save_binary`` / ``load_binary`` cache a dataset as one ``.npz`` (x, src, dst) for the GPU box.
"""
from __future__ import annotations

import csv
import json
import os
import pickle
from dataclasses import dataclass

import numpy as np


@dataclass
class LinkDataset:
    name: str
    x: np.ndarray            # [N, F] float32
    src: np.ndarray          # directed edge rows as loaded (duplicates / self-loops kept)
    dst: np.ndarray

    @property
    def n_nodes(self) -> int:
        return int(self.x.shape[0])


def standardise_rows(x: np.ndarray) -> np.ndarray:
    """(x - mean_row) / std_row with the UNBIASED std (torch.std default), main_disentangled.py:99."""
    x = np.asarray(x, dtype=np.float32)
    return ((x - x.mean(axis=1, keepdims=True)) / x.std(axis=1, ddof=1, keepdims=True)).astype(np.float32)


def load_npz(path: str, name: str | None = None, standardise: bool = True) -> LinkDataset:
    """chameleon / squirrel / crocodile ``.npz`` (dataset.py:119-124): ``features`` and ``edges [E,2]``,
    NOT made undirected; the caller standardises rows (main_disentangled.py:97-101)."""
    with np.load(path, allow_pickle=True) as d:
        x = np.asarray(d["features"], dtype=np.float32)
        e = np.asarray(d["edges"], dtype=np.int64)
    return LinkDataset(name or os.path.basename(path), standardise_rows(x) if standardise else x, e[:, 0].copy(),
                       e[:, 1].copy())


def load_geom_gcn(edges_txt: str, features_txt: str | None = None, name: str | None = None,
                  standardise: bool = True) -> LinkDataset:
    """geom-gcn text files (dataset.py:91-102): a header line, then ``a\\tb`` per edge row; features as
    ``id\\tf1,f2,...\\tlabel``.  Without a feature file (the blob is missing from the reference tree for
    squirrel) the features are left empty and the caller supplies its own."""
    with open(edges_txt) as f:
        rows = [r.split("\t") for r in f.read().split("\n")[1:] if r.strip()]
    e = np.array([[int(a), int(b)] for a, b in rows], dtype=np.int64)
    n = int(e.max()) + 1
    if features_txt is not None:
        with open(features_txt) as f:
            lines = [r for r in f.read().split("\n")[1:] if r.strip()]
        x = np.array([[float(v) for v in r.split("\t")[1].split(",")] for r in lines], dtype=np.float32)
        if standardise:
            x = standardise_rows(x)
    else:
        x = np.zeros((n, 0), dtype=np.float32)
    return LinkDataset(name or os.path.basename(edges_txt), x, e[:, 0].copy(), e[:, 1].copy())


def load_planetoid(raw_dir: str, name: str) -> LinkDataset:
    """Planetoid ``ind.<name>.{x,tx,allx,graph,test.index}`` (what ``Planetoid(root, name)`` of
    main_disentangled.py:117-123 parses): features = vstack(allx, tx) with the test rows put back in
    index order; edges = the adjacency dict made undirected, duplicates and self-loops removed.
    No feature standardisation for these datasets (main_disentangled.py:119-123)."""
    def rd(suffix):
        with open(os.path.join(raw_dir, f"ind.{name}.{suffix}"), "rb") as f:
            return pickle.load(f, encoding="latin1")
    allx, tx, graph = rd("allx"), rd("tx"), rd("graph")
    test_idx = np.array([int(v) for v in open(os.path.join(raw_dir, f"ind.{name}.test.index")).read().split()])
    order = np.sort(test_idx)
    n = max(int(order.max()) + 1, allx.shape[0] + tx.shape[0])
    x = np.zeros((n, allx.shape[1]), dtype=np.float32)
    x[: allx.shape[0]] = allx.toarray()
    x[test_idx] = tx.toarray()                       # citeseer has isolated test ids: rows stay zero
    src = np.array([u for u, nb in graph.items() for _v in nb], dtype=np.int64)
    dst = np.array([v for _u, nb in graph.items() for v in nb], dtype=np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * n + dst, dst * n + src]))
    return LinkDataset(name, x, key // n, key % n)


def load_fb100(mat_path: str, name: str | None = None, standardise: bool = True) -> LinkDataset:
    """facebook100 ``.mat`` (load_data.py:11-19, other_hetero_datasets.py:131-154): ``A`` (sparse adjacency) and
    ``local_info``; features = one-hot of every metadata column except gender (column 1)."""
    import scipy.io
    mat = scipy.io.loadmat(mat_path)
    A = mat["A"].tocoo()
    meta = mat["local_info"].astype(np.int64)
    cols = np.hstack([meta[:, :1], meta[:, 2:]])
    feats = []
    for c in range(cols.shape[1]):
        vals = np.unique(cols[:, c])
        onehot = (cols[:, c][:, None] == vals[None, :]).astype(np.float32)
        # sklearn.label_binarize: one column for two classes, a zero column for a single class
        feats.append(onehot if vals.size > 2 else onehot[:, 1:] if vals.size == 2 else np.zeros_like(onehot))
    x = np.hstack(feats).astype(np.float32)
    return LinkDataset(name or os.path.basename(mat_path), standardise_rows(x) if standardise else x,
                       A.row.astype(np.int64), A.col.astype(np.int64))


def load_twitch(lang_dir: str, lang: str, standardise: bool = True) -> LinkDataset:
    """twitch-e ``musae_<LANG>_{edges.csv,features.json,target.csv}`` (load_data.py:21-70); the script then
    appends the reversed edge rows (main_disentangled.py:114-116)."""
    ids = []
    seen = set()
    with open(os.path.join(lang_dir, f"musae_{lang}_target.csv")) as f:
        r = csv.reader(f)
        next(r)
        for row in r:
            nid = int(row[5])
            if nid not in seen:
                seen.add(nid)
                ids.append(nid)
    n = len(ids)
    with open(os.path.join(lang_dir, f"musae_{lang}_edges.csv")) as f:
        r = csv.reader(f)
        next(r)
        e = np.array([[int(a), int(b)] for a, b in r], dtype=np.int64)
    with open(os.path.join(lang_dir, f"musae_{lang}_features.json")) as f:
        j = json.load(f)
    x = np.zeros((n, 3170), dtype=np.float32)
    for node, feats in j.items():
        if int(node) < n:
            x[int(node), np.array(feats, dtype=np.int64)] = 1
    x = x[:, x.sum(axis=0) != 0]
    src = np.concatenate([e[:, 0], e[:, 1]])
    dst = np.concatenate([e[:, 1], e[:, 0]])
    return LinkDataset(f"twitch-{lang}", standardise_rows(x) if standardise else x, src, dst)


def load_webkb(raw_dir: str, name: str, standardise: bool = True) -> LinkDataset:
    """texas / cornell / wisconsin (``WebKB(root, name)`` of main_disentangled.py:69-71, 91-96): the geom-gcn text
    pair ``out1_node_feature_label.txt`` / ``out1_graph_edges.txt``; edge rows coalesced (sorted by (src, dst),
    duplicates dropped), directed and self-loops kept — what ``<name>/processed/data.pt`` of the reference holds."""
    ds = load_geom_gcn(os.path.join(raw_dir, "out1_graph_edges.txt"), os.path.join(raw_dir, "out1_node_feature_label.txt"),
                       name=name, standardise=standardise)
    n = ds.n_nodes
    key = np.unique(ds.src * n + ds.dst)
    return LinkDataset(name, ds.x, key // n, key % n)


def load_amazon_npz(path: str, name: str = "photo", standardise: bool = True) -> LinkDataset:
    """Amazon photo ``amazon_electronics_photo.npz`` (``Amazon(root, name)`` of main_disentangled.py:66-68, 85-90):
    attributes and adjacency as CSR triplets; attributes binarised (> 0 -> 1), self-loops removed, edge rows made
    undirected and coalesced.  The raw blob is absent from the reference tree: format pinned by a synthetic file only."""
    import scipy.sparse as sp
    with np.load(path, allow_pickle=True) as f:
        x = sp.csr_matrix((f["attr_data"], f["attr_indices"], f["attr_indptr"]), shape=tuple(f["attr_shape"])).toarray()
        adj = sp.csr_matrix((f["adj_data"], f["adj_indices"], f["adj_indptr"]), shape=tuple(f["adj_shape"])).tocoo()
    x = (np.asarray(x) > 0).astype(np.float32)
    n = x.shape[0]
    src, dst = adj.row.astype(np.int64), adj.col.astype(np.int64)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.unique(np.concatenate([src * n + dst, dst * n + src]))
    return LinkDataset(name, standardise_rows(x) if standardise else x, key // n, key % n)


def load_deezer(mat_path: str, standardise: bool = True) -> LinkDataset:
    """deezer-europe ``.mat`` (other_hetero_datasets.py:156-173): ``A.nonzero()`` as the edge rows, ``features`` dense.
    The blob is absent from the reference tree: format pinned by a synthetic file only."""
    import scipy.io
    mat = scipy.io.loadmat(mat_path)
    row, col = mat["A"].nonzero()
    feats = mat["features"]
    x = np.asarray(feats.todense() if hasattr(feats, "todense") else feats, dtype=np.float32)
    return LinkDataset("deezer-europe", standardise_rows(x) if standardise else x, row.astype(np.int64),
                       col.astype(np.int64))


def read_pyg_data(path: str) -> dict:
    """The tensors of a pickled ``torch_geometric.data.Data`` (``torch.save(data, path)``; a dataset's
    ``processed/data.pt`` holds a tuple whose first item is one) without torch_geometric: its classes are unpickled as
    attribute bags — the file is data, nothing of the library is needed or run."""
    import types
    import torch

    class Bag:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            self.__dict__["state"] = state

    class Unpickler(pickle.Unpickler):
        def find_class(self, mod, name):
            return type(name, (Bag,), {}) if mod.startswith("torch_geometric") else super().find_class(mod, name)

    pm = types.ModuleType("pickle")
    pm.Unpickler, pm.load = Unpickler, (lambda f, **k: Unpickler(f, **k).load())
    obj = torch.load(path, pickle_module=pm, weights_only=False, map_location="cpu")
    data = obj[0] if isinstance(obj, (tuple, list)) else obj
    store = data.state
    store = store["_store"].state if "_store" in store else store
    return dict(store.get("_mapping", store))


def load_arxiv_year_mini(path: str, name: str | None = None, standardise: bool = True) -> LinkDataset:
    """``mini/year<id>.pt`` (main_disentangled.py:124-129): a pickled PyG ``Data`` with ``x`` and ``edge_index``; rows
    standardised, edge rows as stored."""
    m = read_pyg_data(path)
    x = m["x"].numpy().astype(np.float32)
    e = m["edge_index"].numpy().astype(np.int64)
    return LinkDataset(name or os.path.basename(path), standardise_rows(x) if standardise else x, e[0].copy(), e[1].copy())


def save_binary(ds: LinkDataset, path: str) -> None:
    np.savez_compressed(path, x=ds.x, src=ds.src.astype(np.int64), dst=ds.dst.astype(np.int64), name=np.array(ds.name))


def load_binary(path: str) -> LinkDataset:
    with np.load(path, allow_pickle=False) as d:
        return LinkDataset(str(d["name"]), d["x"].astype(np.float32), d["src"].astype(np.int64),
                           d["dst"].astype(np.int64))
