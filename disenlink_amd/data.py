"""Seeded synthetic stand-ins for the graphs BASELINE.json names.

The reference's data files do not travel to the GPU box and most of them are absent from the
reference tree anyway, so every benchmark input is regenerated from a seed:
a Chung-Lu style graph whose node count, directed-row count and heavy-tailed degree law match
the named dataset, N(0,1) features standardised per row like main_disentangled.py:99.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# name: (N, directed edge rows as loaded, self-loop rows, median degree, mean sym degree, max sym degree, F)
SPECS = {
    # data/squirrel/geom_gcn/raw/out1_graph_edges.txt: 217,073 rows, 140 self loops -> 396,846 sym nnz
    "squirrel": dict(N=5201, rows=217_073, loops=140, median=17.0, max_deg=1904, F=128),
    # data_pre_false/chameleon/raw/chameleon.npz: 72,202 rows incl. duplicates -> 62,792 unique
    "chameleon": dict(N=2277, rows=36_101, loops=50, median=12.0, max_deg=732, F=128),
    "cora": dict(N=2708, rows=5_278, loops=0, median=3.0, max_deg=168, F=1433),
    # external figures: Penn94 ~2.72M directed nnz, snap-patents ~13.98M directed edges
    "penn94": dict(N=41_554, rows=1_362_229, loops=0, median=38.0, max_deg=4410, F=128),
    "snap_patents": dict(N=2_923_922, rows=13_975_788, loops=0, median=5.0, max_deg=800, F=269),
}


@dataclass
class SyntheticGraph:
    name: str
    n_nodes: int
    src: np.ndarray      # directed edge rows as a loader would return them (edge_index[0])
    dst: np.ndarray
    n_feat: int
    seed: int

    FEATURE_BLOCK = 16384    # rows per independently seeded block

    def features(self, rows: tuple[int, int] | None = None) -> np.ndarray:
        """x [N,F] ~ N(0,1), then (x - mean_row) / std_row with the unbiased std (main_disentangled.py:99).
        ``rows=(r0, r1)``: only those rows.  Every block of FEATURE_BLOCK rows has its own counter-based stream
        (Philox keyed by seed and block index), so a rank of a sharded run generates exactly its own rows — the values
        do not depend on who generates them or on how many ranks there are."""
        r0, r1 = (0, self.n_nodes) if rows is None else rows
        if not (0 <= r0 <= r1 <= self.n_nodes):
            raise ValueError("rows outside [0, n_nodes]")
        out = np.empty((r1 - r0, self.n_feat), dtype=np.float32)
        B = self.FEATURE_BLOCK
        for b in range(r0 // B, (r1 + B - 1) // B if r1 > r0 else r0 // B):
            lo, hi = b * B, min((b + 1) * B, self.n_nodes)
            rng = np.random.Generator(np.random.Philox(key=[self.seed + 1, b]))
            x = rng.standard_normal((hi - lo, self.n_feat), dtype=np.float32)
            x = (x - x.mean(axis=1, keepdims=True)) / x.std(axis=1, ddof=1, keepdims=True)
            a, e = max(lo, r0), min(hi, r1)
            out[a - r0:e - r0] = x[a - lo:e - lo]
        return out


def real_edge_graph(name: str, seed: int = 0) -> SyntheticGraph:
    """`<name>_real`: the REAL edge rows of a dataset shipped as a parity fixture (tests/golden/real_<name>.npz, data
    only) with seeded features — e.g. the geom-gcn squirrel edge list (217,073 rows) the benchmark is quoted on."""
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"real_{name}.npz")
    with np.load(path) as g:
        edges = g["edges"].astype(np.int64)
        n = int(g["feat_shape"][0]) if "feat_shape" in g else int(g["features"].shape[0])
    return SyntheticGraph(f"{name}_real", n, edges[:, 0].copy(), edges[:, 1].copy(), SPECS[name]["F"], seed)


def synthetic_graph(name: str, seed: int = 0, scale: float = 1.0) -> SyntheticGraph:
    """Graph with the node count / row count / degree skew of `name` (optionally scaled down)."""
    if name.endswith("_real"):
        return real_edge_graph(name[:-5], seed)
    sp = SPECS[name]
    N = max(8, int(round(sp["N"] * scale)))
    rows = max(8, int(round(sp["rows"] * scale)))
    loops = int(round(sp["loops"] * scale))
    rng = np.random.default_rng(seed)
    mean_deg = 2.0 * rows / N
    # lognormal weights: median and mean fixed -> sigma from mean/median, clipped at the max degree
    mu = np.log(sp["median"])
    sigma = np.sqrt(max(2.0 * np.log(max(mean_deg / sp["median"], 1.05)), 0.1))
    w = np.exp(mu + sigma * rng.standard_normal(N))
    w = np.minimum(w, min(sp["max_deg"], N - 1))
    prob = w / w.sum()
    want = rows - loops
    keys = np.zeros(0, dtype=np.int64)
    while keys.size < want:
        need = int((want - keys.size) * 1.3) + 64
        i = rng.choice(N, size=need, p=prob)
        j = rng.choice(N, size=need, p=prob)
        ok = i != j
        lo, hi = np.minimum(i[ok], j[ok]), np.maximum(i[ok], j[ok])
        keys = np.unique(np.concatenate([keys, lo * N + hi]))
    keys = rng.permutation(keys)[:want]
    a, b = keys // N, keys % N
    flip = rng.random(want) < 0.5                      # one stored direction per undirected edge
    src = np.where(flip, a, b)
    dst = np.where(flip, b, a)
    if loops:
        lp = rng.choice(N, size=loops, replace=False)
        src, dst = np.concatenate([src, lp]), np.concatenate([dst, lp])
    perm = rng.permutation(src.size)
    return SyntheticGraph(name, N, src[perm].astype(np.int64), dst[perm].astype(np.int64), sp["F"], seed)
