"""Drop-in ``Disentangle`` module: same constructor, ``forward(x, adj)`` signature and
``state_dict`` keys as the reference's ``model.Disentangle`` (model.py:91-114), with the
routing / aggregation / scoring path running on libdisenlink_hip.so.

    from disenlink_amd.model import Disentangle          # instead of: from model import Disentangle
    model = Disentangle(nfeat, nhidden, nembed, nfactor=K, beta=b, t=t).to(device)
    emb, a_pred = model(x, adj_sym)                       # main_disentangled.py:194

For graphs where ``[N,N]`` cannot exist, ``forward_pairs(x, graph, pairs)`` scores a pair
list instead; the reference has no counterpart for it.
"""
from __future__ import annotations

import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import Graph, PairList


class Factor(nn.Module):
    """One Linear(F -> d); used when nhid == 1 (model.py:7-15, :94-95)."""

    def __init__(self, nfeat, nhid):
        super().__init__()
        self.mlp = nn.Linear(nfeat, nhid)

    def forward(self, x):
        return self.mlp(x)


class Factor2(nn.Module):
    """Linear(F -> nmid) -> ReLU -> Linear(nmid -> d) (model.py:16-27, :96-97)."""

    def __init__(self, nfeat, nmid, nhid):
        super().__init__()
        self.mlp1 = nn.Linear(nfeat, nmid)
        self.mlp2 = nn.Linear(nmid, nhid)

    def forward(self, x):
        return self.mlp2(F.relu(self.mlp1(x)))


class Disentangle(nn.Module):
    def __init__(self, nfeat, nhid, nebed, nfactor, beta, t=1, table_dtype=torch.float32, projection="auto",
                 use_torch_ops=False):
        """Extensions (the reference has neither):
        ``table_dtype``: storage type of the gathered Z / H tables in ``forward_pairs`` — torch.float32
        (reference precision) or torch.bfloat16 (half the gather bytes, fp32 arithmetic and gradients).
        ``projection``: "mfma" = the fused matrix-core kernels of libdisenlink_hip.so (forward and backward; fp32
        results — layer 1 and the dW1 contraction run as six exact bf16 products per term; the hidden
        layer is written once, transposed, for the backward while it fits a quarter of the device memory (at most
        64 GiB) and recomputed beyond that),
        "library" = library GEMMs (rocBLAS through torch), "auto" = the kernels wherever they support the factor width
        (any d <= 128; widths other than 32 / 64 / 128 run at the next of those): measured equal or faster than the library path at every feature width (squirrel epoch
        1.64 vs 1.75 ms at F=128, 1.93 vs 2.06 at 512, 2.17 vs 2.36 at 1024, 2.74 vs 3.20 at 2089; real Cora
        (F=1433) 1.32 vs 1.60; tools/epoch_time.py), and they never materialise the [N,K,nhid] activations.
        ``use_torch_ops``: ``forward_pairs`` goes through the REGISTERED operators ``torch.ops.disenlink.*``
        (disenlink_amd/torch_ops.py: schema + fake + autograd via torch.library over the same C ABI) instead of the
        ctypes-calling autograd.Functions — same kernels and bits; what it buys is dispatcher visibility:
        ``torch.compile(model.forward_pairs)`` traces the whole step without graph breaks."""
        super().__init__()
        self.use_torch_ops = bool(use_torch_ops)
        if projection not in ("auto", "mfma", "library"):
            raise ValueError("projection must be 'auto', 'mfma' or 'library'")
        self.table_dtype = table_dtype
        self.projection = projection
        # creation order == the reference's (model.py:93-99), so a seeded init draws the same stream
        if nhid == 1:
            factors = [Factor(nfeat, nebed) for _ in range(nfactor)]
        else:
            factors = [Factor2(nfeat, nhid, nebed) for _ in range(nfactor)]
        for i, f in enumerate(factors):
            self.add_module("factor_{}".format(i), f)
        self.single_layer = nhid == 1
        self.temperature = t
        self.nfactor = nfactor
        self.nebed = nebed
        self.beta = beta
        self._graph_cache = None
        self._dense_plan = ops.DensePairPlanCache()      # backward of the dense link_pred: this module's own cache
        self._stacked = {}
        self._restack()

    # ------------------------------------------------------------------ stacked parameter storage
    # The K factor modules keep their own Parameters (and state_dict keys), but every group of K same-shaped
    # parameters shares ONE contiguous [K, ...] buffer: parameter i's .data is the view buf[i].  The projection
    # kernel reads the buffers directly — no torch.stack copies per call — and Adam / load_state_dict, which
    # update parameters in place, keep the buffers current.
    def _param_groups(self):
        names = ("mlp",) if self.single_layer else ("mlp1", "mlp2")
        for name in names:
            for attr in ("weight", "bias"):
                yield (name, attr), [getattr(getattr(f, name), attr) for f in self.factors]

    def _param_slots(self):
        """(key, [(owning module's _parameters dict, attribute name)]) per group: where the LIVE Parameter objects are
        registered — the fast path checks identity against these slots, so a replaced Parameter object
        (load_state_dict(assign=True), setattr, weight tying) is seen."""
        names = ("mlp",) if self.single_layer else ("mlp1", "mlp2")
        for name in names:
            for attr in ("weight", "bias"):
                yield (name, attr), [(getattr(f, name)._parameters, attr) for f in self.factors]

    def _restack(self):
        # (owner's _parameters dict, attribute, parameter, expected data pointer, expected shape) of every view
        self._stacked_check = []
        for key, slots in self._param_slots():
            ps = [reg[attr] for reg, attr in slots]
            buf = torch.stack([p.data for p in ps]).contiguous()
            for i, (p, (reg, attr)) in enumerate(zip(ps, slots)):
                p.data = buf[i]
                self._stacked_check.append((reg, attr, p, p.data_ptr(), tuple(p.shape)))
            self._stacked[key] = buf

    def _apply(self, fn, *args, **kwargs):                     # .to(device) / .float() replace .data: re-stack
        out = super()._apply(fn, *args, **kwargs)
        self._restack()
        return out

    def snapshot_state(self):
        """deepcopy(state_dict()) (main_disentangled.py:209) with one clone per shared buffer instead of one per
        parameter; the result loads back with load_state_dict like any state_dict."""
        if self._stacked_params() is None:
            from copy import deepcopy
            return deepcopy(self.state_dict())
        clones = {key: buf.clone() for key, buf in self._stacked.items()}
        out = {}
        for i in range(self.nfactor):
            for (name, attr), buf in clones.items():
                out[f"factor_{i}.{name}.{attr}"] = buf[i]
        return {k: out[k] for k in self.state_dict().keys()}

    def _stacked_params(self):
        """The flat parameter list if every parameter still aliases its place in the shared buffers, else None.  Runs
        every forward: compares each parameter's data pointer with the one recorded when the buffers were built (indexing
        the buffers here — 4K tiny view tensors per call — cost the eager loop ~50 us of host time per epoch)."""
        chk = self.__dict__.get("_stacked_check")
        if chk and all(reg.get(attr) is p and p.data_ptr() == ptr and tuple(p.shape) == shape
                       for reg, attr, p, ptr, shape in chk):
            return [c[2] for c in chk]
        # pointers moved (a deep copy, an unpickled module) or a Parameter object was replaced: look at the LIVE
        # parameters and the buffers themselves, and record anew
        fresh = []
        for key, slots in self._param_slots():
            buf = self._stacked.get(key)
            ps = [reg.get(attr) for reg, attr in slots]
            if buf is None or any(p is None or p.data_ptr() != buf[i].data_ptr() or p.shape != buf[i].shape
                                  for i, p in enumerate(ps)):
                return None
            fresh += [(reg, attr, p, p.data_ptr(), tuple(p.shape)) for p, (reg, attr) in zip(ps, slots)]
        self._stacked_check = fresh
        return [c[2] for c in fresh]

    def restack(self):
        """Re-establish the shared [K, ...] buffers from the LIVE parameters (after load_state_dict(assign=True) or any
        other replacement of Parameter objects, which leaves project() on the slower stacking path until this is
        called).  Optimisers built over the old Parameter objects must be rebuilt, as torch requires after assign."""
        self._restack()
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """torch's load_state_dict; with ``assign=True`` — which REPLACES the Parameter objects instead of copying into
        them — the shared buffers are rebuilt from the new parameters afterwards, so the kernels, snapshot_state() and
        the live state_dict keep reading the same storage."""
        out = super().load_state_dict(state_dict, strict=strict, assign=assign)
        if assign:
            self._restack()
        return out

    def train(self, mode: bool = True):
        """No layer of this model depends on the mode (main_disentangled.py:193,201 call train()/eval() every
        epoch as no-ops): set the flag without walking the K factor modules."""
        self.training = mode
        return self

    @property
    def factors(self):
        return [getattr(self, "factor_{}".format(i)) for i in range(self.nfactor)]

    # ------------------------------------------------------------------ projection (model.py:106)
    def project(self, x: torch.Tensor) -> torch.Tensor:
        """Z [N,K,d] = K independent MLPs of x.  On the GPU: the fused MFMA kernels of libdisenlink_hip.so (d <= 128).
        The library-GEMM form (one wide GEMM + one K-batched GEMM: the reference's own ops) runs ONLY when asked for —
        ``Disentangle(projection="library")``, kept for timing comparisons — or for CPU tensors, which no product path
        serves (the CPU tests of the host logic and of the sharding choreography use it with the oracle as the backend);
        a CUDA tensor whose shape the kernels do not serve raises instead of falling back."""
        fs = self.factors
        K, d = self.nfactor, self.nebed
        if x.is_cuda and self.projection != "library" and not (x.dtype == torch.float32 and ops.project_supported(d)):
            raise ops._lib.DisenlinkHipError(
                f"the projection kernels serve fp32 features and factor widths d <= 128 (got {x.dtype}, d = {d}); "
                "there is no eager fallback on the GPU — construct the module with projection=\"library\" to use the library GEMMs")
        use_kernel = x.is_cuda and self.projection != "library"
        if use_kernel:
            flat = self._stacked_params()
            if flat is not None:                                # zero-copy: the kernel reads the shared buffers
                st = self._stacked
                bufs = ((st[("mlp", "weight")], st[("mlp", "bias")], None, None) if self.single_layer else
                        (st[("mlp1", "weight")], st[("mlp1", "bias")], st[("mlp2", "weight")], st[("mlp2", "bias")]))
                from . import native
                if native.project_ok(x, d, self.single_layer):   # the same kernels from a C++ autograd node (no Python in the backward)
                    return native.project_stacked(x, bufs, flat)
                return ops.ProjectStacked.apply(x, bufs, K, *flat)
            # parameters were re-pointed by the caller: stack them (one copy per call)
            if self.single_layer:
                return ops.Project.apply(x, torch.stack([f.mlp.weight for f in fs]),
                                         torch.stack([f.mlp.bias for f in fs]), None, None)
            return ops.Project.apply(x, torch.stack([f.mlp1.weight for f in fs]), torch.stack([f.mlp1.bias for f in fs]),
                                     torch.stack([f.mlp2.weight for f in fs]), torch.stack([f.mlp2.bias for f in fs]))
        if self.single_layer:
            W = torch.cat([f.mlp.weight for f in fs], dim=0)             # [K*d, F]
            b = torch.cat([f.mlp.bias for f in fs], dim=0)
            return F.linear(x, W, b).view(-1, K, d)
        W1 = torch.cat([f.mlp1.weight for f in fs], dim=0)               # [K*nhid, F]
        b1 = torch.cat([f.mlp1.bias for f in fs], dim=0)
        hid = F.relu(F.linear(x, W1, b1)).view(x.shape[0], K, -1)        # [N,K,nhid]
        W2 = torch.stack([f.mlp2.weight for f in fs], dim=0)             # [K,d,nhid]
        b2 = torch.stack([f.mlp2.bias for f in fs], dim=0)               # [K,d]
        Z = torch.einsum("nkh,kdh->nkd", hid, W2) + b2
        return Z.contiguous()

    # ------------------------------------------------------------------ graph cache
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_graph_cache"] = None              # holds a weak reference: not copyable / picklable, and only a cache
        state["_dense_plan"] = ops.DensePairPlanCache()     # a copy starts without a plan (declare the masks again)
        return state

    def set_loss_pairs(self, *supports, n_nodes: int | None = None):
        """Declare where the caller takes its loss on the dense ``link_pred`` of ``forward(x, adj)``: dense [N,N] masks
        (the reference's ``pos_train_adj``, ``neg_train_adj``; entries != 0 count, main_disentangled.py:167-190) and / or
        ``(rows, cols)`` index tuples (then give ``n_nodes`` unless a mask comes along).  Their union becomes the pair
        plan of the dense backward (dl_score_allpairs_bwd), built once, on the module's device.  Without it the plan is
        learnt from the gradients (it grows until it covers the masks; ops.DensePairPlanCache).
        ``set_loss_pairs()`` with no argument forgets a declared set."""
        if not supports:
            self._dense_plan.clear()
            return self
        for sup in supports:
            if torch.is_tensor(sup) and sup.dim() == 2 and n_nodes is None:
                n_nodes = sup.shape[0]
        if n_nodes is None:
            raise ValueError("index pairs alone do not say how many nodes there are: pass n_nodes=")
        self._dense_plan.set_pairs(int(n_nodes), next(self.parameters()).device, *supports)
        return self

    def assume_static_loss_masks(self, *masks, static: bool = True):
        """``assume_static_loss_masks(pos_train_adj, neg_train_adj)``: the caller promises that every step's loss is
        taken inside these masks (the reference builds them once per run, main_disentangled.py:167-190).  The dense
        backward then runs on their support without any host read: the subset check "no gradient outside the plan"
        stays on the device and turns the gradients into NaN — not into something silently wrong — should the promise
        be broken.  The masks are REQUIRED (here or through set_loss_pairs before): the support of a loss cannot be
        inferred from a gradient, whose non-zero set moves with fp32 sigmoid saturation.
        ``assume_static_loss_masks(static=False)`` returns to the validated mode (one 16-byte read per backward)."""
        if len(masks) == 1 and isinstance(masks[0], bool):        # round-2 spelling: assume_static_loss_masks(False)
            static, masks = masks[0], ()
        if masks:
            self.set_loss_pairs(*masks)
        if static and not self._dense_plan.from_masks:
            raise ValueError("assume_static_loss_masks needs the loss masks: assume_static_loss_masks(pos_train_adj, "
                             "neg_train_adj), or call set_loss_pairs(...) first")
        self._dense_plan.static = bool(static)
        return self

    def _graph_for(self, adj: torch.Tensor) -> Graph:
        """CSR + plans of a dense adjacency, built once per adjacency tensor: keyed on the tensor OBJECT (weak
        reference) and its version counter — an address can be reused by a different tensor, an object cannot."""
        hit = self._graph_cache
        if hit is None or hit[0]() is not adj or hit[1] != adj._version:
            hit = (weakref.ref(adj), adj._version, Graph.from_dense(adj, row_bytes=self.nfactor * self.nebed * 4))
            self._graph_cache = hit
        return hit[2]

    # ------------------------------------------------------------------ forward
    def forward(self, x, adj):
        """(emb [N,K*d], link_pred [N,N]) exactly as model.py:105-114; ``adj`` is the dense adj_sym
        (or a prebuilt ``Graph``)."""
        graph = adj if isinstance(adj, Graph) else self._graph_for(adj)
        Z = self.project(x)
        H = ops.RouteAggregate.apply(Z, graph, float(self.beta), float(self.temperature))
        link_pred = ops.ScoreAllPairs.apply(Z, H, float(self.temperature), self._dense_plan)
        # (a Tensor whose indexing reports the entries taken to this module's pair-plan cache: ops.LinkPred)
        return H.view(H.shape[0], -1), ops.as_link_pred(link_pred, self._dense_plan)

    def forward_pairs_loss(self, x, graph: Graph, pairs: PairList, label, weight):
        """(emb [N,K*d], prob [P], loss) with loss = sum_q weight BCE(prob, label) (main_disentangled.py:195 on a pair
        list; metrics.pair_bce_weights): the training step's scorer runs forward and backward in one pass
        (ops.HotPathPairsLoss).  Falls back to forward_pairs + the fused loss where no tuned kernel exists."""
        Z = self.project(x)
        dt = ops._lib.DL_F32 if self.table_dtype == torch.float32 else ops._lib.DL_BF16
        one_pass = ops.one_pass_scorer_wanted(self.table_dtype, Z.shape[0], Z.shape[1], Z.shape[2])     # the rule and its numbers: there
        if one_pass and ops.score_pairs_train_supported(pairs, Z.shape[1], Z.shape[2], dt):
            if Z.dtype == torch.float32 and graph.n_rows == graph.n_nodes:
                from . import native                            # the compiled binding: the same step as ONE C++ autograd node
                if native.available():
                    H, prob, loss = native.hot_path_pairs_loss(Z, graph, pairs, float(self.beta), float(self.temperature),
                                                               label, weight, self.table_dtype)
                    return H.view(H.shape[0], -1), prob, loss
            H, prob, loss = ops.HotPathPairsLoss.apply(Z, graph, pairs, float(self.beta), float(self.temperature),
                                                       self.table_dtype, label, weight)
            return H.view(H.shape[0], -1), prob, loss
        H, prob = ops.HotPathPairs.apply(Z, graph, pairs, float(self.beta), float(self.temperature), self.table_dtype)
        return H.view(H.shape[0], -1), prob, ops.PairBCE.apply(prob, label, weight)

    def forward_pairs(self, x, graph: Graph, pairs: PairList):
        """(emb [N,K*d], prob [P]) — the same model evaluated on a pair list only."""
        if self.use_torch_ops and not self.single_layer and self.table_dtype == torch.float32 and x.is_cuda \
                and ops.project_supported(self.nebed):
            from . import torch_ops
            return torch_ops.forward_pairs(self, x, torch_ops.register_graph(graph), torch_ops.register_pairs(pairs))
        Z = self.project(x)
        H, prob = ops.HotPathPairs.apply(Z, graph, pairs, float(self.beta), float(self.temperature), self.table_dtype)
        return H.view(H.shape[0], -1), prob
