"""D-VAE encoder variants of the same hot path: `DAGNN_NA` (= `dvae/dagnn.py:18-184`, class `DAGNN`
there) for ENAS neural-architecture DAGs and `DAGNN_BN` (`dvae/dagnn_bn.py:19-177`) for
Bayesian-network DAGs.

Constructor arguments, parameter names and shapes follow the reference (including the decoder-side
parameters of `DVAE_PYG` / `DVAE_BN_PYG`, `dvae/models_pyg.py:18-85,539-560`, which this build
does not use but keeps so that `state_dict`s are interchangeable).  `forward(G)` / `encode(list)`
run the layer-by-layer message passing in HIP; the igraph teacher-forced decoder and loss are out
of scope.
"""
from __future__ import annotations

import copy
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import constants as K
from . import engine
from .core import DerivedCache, default_schedule, derive_cell, pack_lockstep, run_stack
from .data import GraphBatch

ENCODE_FUSED = int(__import__("os").environ.get("DAGNN_AMD_ENCODE_FUSED", "1"))   # 1: evaluation passes of the encoders as ONE library call (csrc/encode.hip)
OWN_LINEAR_MAX = 1 << 22   # multiply-adds up to which the final Linear of an evaluation pass runs on dagnn_gemm_nt_bias (see forward)
from .model import _EdgeAttnParams, _SelfAttnParams


class _NoParams(object):
    wea = False   # no edge encoder (num_rels = 1 in the D-VAE models)


class _GatedParams(object):
    wea = False

    def __init__(self, gate, mapper):
        self.gate, self.mapper = gate, mapper   # gate: Sequential(Linear, Sigmoid); mapper: Linear(bias=False)


class _AggView(object):
    """What `dagnn_amd.variants` reads from a model, for a D-VAE encoder with agg in {add, max}: GRU cells, values = the
    cell's own states, no edge features, and - unlike the ogbg model - one AggConv PER direction (`reverse=True` for the
    second, dvae/dagnn.py:66-70), so the messages land on the frontier in both."""
    agg_x = False
    recurr = 1
    agg_attn = False
    agg_attn_x = False
    shared_agg_flow = False

    vid_nodes = 0

    def __init__(self, m):
        self._m = m
        self.agg, self.hidden_dim, self.num_layers, self.dirs = m.agg, m.hidden_dim, m.num_layers, m.dirs
        self.emb_dim = m.cells_0[0].weight_ih.shape[1]   # the node inputs are the one-hot vertex types (nvt wide)
        if m.agg == K.NA_GATED_SUM:
            # GatedSumConv over hs_j = [state ; one-hot vertex id] (dvae/dagnn.py:124-137,269-299): gate / mapper are Linear(hs +
            # num_nodes, hs); csrc/variants.hip takes the one-hot columns as a per-vertex-id bias of the per-node projections
            self.vid_nodes = m.num_nodes
            self.node_aggr_0 = [_GatedParams(m.gate_forward[l], m.mapper_forward[l][0]) for l in range(m.num_layers)]
            self.node_aggr_1 = [_GatedParams(m.gate_backward[l], m.mapper_backward[l][0]) for l in range(m.num_layers)]
            return
        self.node_aggr_0 = [_NoParams() for _ in range(m.num_layers)]
        self.node_aggr_1 = [_NoParams() for _ in range(m.num_layers)]

    def __getattr__(self, name):   # cells_0 / cells_1, training, ...
        return getattr(self.__dict__["_m"], name)

    def parameters(self):
        return self._m.parameters()


class _CellView:
    """The four GRUCell tensors `engine.iprop_step` reads, without the module around them."""
    __slots__ = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")

    def __init__(self, w_ih, w_hh, b_ih, b_hh):
        self.weight_ih, self.weight_hh, self.bias_ih, self.bias_hh = w_ih, w_hh, b_ih, b_hh


def _iprop_dense(values, pred_vid, w_key, vid_bias, H, X, cells):
    """The decoder-side step as dense device-side torch ops, the form its reverse pass differentiates (`dvae/dagnn.py:187-239`):
    padded slots score 0 (their key is a zero row, and the query term `w_q.q + b` is common to every slot of a soft-max
    row, so it cancels - the query half of `attn_lin` and its bias get exact zero gradients), the aggregate of the
    layer-0 states feeds every stacked layer."""
    B = X.shape[0]
    if H is None:
        if values is None or values.shape[1] == 0:
            H = X.new_zeros(B, cells[0].weight_hh.shape[1])
        else:
            scores = values @ w_key
            if vid_bias is not None:
                real = pred_vid >= 0
                scores = scores + torch.where(real, vid_bias[pred_vid.clamp(min=0).long()], torch.zeros_like(scores))
            H = torch.einsum("bp,bpj->bj", torch.softmax(scores, dim=-1), values)
    Hv, out = X, []
    for c in cells:
        gi = F.linear(Hv, c.weight_ih, c.bias_ih)
        gh = F.linear(H, c.weight_hh, c.bias_hh)
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = gh.chunk(3, 1)
        r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        Hv = n + z * (H - n)
        out.append(Hv)
    return torch.stack(out, 0)


class _IpropStep(torch.autograd.Function):
    """`_ipropagate_to` under autograd: forward = the ONE HIP launch (`dagnn_iprop_step`), backward = the reverse pass of
    the same step on the device (the step is a handful of [B, hs] products: recomputed densely from the saved inputs and
    differentiated there - no state of the forward launch is kept besides its inputs)."""

    @staticmethod
    def forward(ctx, values, pred_vid, w_key, vid_bias, H, X, L, *flat):
        cells = [_CellView(*flat[4 * l:4 * l + 4]) for l in range(L)]
        ctx.save_for_backward(*[t for t in (values, pred_vid, w_key, vid_bias, H, X) if t is not None], *flat)
        ctx.have = [t is not None for t in (values, pred_vid, w_key, vid_bias, H, X)]
        ctx.L = L
        return engine.iprop_step(values, pred_vid, w_key, vid_bias, H, X, cells)

    @staticmethod
    def backward(ctx, g_states):
        saved = list(ctx.saved_tensors)
        head = [saved.pop(0) if h else None for h in ctx.have]
        values, pred_vid, w_key, vid_bias, H, X = head
        flat = saved
        need = list(ctx.needs_input_grad)
        leaves, slots = [], []

        def leaf(t, slot):
            if t is None or not t.is_floating_point():
                return t
            t = t.detach().requires_grad_(need[slot])
            if need[slot]:
                leaves.append(t)
                slots.append(slot)
            return t

        values, w_key, vid_bias, H = leaf(values, 0), leaf(w_key, 2), leaf(vid_bias, 3), leaf(H, 4)
        flat = [leaf(t, 7 + k) for k, t in enumerate(flat)]
        grads = [None] * (7 + len(flat))
        if leaves:
            with torch.enable_grad():
                cells = [_CellView(*flat[4 * l:4 * l + 4]) for l in range(ctx.L)]
                out = _iprop_dense(values, pred_vid, w_key, vid_bias, H, X, cells)
            for slot, g in zip(slots, torch.autograd.grad(out, leaves, g_states.contiguous(), allow_unused=True)):
                grads[slot] = g
        return tuple(grads)


class _DvaeBase(nn.Module):
    """Parameters of `DVAE_PYG.__init__` (`dvae/models_pyg.py:18-85`), same names and order."""

    def __init__(self, max_n, nvt, START_TYPE, END_TYPE, hs=501, nz=56, bidirectional=False, vid=True,
                 num_layers=1):
        super().__init__()
        self.max_n, self.nvt, self.START_TYPE, self.END_TYPE = max_n, nvt, START_TYPE, END_TYPE
        self.hs, self.nz, self.gs = hs, nz, hs
        self.bidir, self.vid = bidirectional, vid
        self.vs = hs + max_n if vid else hs
        self.num_layers = num_layers
        gru = lambda: nn.ModuleList([nn.GRUCell(nvt if l == 0 else hs, hs) for l in range(num_layers)])  # noqa: E731
        self.grue_forward = gru()
        self.grue_backward = gru()
        self.fc1 = nn.Linear(self.gs, nz)
        self.fc2 = nn.Linear(self.gs, nz)
        self.grud = gru()
        self.fc3 = nn.Linear(nz, hs)
        self.add_vertex = nn.Sequential(nn.Linear(hs, hs * 2), nn.ReLU(), nn.Linear(hs * 2, nvt))
        self.add_edge = nn.Sequential(nn.Linear(hs * 2, hs * 4), nn.ReLU(), nn.Linear(hs * 4, 1))
        self.gate_forward = nn.ModuleList([nn.Sequential(nn.Linear(self.vs, hs), nn.Sigmoid())
                                           for _ in range(num_layers)])
        self.gate_backward = nn.ModuleList([nn.Sequential(nn.Linear(self.vs, hs), nn.Sigmoid())
                                            for _ in range(num_layers)])
        self.mapper_forward = nn.ModuleList([nn.Sequential(nn.Linear(self.vs, hs, bias=False))
                                             for _ in range(num_layers)])
        self.mapper_backward = nn.ModuleList([nn.Sequential(nn.Linear(self.vs, hs, bias=False))
                                              for _ in range(num_layers)])
        if self.bidir:
            self.hv_unify = nn.Sequential(nn.Linear(hs * 2, hs))
            self.hg_unify = nn.Sequential(nn.Linear(self.gs * 2 * num_layers, self.gs))
        self.relu, self.sigmoid, self.tanh = nn.ReLU(), nn.Sigmoid(), nn.Tanh()
        self.logsoftmax1 = nn.LogSoftmax(1)
        self.device = None

    def get_device(self):
        if self.device is None:
            self.device = next(self.parameters()).device
        return self.device

    def _collate_fn(self, G):
        return [copy.deepcopy(g) for g in G]


class _DvaeDagnn(_DvaeBase):
    _use_vids = True

    def _setup(self, emb_dim, hidden_dim, out_dim, num_layers, bidirectional, agg, out_wx, out_pool_all, out_pool,
               dropout, num_nodes):
        self.num_nodes = num_nodes
        self.agg = agg
        self.agg_attn = "attn" in agg
        self.agg_attn_x = "_x" in agg
        self.bidirectional = bidirectional
        self.dirs = [0, 1] if bidirectional else [0]
        self.out_wx = out_wx
        self.output_all = out_pool_all
        self.out_pool = out_pool
        self.emb_dim = emb_dim
        self.hidden_dim = hidden_dim
        self.out_hidden_dim = emb_dim + hidden_dim * num_layers if out_wx else hidden_dim * num_layers
        self._agg_plain = agg in (K.NA_GATED_SUM, K.NA_SUM, K.NA_MAX)
        if not self._agg_plain and agg not in (K.NA_ATTN_H, K.NA_SELF_ATTN_H):
            raise NotImplementedError("D-VAE encoders: agg=%r (attn_h - the reference's default, dvae/train.py:86 -, self_attn_h, "
                                      "gated_sum, add, max)" % (agg,))
        if agg == K.NA_GATED_SUM and not self._use_vids:
            raise NotImplementedError("DAGNN_BN with agg='gated_sum': the reference itself cannot run it - DVAE_BN_PYG re-creates the "
                                      "first layer's mapper / gate with nvt inputs (dvae/models_pyg.py:539-560) and GatedSumConv feeds "
                                      "them hs-wide states (dvae/dagnn_bn.py:289): a shape error in its first message")
        extra = num_nodes if self._use_vids else 0
        pred_dim = hidden_dim + extra
        attn_dim = hidden_dim + extra
        if self._agg_plain:
            # GatedSumConv / AggConv own no parameters of their own here: gated_sum uses the base class's mapper / gate
            # (dvae/dagnn.py:60-65), add / max have none (num_rels = 1: no edge encoder)
            def conv(mapper, gate):   # (same `state_dict` names as the reference's GatedSumConv: it registers the shared modules)
                m = nn.Module()
                if agg == K.NA_GATED_SUM:
                    m.mapper, m.gate = mapper, gate
                return m
            self.node_aggr_0 = nn.ModuleList([conv(self.mapper_forward[l], self.gate_forward[l]) for l in range(num_layers)])
            self.node_aggr_1 = nn.ModuleList([conv(self.mapper_backward[l], self.gate_backward[l]) for l in range(num_layers)])
        elif agg == K.NA_SELF_ATTN_H:   # keys scored alone: `attn_lin` has no query half (dvae/dagnn.py:49-54, 301-312)
            self.node_aggr_0 = nn.ModuleList([_SelfAttnParams(attn_dim, num_relations=1) for _ in range(num_layers)])
            self.node_aggr_1 = nn.ModuleList([_SelfAttnParams(attn_dim, num_relations=1, reverse=True)
                                              for _ in range(num_layers)])
        else:
            self.node_aggr_0 = nn.ModuleList([
                _EdgeAttnParams(emb_dim if l == 0 else attn_dim, pred_dim, num_relations=1, attn_dim=attn_dim)
                for l in range(num_layers)])
            self.node_aggr_1 = nn.ModuleList([
                _EdgeAttnParams(emb_dim if l == 0 else attn_dim, pred_dim, num_relations=1, attn_dim=attn_dim,
                                reverse=True) for l in range(num_layers)])
        # the cells ARE the base class's encoder GRUs (aliased names, dvae/dagnn.py:73-75)
        self.cells_0 = self.grue_forward
        if bidirectional:
            self.cells_1 = self.grue_backward
        self.dropout = nn.Dropout(dropout)
        self.out_linear = nn.Linear(self.out_hidden_dim, out_dim) if num_layers > 1 else None
        self._derived = {}
        self._arenas = {}  # per device: granule buffers of the persistent tail kernel
        self.schedule = default_schedule()  # 'lockstep' (frontier launches) or 'pergraph' (persistent workgroups)

    def _cells(self, fresh: bool = False):
        # (the registries are read directly: `getattr(self, "cells_0")[i].weight_ih` is three trips through
        # nn.Module.__getattr__ - 15 us for the ten tensors of cfg 1, whose forward is host-bound; same objects, always current)
        srcs: List[torch.Tensor] = []
        mods = self._modules
        for d in self.dirs:
            cells, aggrs = mods["cells_%d" % d]._modules, mods["node_aggr_%d" % d]._modules
            for i in range(self.num_layers):
                c, a = cells[str(i)], aggrs[str(i)]._modules["attn_lin"]
                cp = c._parameters
                for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
                    t = cp.get(name)
                    srcs.append(t if t is not None else getattr(c, name))   # (a re-parametrised weight is a plain attribute)
                t = a._parameters.get("weight")
                srcs.append(t if t is not None else a.weight)
        extra = self.num_nodes if self._use_vids else 0

        def make():
            out = {}
            for d in self.dirs:
                for i in range(self.num_layers):
                    c = getattr(self, "cells_%d" % d)[i]
                    a = getattr(self, "node_aggr_%d" % d)[i]
                    dq = self._key_offset(i)
                    out[(d, i)] = derive_cell(c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh, a.attn_lin.weight,
                                              self.hidden_dim, dq, i > 0, None, extra, schedule=self.schedule,
                                              pack=False, stacked=self.num_layers)
            if self.schedule == "lockstep":
                pack_lockstep(out.values())
            return out

        return self._derived.setdefault(self.schedule, DerivedCache()).get(srcs, make, fresh=fresh or self.training)

    def train(self, mode: bool = True):
        """Mode switches drop the derived-weight caches (see core.DerivedCache)."""
        # (the `add` / `max` aggregators derive their weights on the cached view object - variants._derive keeps its DerivedCache
        # on the module it is handed: fused optimizers and `.data` writes do not bump `_version`, so a train() / eval() switch
        # must drop that cache too, exactly as DAGNN.train() does for its own)
        self.invalidate_caches()
        return super().train(mode)

    # ---- hooks of autograd.Recurrence
    @property
    def _vid_nodes(self) -> int:
        return self.num_nodes if self._use_vids else 0

    def _key_offset(self, i: int) -> int:
        """Where the key half of `attn_lin.weight` starts: behind the query half - which `self_attn_h` does not have."""
        if self.agg == K.NA_SELF_ATTN_H:
            return 0
        return self.emb_dim if i == 0 else self.hidden_dim + self._vid_nodes

    def _static_scores(self, x, cells):
        return None

    def check(self) -> None:
        """Blocking check for device-side failures of every pass launched so far (`core.check_arenas`): call it where
        the outputs of the LAST forward of a loop are consumed - the non-blocking poll inside `forward` only reports
        earlier passes."""
        from .core import check_arenas
        check_arenas(self)

    def _arena_for(self, x, role="forward"):
        key = (role, x.device, engine._stream(x))   # one arena per stream: passes on different streams may overlap
        arena = self._arenas.get(key)
        if arena is None:
            arena = engine.GranuleArena()
            # passes of further streams (micro-batches in flight) start their workgroup packing two XCDs further on
            arena.xcd_first = 2 * sum(1 for k in self._arenas if k[0] == role and k[1] == x.device) % 8
            self._arenas[key] = arena
        return arena

    def _readout(self, plan, B, x, h):
        """End vertex of every graph for d = 0, start vertex for d = 1 (dvae/dagnn.py:147-161, dagnn_bn.py:138-152)."""
        L, H, nn_ = self.num_layers, self.hidden_dim, self.num_nodes
        hcat = torch.empty(B, len(self.dirs) * L * H, dtype=torch.float32, device=x.device)
        jobs = [(h[0][i], nn_ - 1, i * H) for i in range(L)]
        if self.bidirectional:
            jobs += [(h[1][i], 0, (L + i) * H) for i in range(L)]
        if len(jobs) <= 16:
            engine.gather_rows_batch(jobs, B, nn_, hcat)   # one launch for every (direction, stacked layer)
        else:
            for t, off, col in jobs:
                engine.gather_rows(t, B, nn_, off, hcat, col)
        return hcat

    def _readout_backward(self, plan, x, h, gout, g_ext, dx):
        L, H, nn_ = self.num_layers, self.hidden_dim, self.num_nodes
        for i in range(L):
            g_ext[0][i][nn_ - 1::nn_, :H] = gout[:, i * H:(i + 1) * H]
            if self.bidirectional:
                g_ext[1][i][0::nn_, :H] = gout[:, (L + i) * H:(L + i + 1) * H]

    def _train_params(self):
        flat = []
        for d in self.dirs:
            for i in range(self.num_layers):
                c = getattr(self, "cells_%d" % d)[i]
                a = getattr(self, "node_aggr_%d" % d)[i]
                flat += [c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh, a.attn_lin.weight, a.attn_lin.bias, None, None]
        return flat

    def _training_pass(self) -> bool:
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return False
        if self.schedule != "lockstep":
            raise NotImplementedError("the HIP backward pass needs the lock-step schedule")
        return True

    def forward(self, G):
        """`dvae/dagnn.py:99-175` / `dvae/dagnn_bn.py:98-168`."""
        out = self._forward(G)
        if not self.training and engine.PARAM_GUARD:
            # evaluation passes: the parameters behind the derived-weight caches still are what the caches were built from
            # (core.ParamGuard; one small launch, reported like a device-side failure)
            from .core import guard_params
            p0 = next(self.parameters())
            if p0.is_cuda:
                guard_params(self, self._arena_for(p0).err)
        return out

    def invalidate_caches(self) -> None:
        """Drop every tensor derived from the parameters (what `train()` / `eval()` do): call it after updating parameters in
        evaluation mode through a path the version counters do not see (`.data`, a fused optimizer)."""
        for c in self.__dict__.get("_derived", {}).values():
            c.invalidate()
        view = self.__dict__.get("_agg_view_obj")
        if view is not None:
            vc = view.__dict__.get("_variant_cache")
            if vc is not None:
                vc.invalidate()
        from .core import drop_guard
        drop_guard(self)

    def _forward(self, G):
        if self.output_all and self.out_pool not in (K.P_MAX, K.P_MEAN, K.P_ADD):
            raise NotImplementedError("out_pool=%r over all nodes: the reference's own self-attention pooling of the "
                                      "D-VAE models references an undefined layer (dvae/dagnn.py:85-88)" % self.out_pool)
        train = self._training_pass()
        device = self.get_device()
        G = G.to(device)
        x = G.x.float().contiguous()
        N = x.shape[0]
        L, H, nn_ = self.num_layers, self.hidden_dim, self.num_nodes
        if N % nn_ != 0:
            raise ValueError("every graph must have exactly num_nodes=%d nodes (dvae/dagnn.py:150-158)" % nn_)
        B = N // nn_
        bl = G.bi_layer_index
        if not train and not self._agg_plain and not self.output_all and ENCODE_FUSED and self.schedule == "lockstep" \
                and engine.TIMER is None:
            out = self._encode_fused(G, x, B)
            if out is not None:
                return out
        plan = engine.build_plan(G.edge_index, bl[0][0], bl[1][0], G.batch, B, None)
        if self._agg_plain:
            return self._forward_plain_agg(G, plan, x, B, train)
        if self.output_all:   # pool over ALL nodes (dvae/dagnn.py:163-172): states, per-node projection, torch pooling
            if train:
                from .autograd import Recurrence
                flat = Recurrence.apply(self, plan, B, False, x, *self._train_params())
            else:
                hh = run_stack(plan, x, self._cells(), self.dirs, L, H, vid_nodes=self._vid_nodes,
                               schedule=self.schedule, arena=self._arena_for(x))
                flat = [hh[d][i] for d in self.dirs for i in range(L)]
            G.h = torch.cat(([x] if self.out_wx else []) + list(flat), dim=-1)
            if self.bidirectional:
                G.h = self.hg_unify(G.h)
            elif L > 1:
                G.h = self.out_linear(G.h)
            if not (torch.is_grad_enabled() and G.h.requires_grad):   # HIP pooling over the nodes of every graph
                out = torch.empty(B, G.h.shape[1], dtype=torch.float32, device=G.h.device)
                engine.readout_pool(plan, G.h, 2, self.out_pool, out, 0)
                return out
            idx = G.batch.view(-1, 1).expand_as(G.h)
            out = G.h.new_zeros(B, G.h.shape[1])
            if self.out_pool == K.P_MAX:
                return out.scatter_reduce(0, idx, G.h, "amax", include_self=False)
            out = out.scatter_add(0, idx, G.h)
            return out / nn_ if self.out_pool == K.P_MEAN else out
        if train:
            from .autograd import Recurrence
            hcat = Recurrence.apply(self, plan, B, True, x, *self._train_params())[0]
        else:
            h = run_stack(plan, x, self._cells(), self.dirs, L, H, vid_nodes=self._vid_nodes,
                          schedule=self.schedule, arena=self._arena_for(x))
            hcat = self._readout(plan, B, x, h)
        G.h = hcat
        G.batch = G.batch[0::nn_] if self.bidirectional else G.batch[nn_ - 1::nn_]
        lin = self.hg_unify if self.bidirectional else (self.out_linear if L > 1 else None)
        if lin is None:
            return G.h
        if isinstance(lin, nn.Sequential) and len(lin) == 1:   # (`hg_unify` is a Sequential of one Linear: dvae/dagnn.py:66-68)
            lin = lin[0]
        if train or not isinstance(lin, nn.Linear) or G.h.shape[0] * lin.weight.numel() > OWN_LINEAR_MAX:
            return lin(G.h)
        # evaluation, small product (cfg 1: 64 x 256 x 128): on the path's own GEMM - nn.Linear costs the host ~27 us of
        # library dispatch and cfg 1 is host-bound (scripts/small_host_profile.py: 155 -> 134 us per forward); cfg 4's
        # 128 x 1024 x 256 stays with the library (its split-K kernel is the faster one there: 196 vs 285 us)
        return engine.gemm_nt_bias([G.h], [lin.weight.detach()], [None if lin.bias is None else lin.bias.detach()])[0]

    # ------------------------------------------------------------------ agg in {gated_sum, add, max} (dvae/dagnn.py:60-70)
    def _agg_view(self):
        v = self.__dict__.get("_agg_view_obj")
        if v is None:
            v = self.__dict__["_agg_view_obj"] = _AggView(self)
        return v

    def _gated_hip_ok(self) -> bool:
        """`gated_sum` through csrc/variants.hip: the states must be as wide as the gate / mapper inputs say (hs + num_nodes)."""
        w = self.gate_forward[0][0].weight
        return w.shape[1] == self.hidden_dim + self.num_nodes and w.shape[0] == self.hidden_dim

    def _gated_sum_states(self, G, x):
        """`gated_sum` on the NA encoder: the messages are gate(hs_j) * mapper(hs_j) with hs_j = [state ; one-hot vertex id]
        (dvae/dagnn.py:124-137, 269-299), i.e. per NODE P_j = W_g[:, :H] h_j + W_g[:, H + j mod n] + b_g (likewise the
        mapper) - the vertex-id columns are a per-node bias.  Layer by layer on device-side torch ops (a D-VAE batch has
        as many layers as a graph has vertices: <= ~10 steps; differentiable as it stands)."""
        N, H, L, nn_ = x.shape[0], self.hidden_dim, self.num_layers, self.num_nodes
        dev = x.device
        vid = torch.arange(N, device=dev) % nn_
        ei = G.edge_index
        h = [[None] * L for _ in range(2)]
        for d in self.dirs:
            layer_of = G.bi_layer_index[d][0]
            T = int(layer_of.max()) + 1 if N else 0
            feed, other = ei[1 - d], ei[d]          # an edge feeds node `feed` from node `other`
            order = torch.argsort(layer_of[feed] * N + feed, stable=True)
            counts = torch.bincount(layer_of[feed], minlength=T).cumsum(0).cpu().tolist()
            gates = self.gate_forward if d == 0 else self.gate_backward
            maps = self.mapper_forward if d == 0 else self.mapper_backward
            cells = getattr(self, "cells_%d" % d)
            hs = [x.new_zeros(N, H) for _ in range(L)]
            ids = torch.arange(N, device=dev)
            for t in range(T):
                rows = ids[layer_of == t]
                inp = x[rows]
                if t > 0:
                    eids = order[(counts[t - 1]):(counts[t])]
                    src, dst = other[eids], feed[eids]
                for i in range(L):
                    ps = None
                    if t > 0:
                        wg, bg, wm = gates[i][0].weight, gates[i][0].bias, maps[i][0].weight
                        hj = hs[i][src]
                        g = torch.sigmoid(hj @ wg[:, :H].t() + wg[:, H + vid[src]].t() + bg)
                        m = hj @ wm[:, :H].t() + wm[:, H + vid[src]].t()
                        ps = x.new_zeros(N, H).index_add_(0, dst, g * m)[rows]
                    inp = cells[i](inp, ps)
                    hs[i] = hs[i].index_add(0, rows, inp)   # `G.h[d][i][layer] += inp` (dvae/dagnn.py:145)
            h[d] = hs
        return h

    def _forward_plain_agg(self, G, plan, x, B, train):
        """`forward` for agg in {gated_sum, add, max}: `add` / `max` through the generic HIP kernels of csrc/variants.hip
        (`dagnn_variant_run`; training: the reverse sweep of csrc/variants_bwd.hip where it applies, `variants.hip_backward_
        supported`), `gated_sum` (NA) on torch ops; then the read-outs of `dvae/dagnn.py:147-172`."""
        from . import variants
        L, H, nn_ = self.num_layers, self.hidden_dim, self.num_nodes
        if self.agg == K.NA_GATED_SUM and (train or not self._gated_hip_ok()):
            if train:   # differentiable device-side torch ops (the reverse sweep of csrc/variants_bwd.hip has no vertex-id columns)
                h = self._gated_sum_states(G, x)
            else:
                with torch.no_grad():
                    h = self._gated_sum_states(G, x)
        elif self.agg == K.NA_GATED_SUM:   # evaluation: the generic HIP kernels, the one-hot columns as a per-vertex-id bias
            h = variants.run_hip(self._agg_view(), G, x, plan)
        else:
            view = self._agg_view()
            if train:
                if variants.hip_backward_supported(view, G):
                    flat_params = [p for d in self.dirs for i in range(L) for _, p in variants._cell_params(view, d, i)]
                    flat = variants.VariantRecurrence.apply(view, G, plan, x, *flat_params)
                    h = [[None] * L for _ in range(2)]
                    for q, d in enumerate(self.dirs):
                        for i in range(L):
                            h[d][i] = flat[q * L + i]
                else:
                    variants.warn_torch_path(view, G)
                    h = variants.run(view, G, x)
            else:
                h = variants.run_hip(view, G, x, plan)
        N = x.shape[0]
        if self.output_all:
            G.h = torch.cat(([x] if self.out_wx else []) + [h[d][i] for d in self.dirs for i in range(L)], dim=-1)
            if self.bidirectional:
                G.h = self.hg_unify(G.h)
            elif L > 1:
                G.h = self.out_linear(G.h)
            idx = G.batch.view(-1, 1).expand_as(G.h)
            out = G.h.new_zeros(B, G.h.shape[1])
            if self.out_pool == K.P_MAX:
                return out.scatter_reduce(0, idx, G.h, "amax", include_self=False)
            out = out.scatter_add(0, idx, G.h)
            return out / nn_ if self.out_pool == K.P_MEAN else out
        first = torch.arange(0, N, nn_, device=x.device)
        last = first + (nn_ - 1)
        parts = [h[0][i][last] for i in range(L)]
        if self.bidirectional:
            parts += [h[1][i][first] for i in range(L)]
        G.h = torch.cat(parts, dim=-1)
        G.batch = G.batch[first] if self.bidirectional else G.batch[last]
        if self.bidirectional:
            return self.hg_unify(G.h)
        return self.out_linear(G.h) if L > 1 else G.h

    def _encode_fused(self, G, x, B):
        """The evaluation pass up to (and including, when it is small) the final Linear as ONE call into the library
        (`dagnn_encode_forward`, csrc/encode.hip): the same launches as the step-by-step path below, without the Python
        between them - these batches (64 x 8 / 128 x 10 nodes) are host-bound.  None: the shape is not one the dataflow
        kernel serves (the caller takes the general path)."""
        import ctypes as C
        from . import _lib
        from .core import pack_dataflow
        L, H, nn_, dirs = self.num_layers, self.hidden_dim, self.num_nodes, self.dirs
        cells = self._cells()
        Hp = cells[(dirs[0], 0)].Hp
        N, dev = x.shape[0], x.device
        if N == 0 or not engine.dataflow_width(Hp) or Hp > 256 or N * 3 * Hp >= (1 << 31):
            return None
        groups = engine.dataflow_groups(dev, len(dirs), L, Hp, B)
        if groups <= 0:
            return None
        lib = _lib.load()
        arena = self._arena_for(x)
        arena.poll()   # a failure an earlier pass reported (no synchronisation)
        pack_dataflow(cells.values())
        ei = engine._dev(G.edge_index, "edge_index", torch.int64)
        bl = G.bi_layer_index
        lf, lb = engine._dev(bl[0][0], "layer ids", torch.int64), engine._dev(bl[1][0], "layer ids", torch.int64)
        batch = engine._dev(G.batch, "batch", torch.int64)
        plan = engine.PlanHandle(N, ei.shape[1], B, 0, dev)
        a = _lib.EncodeArgs()
        a.plan = plan.desc
        a.edge_index, a.layer_fwd, a.layer_bwd, a.batch = ei.data_ptr(), lf.data_ptr(), lb.data_ptr(), batch.data_ptr()
        a.plan_status = plan.status.data_ptr()
        f32 = dict(dtype=torch.float32, device=dev)
        gi = [None, None]
        for q, d in enumerate(dirs):
            c = cells[(d, 0)]
            gi[d] = torch.empty(N, 3 * Hp, **f32)
            a.gemm[q] = _lib.GemmGroup(x.data_ptr(), c.w_ih.data_ptr(), c.b_ih.data_ptr(), gi[d].data_ptr())
        a.num_gemm, a.gemm_cols, a.in_dim, a.ld_x = len(dirs), 3 * Hp, x.shape[1], x.stride(0)
        sbytes = lib.dagnn_dataflow_bytes(N, B, groups)
        sched = torch.empty((sbytes + 3) // 4, dtype=torch.int32, device=dev)
        plan.__dict__["_df"] = {(int(groups), engine.DF_COST_LAYER, engine.DF_COST_ROW): sched}   # (built inside the call)
        a.schedule, a.schedule_bytes = sched.data_ptr(), sbytes
        a.cost_layer, a.cost_row = engine.DF_COST_LAYER, engine.DF_COST_ROW
        ld = engine.frontier_ld(Hp)
        h = [[torch.empty(N, ld, **f32) if d in dirs else None for _ in range(L)] for d in range(2)]
        engine.dataflow_args(plan, dirs, L, Hp, cells, gi, h, groups, vid_mod=self._vid_nodes, arena=arena, args=a.df)
        hcat = torch.empty(B, len(dirs) * L * H, **f32)
        jobs = [(h[0][i], nn_ - 1, i * H) for i in range(L)]
        if self.bidirectional:
            jobs += [(h[1][i], 0, (L + i) * H) for i in range(L)]
        if len(jobs) > 16:
            return None
        for k, (t, off, col) in enumerate(jobs):
            a.jobs[k] = _lib.GatherJob0(t.data_ptr(), t.stride(0), H, int(off), int(col))
        a.num_jobs, a.stride, a.hcat, a.ld_hcat = len(jobs), nn_, hcat.data_ptr(), hcat.shape[1]
        lin = self.hg_unify if self.bidirectional else (self.out_linear if L > 1 else None)
        if isinstance(lin, nn.Sequential) and len(lin) == 1:
            lin = lin[0]
        out = None
        fused_lin = isinstance(lin, nn.Linear) and hcat.shape[0] * lin.weight.numel() <= OWN_LINEAR_MAX and \
            lin.weight.shape[1] == hcat.shape[1] and hcat.shape[1] % 4 == 0
        if fused_lin:
            w = lin.weight.detach()
            out = torch.empty(B, w.shape[0], **f32)
            a.w_out, a.b_out = w.data_ptr(), (None if lin.bias is None else lin.bias.detach().data_ptr())
            a.out, a.out_dim = out.data_ptr(), w.shape[0]
        with engine.persistent_launch(x):
            engine.check(lib.dagnn_encode_forward(C.byref(a), engine._stream(x)), "dagnn_encode_forward")
        arena.watch(None, folded=True)
        G.h = hcat
        G.batch = G.batch[0::nn_] if self.bidirectional else G.batch[nn_ - 1::nn_]
        if lin is None:
            return hcat
        return out if fused_lin else lin(hcat)

    def encode(self, G):
        """(mu, logvar) of a list of graphs (`dvae/dagnn.py:177-184`)."""
        if type(G) != list:
            G = [G]
        b = GraphBatch.from_data_list(G)
        Hg = self(b)
        return self.fc1(Hg), self.fc2(Hg)


    # ------------------------------------------------------------------ decoder-side single-vertex step
    def _get_zeros(self, n, length):
        return torch.zeros(n, length, device=self.get_device())

    def _get_zero_hidden(self, n=1):
        return self._get_zeros(n, self.hs)

    def _one_hot(self, idx, length):
        """`models_pyg.py:98-107`: a list gives one row per entry (None for an empty list), an int one row."""
        if type(idx) in (list, range):
            if len(idx) == 0:
                return None
            ids = torch.tensor(list(idx), dtype=torch.long).view(-1, 1)
        else:
            ids = torch.tensor([[int(idx)]], dtype=torch.long)
        return torch.zeros(ids.shape[0], length).scatter_(1, ids, 1).to(self.get_device())

    def _ipropagate_to(self, G, v, propagator, H=None, reverse=False):
        """New states at vertex `v` of every graph in `G` that has one, from the states of its predecessors
        (`dvae/dagnn.py:187-239`, `dvae/dagnn_bn.py:179-238`; called by the decoder as `_update_iv`,
        `models_pyg.py:247-250`, with `propagator = self.grud`).  `G` holds igraph-style graphs: `g.vcount()`,
        `g.predecessors(v)`, `g.vs[x]['type']`, `g.vs[x]['H_forward<l>']` ([1, hs] tensors, written for `v`).

        ONE HIP launch per call (`dagnn_iprop_step`, csrc/misc.hip) for the padded soft-max aggregate and the L
        stacked GRU cells of all graphs; there is no CPU path (the model must live on the GPU).  Reproduced as the
        reference computes it, quirks included: the predecessor lists are padded to the longest one with zero rows and
        the attention soft-max runs over the padding as well (a zero key scores `w_q.q + b`; that term is common to all
        slots and cancels, so padded slots score 0 and take weight away from the real predecessors - which is why this
        is NOT the encoder's aggregate); and the aggregate is computed from the layer-0 states only and reused by every
        layer above (`H` is no longer None in the later iterations of the reference's loop).  The host side only
        gathers the igraph-style inputs into dense tensors and writes the new states back into the vertices."""
        assert not reverse
        if self.agg == K.NA_SELF_ATTN_H:
            raise NotImplementedError("the decoder-side step with agg='self_attn_h': the reference's own SelfAttnConv calls an "
                                      "undefined `attn_linear` on this path (dvae/dagnn.py:317-321)")
        G = [g for g in G if g.vcount() > v]
        if len(G) == 0:
            return None
        dev = self.get_device()
        if H is not None:
            H = H[list(range(len(G)))].to(dev)   # the reference indexes with the positions of the already filtered list
        X = self._one_hot([g.vs[v]["type"] for g in G], self.nvt)
        values = pred_vid = None
        lin = self.node_aggr_0[0].attn_lin   # AttnConv.forward with edge_index=None (`dagnn.py:391-399`), stacked layer 0
        dq = self._key_offset(0)
        w = lin.weight[0]   # (not detached: the training caller `loss()` -> `_update_iv` backpropagates through this step)
        if H is None:
            preds = [g.predecessors(v) for g in G]
            P = max(len(p) for p in preds)
            if P > 0:
                rows, ids = [], []
                zero = self._get_zeros(1, self.hs)
                for g, p in zip(G, preds):
                    rows += [g.vs[x]["H_forward0"].to(dev) for x in p] + [zero] * (P - len(p))
                    ids += list(p) + [-1] * (P - len(p))
                values = torch.cat(rows, 0).view(len(G), P, self.hs)
                pred_vid = torch.tensor(ids, dtype=torch.int32).view(len(G), P).to(dev)
        cells = list(propagator)[:self.num_layers]
        w_key = w[dq:dq + self.hs]
        vid_bias = w[dq + self.hs:dq + self.hs + self.max_n] if self._use_vids else None
        flat = [t for c in cells for t in (c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh)]
        diff = [t for t in [values, H, w_key, vid_bias] + flat if t is not None]
        if torch.is_grad_enabled() and any(t.requires_grad for t in diff):
            # training (`models_pyg.py:398-442`: the reconstruction loss reaches `grud`, `attn_lin` and `H0 = tanh(fc3(z))`
            # through these states): the same HIP launch forward, its reverse pass behind an autograd.Function
            states = _IpropStep.apply(values, pred_vid, w_key, vid_bias, H, X, len(cells), *flat)
        else:
            states = engine.iprop_step(values, pred_vid, w_key, vid_bias, H, X, cells)
        for l in range(self.num_layers):
            for i, g in enumerate(G):
                g.vs[v]["H_forward%d" % l] = states[l, i:i + 1]
        return states[self.num_layers - 1]


class DAGNN_NA(_DvaeDagnn):
    """The reference's `dvae/dagnn.py::DAGNN` (keys carry a one-hot vertex id, `:130-134`)."""
    _use_vids = True

    def __init__(self, emb_dim, hidden_dim, out_dim, max_n, nvt, START_TYPE, END_TYPE, hs, nz,
                 num_layers=2, bidirectional=False, agg=K.NA_ATTN_H, out_wx=False, out_pool_all=False,
                 out_pool=K.P_MAX, dropout=0.0, num_nodes=8):
        super().__init__(max_n, nvt, START_TYPE, END_TYPE, hs, nz, bidirectional=bidirectional, num_layers=num_layers)
        self._setup(emb_dim, hidden_dim, out_dim, num_layers, bidirectional, agg, out_wx, out_pool_all, out_pool,
                    dropout, num_nodes)


class DAGNN_BN(_DvaeDagnn):
    """The reference's `dvae/dagnn_bn.py::DAGNN_BN` on `DVAE_BN_PYG` (`models_pyg.py:539-560`)."""
    _use_vids = False

    def __init__(self, emb_dim, hidden_dim, out_dim, max_n, nvt, START_TYPE, END_TYPE, hs, nz, num_layers=2,
                 bidirectional=True, agg=K.NA_ATTN_H, out_wx=False, out_pool_all=False, out_pool=K.P_MAX,
                 dropout=0.0, num_nodes=8):
        super().__init__(max_n, nvt, START_TYPE, END_TYPE, hs, nz, bidirectional=bidirectional, vid=False,
                         num_layers=num_layers)
        # DVAE_BN_PYG (aggx=0) re-creates these with the first layer reading node types
        lin = lambda l, bias: nn.Linear(self.nvt if l == 0 else hs, hs, bias=bias)  # noqa: E731
        self.mapper_forward = nn.ModuleList([nn.Sequential(lin(l, False)) for l in range(num_layers)])
        self.mapper_backward = nn.ModuleList([nn.Sequential(lin(l, False)) for l in range(num_layers)])
        self.gate_forward = nn.ModuleList([nn.Sequential(lin(l, True), nn.Sigmoid()) for l in range(num_layers)])
        self.gate_backward = nn.ModuleList([nn.Sequential(lin(l, True), nn.Sigmoid()) for l in range(num_layers)])
        self.add_edge = nn.Sequential(nn.Linear(hs * 3, hs), nn.ReLU(), nn.Linear(hs, 1))
        self._setup(emb_dim, hidden_dim, out_dim, num_layers, bidirectional, agg, out_wx, out_pool_all, out_pool,
                    dropout, num_nodes)
