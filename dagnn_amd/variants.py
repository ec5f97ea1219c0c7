"""The aggregator / cell variants no BASELINE configuration exercises.

`agg` in {`mattn_h`, `gated_sum`, `add`, `max`}, `agg_x=True`, `recurr=0` of `ogbg-code/model/dagnn.py`: the module
keeps the reference's constructor and `state_dict` for them, and `forward` lands here instead of in the tuned HIP
recurrence.  Two paths, same values:

* `run_hip` (evaluation / inference, i.e. whenever no gradient is requested): one lock-step pass of the generic HIP
  kernels of `csrc/variants.hip` (`dagnn_variant_run`, `dagnn_variant_aggregate`): an aggregate launch, a cell launch
  and - for `mattn_h` / `gated_sum` - a launch of per-node projections per step.  This module only derives the
  kernel-ready weights (transposes, the `[dim, R]` products that fold the edge encoder into the projections; cached
  per parameter version) and marshals pointers.
* `run` (training): the reference's loop nest (`dagnn.py:144-182`) on differentiable torch-ROCm ops, with its
  O(F*E) per-node edge scan replaced by one stable sort of the edges by the layer of the node they feed.

Neither runs anything on the CPU.

Conv semantics restated from `dagnn.py:232-313,347-409` and PyG-1.6 `propagate` (messages flow j -> i; the result of a
conv is a full [N, .] tensor that is zero where no edge lands, of which the caller reads the frontier rows):

* `AggConv` (`add` | `max`): message h_j + e.  The reference builds ONE module for both directions
  (`dagnn.py:74-75`), so its flow is always source -> target: in the reverse direction the messages land on the
  successors, not on the frontier, and the frontier rows read zeros.  Reproduced as is.
* `GatedSumConv`: message sigmoid(W_g (h_j + e) + b_g) * (W_m (h_j + e) [+ b_m]), summed.
* `MultAttnConv`: logit = sum((W_l q_i + b_l) * (W_r (k_j + e) + b_r)), segment softmax, sum alpha h_j.
* `AttnConv` / `SelfAttnConv` (only reached through `agg_x` / `recurr=0`): logit = attn_lin([q_i ; k_j + e]) /
  attn_lin(k_j + e).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib, engine
from .core import DerivedCache


def _segment_softmax(logit: torch.Tensor, seg: torch.Tensor, num_seg: int) -> torch.Tensor:
    """PyG-1.6 `softmax`: exp(x - segment max) / (segment sum + 1e-16)."""
    mx = logit.new_full((num_seg,), float("-inf")).scatter_reduce_(0, seg, logit, "amax", include_self=True)
    ex = (logit - mx[seg]).exp()
    sm = logit.new_zeros(num_seg).scatter_add_(0, seg, ex)
    return ex / (sm[seg] + 1e-16)


def _conv(agg: str, p, vals: torch.Tensor, keys: Optional[torch.Tensor], query: Optional[torch.Tensor],
          src: torch.Tensor, seg: torch.Tensor, rows: torch.Tensor, edge_attr: Optional[torch.Tensor],
          lands_on_frontier: bool) -> torch.Tensor:
    """Frontier rows of one conv: `src` = node j of every edge, `seg` = frontier index of its node i, `rows` = frontier
    node ids.  `p` is the parameter holder of the conv (model._*Params)."""
    F_, dim = rows.shape[0], vals.shape[1]
    out = vals.new_zeros(F_, dim)
    if not lands_on_frontier or src.numel() == 0:
        return out
    e = p.edge_encoder(edge_attr) if getattr(p, "wea", False) else None
    hj = vals[src]
    if agg in ("add", "max"):
        msg = hj + e if e is not None else hj
        if agg == "add":
            return out.index_add_(0, seg, msg)
        return out.scatter_reduce_(0, seg.view(-1, 1).expand_as(msg), msg, "amax", include_self=False)
    if agg == "gated_sum":
        m = hj + e if e is not None else hj
        return out.index_add_(0, seg, p.gate(m) * p.mapper(m))
    k = (keys if keys is not None else vals)[src]
    if e is not None:
        k = k + e
    if "mattn" in agg:
        logit = (p.attn_linl(query[rows])[seg] * p.attn_linr(k)).sum(dim=1)
    elif "self_attn" in agg:
        logit = p.attn_lin(k).squeeze(-1)
    else:
        logit = p.attn_lin(torch.cat([query[rows][seg], k], dim=-1)).squeeze(-1)
    alpha = _segment_softmax(logit, seg, F_)
    return out.index_add_(0, seg, hj * alpha.unsqueeze(-1))


_TORCH_PATH_SEEN = set()


def warn_torch_path(mod, G) -> None:
    """Say ONCE per model shape that a training step of this constructor string runs the reference's loop nest on torch
    ops (`run` below) instead of the HIP reverse sweep (`VariantRecurrence`): 10-100x slower per step, otherwise silent.
    The HIP sweep covers up to 8 cells, widths that are multiples of 4 and at most two edge features behind an edge
    encoder (`hip_backward_supported`)."""
    key = (mod.agg, bool(mod.agg_x), bool(mod.recurr), mod.hidden_dim, mod.emb_dim, mod.num_layers, len(mod.dirs))
    if key in _TORCH_PATH_SEEN:
        return
    _TORCH_PATH_SEEN.add(key)
    import warnings
    warnings.warn("dagnn_amd: training agg=%r (agg_x=%s, recurr=%s, hidden %d, %d stacked layers, %d direction(s)) runs on "
                  "torch ops, layer by layer - the HIP reverse sweep of the constructor-string variants takes at most 8 "
                  "cells, widths that are multiples of 4 and at most two edge features; expect 10-100x the step time"
                  % (mod.agg, mod.agg_x, mod.recurr, mod.hidden_dim, mod.num_layers, len(mod.dirs)),
                  RuntimeWarning, stacklevel=3)


def run(mod, G, x: torch.Tensor) -> List[List[Optional[torch.Tensor]]]:
    """h[d][i] ([N, hidden], d over both direction slots, None for an unused direction) for the variant configured on
    `mod` (a dagnn_amd.DAGNN).  Differentiable: plain torch ops throughout."""
    N, H, L = x.shape[0], mod.hidden_dim, mod.num_layers
    dev = x.device
    ids = torch.arange(N, device=dev)
    ei = G.edge_index
    edge_attr = G.edge_attr if getattr(G, "edge_attr", None) is not None else None
    h: List[List[Optional[torch.Tensor]]] = [[None] * L for _ in range(2)]
    # one AggConv for both directions (ogbg-code/model/dagnn.py:74-75): flow is always source -> target.  The D-VAE models
    # build one per direction (dvae/dagnn.py:66-70: `reverse=True` for the second), so there the messages land on the frontier
    shared_flow = mod.agg in ("add", "max") and getattr(mod, "shared_agg_flow", True)
    for d in mod.dirs:
        aggr = getattr(mod, "node_aggr_%d" % d)
        cells = getattr(mod, "cells_%d" % d)
        if mod.agg_x:
            # the aggregator reads the node inputs only (dagnn.py:159-169): nothing couples the layers, so the whole
            # direction is ONE conv over all edges and one cell call per stacked layer over all nodes (a node of
            # layer 0 has no in-edge: its aggregate row is zero, which is GRUCell(x, None) / the zero block of the
            # Linear cell)
            lands = not (shared_flow and d == 1)
            ps_x = _conv(mod.agg, aggr[0], x, x if mod.agg_attn else None, x if mod.agg_attn else None, ei[d],
                         ei[1 - d], ids, edge_attr, lands)
            if ps_x.shape[1] < H:
                ps_x = torch.cat([ps_x, ps_x.new_zeros(N, H - ps_x.shape[1])], dim=-1)
            inp, hs = x, []
            for i in range(L):
                inp = cells[i](inp, ps_x) if mod.recurr else cells[i](torch.cat([inp, ps_x], dim=1))
                hs.append(inp)
            h[d] = hs
            continue
        layer_of = G.bi_layer_index[d][0]
        T = int(layer_of.max()) + 1 if N else 0
        hs = [x.new_zeros(N, H) for _ in range(L)]
        # edges grouped by the layer of the node they feed (target for d = 0, source for d = 1), inside a layer by node
        # id, inside a node in original order - the order the reference's per-node scan concatenates them in
        feed, other = ei[1 - d], ei[d]
        order = torch.argsort(layer_of[feed] * N + feed, stable=True)
        e_ptr = torch.zeros(T + 1, dtype=torch.long)
        e_ptr[1:] = torch.bincount(layer_of[feed], minlength=T).cumsum(0).cpu()
        local = torch.empty(N, dtype=torch.long, device=dev)
        for t in range(T):
            rows = ids[layer_of == t]
            inp = x[rows]
            if t > 0:
                eids = order[int(e_ptr[t]):int(e_ptr[t + 1])]
                src = other[eids]
                local[rows] = torch.arange(rows.shape[0], device=dev)
                seg = local[feed[eids]]
                ea = edge_attr[eids] if edge_attr is not None else None
                lands = not (shared_flow and d == 1)
            for i in range(L):
                if t == 0:
                    ps = None if mod.recurr else x.new_zeros(rows.shape[0], H)
                else:
                    keys = query = None
                    if mod.agg_attn:
                        keys = x if mod.agg_attn_x else hs[i]
                        query = x if mod.agg_attn_x else (hs[i - 1] if i > 0 else x)
                    ps = _conv(mod.agg, aggr[i], hs[i], keys, query, src, seg, rows, ea, lands)
                inp = cells[i](inp, ps) if mod.recurr else cells[i](torch.cat([inp, ps], dim=1))
                hs[i].index_add_(0, rows, inp)   # `G.h[d][i][layer] += inp` (dagnn.py:182)
        h[d] = hs
    return h


# ---------------------------------------------------------------------------------------------- HIP path
def _derive(mod):
    """Kernel-ready weights of every cell: k-major cell weights, the aggregator's projections and the edge-encoder
    products.  Cached on the module per parameter version."""
    H, L, E = mod.hidden_dim, mod.num_layers, mod.emb_dim

    def make():
        out = {}
        for d in mod.dirs:
            for i in range(L):
                c = getattr(mod, "cells_%d" % d)[i]
                a = getattr(mod, "node_aggr_%d" % d)[0 if mod.agg_x else i]
                in_dim = E if i == 0 else H
                p = {}
                if mod.recurr:
                    p["w_in_t"], p["w_agg_t"] = c.weight_ih.t().contiguous(), c.weight_hh.t().contiguous()
                    p["b_in"], p["b_agg"] = c.bias_ih.contiguous(), c.bias_hh.contiguous()
                else:
                    p["w_in_t"] = c.weight[:, :in_dim].t().contiguous()
                    p["w_agg_t"] = c.weight[:, in_dim:].t().contiguous()
                    p["b_in"], p["b_agg"] = c.bias.contiguous(), None
                We = a.edge_encoder.weight if getattr(a, "wea", False) else None   # [dim, R]
                be = a.edge_encoder.bias if We is not None else None
                if mod.agg in ("add", "max"):
                    p["edge_mat0"] = We.contiguous() if We is not None else None
                    p["edge_vec0"] = be.contiguous() if We is not None else None
                elif mod.agg == "gated_sum":
                    Wg, bg, Wm, bm = a.gate[0].weight, a.gate[0].bias, a.mapper.weight, a.mapper.bias
                    nv = int(getattr(mod, "vid_nodes", 0))
                    if nv:   # D-VAE NA: the mapped vector is [state ; one-hot(vertex id)] - the one-hot columns are a per-id bias
                        p["pq_vid"] = torch.cat([Wg[:, H:H + nv], Wm[:, H:H + nv]], 0).t().contiguous()   # [nv, 2H]
                        Wg, Wm = Wg[:, :H], Wm[:, :H]
                    p["pq_w_t"] = torch.cat([Wg, Wm], 0).t().contiguous()
                    p["pq_b"] = torch.cat([bg, bm if bm is not None else torch.zeros_like(bg)])
                    if We is not None:
                        p["edge_mat0"], p["edge_vec0"] = (Wg @ We).contiguous(), (Wg @ be).contiguous()
                        p["edge_mat1"], p["edge_vec1"] = (Wm @ We).contiguous(), (Wm @ be).contiguous()
                elif "mattn" in mod.agg:
                    Wl, Wr = a.attn_linl.weight, a.attn_linr.weight
                    p["ql_w"], p["ql_b"], p["kr_w"], p["kr_b"] = Wl, a.attn_linl.bias, Wr, a.attn_linr.bias
                    p["ql_w_t"], p["kr_w_t"] = Wl.t().contiguous(), Wr.t().contiguous()
                    if We is not None:
                        p["edge_mat0"], p["edge_vec0"] = (Wr @ We).contiguous(), (Wr @ be).contiguous()
                else:   # additive attention: only the key half of attn_lin matters inside a softmax segment
                    off = 0 if "self_attn" in mod.agg else (E if (i == 0 or mod.agg_x) else
                                                            (E if mod.agg_attn_x else H))
                    kd = E if (mod.agg_x or mod.agg_attn_x) else H
                    wk = a.attn_lin.weight[0, off:off + kd].contiguous()
                    p["edge_vec0"] = wk
                    p["edge_mat0"] = (We.t() @ wk).contiguous() if We is not None else None   # [R]
                out[(d, i)] = p
        return out

    srcs = [q for q in mod.parameters()]
    cache = mod.__dict__.setdefault("_variant_cache", DerivedCache())
    return cache.get(srcs, make, fresh=mod.training)


_MODES = {"add": _lib.AGG_ADD, "max": _lib.AGG_MAX, "gated_sum": _lib.AGG_GATED}


def _ptr(t):
    return None if t is None else t.data_ptr()


def run_plain_dataflow(mod, x: torch.Tensor, plan) -> Optional[List[List[Optional[torch.Tensor]]]]:
    """`agg` = `add` / `max` with GRU cells on the persistent dataflow kernel (csrc/dataflow.hip: the generic loader folds the
    messages h_j + edge_encoder(edge_attr_j) by sum or maximum instead of the attention soft-max; the GRU side is the
    kernel's own).  The reference builds ONE AggConv for both directions (dagnn.py:74-75): in the reverse direction its
    messages land on the successors, the frontier rows read zeros - those cells run with an empty aggregate and poll nothing.
    Returns None where the kernel does not apply (the caller keeps the per-layer variant launches)."""
    from .core import derive_cell, pack_dataflow
    H, L = mod.hidden_dim, mod.num_layers
    dirs = list(mod.dirs)
    N, dev = x.shape[0], x.device
    if not getattr(mod, "_plain_dataflow_ok", False) or not engine.VARIANT_DATAFLOW or not engine.DATAFLOW \
            or mod.agg not in ("add", "max") or not mod.recurr or mod.agg_x \
            or N == 0 or plan.R > 2 or mod.schedule != "lockstep":
        return None
    conv = mod.node_aggr_0[0]
    wea = bool(getattr(conv, "wea", False))
    if wea and plan.R == 0:
        return None
    shared_flow = getattr(mod, "shared_agg_flow", True)
    srcs = [p for d in dirs for cell in getattr(mod, "cells_%d" % d) for p in cell.parameters()] + \
        ([conv.edge_encoder.weight, conv.edge_encoder.bias] if wea else [])

    def make():
        out = {}
        zkey = None
        for d in dirs:
            for i, cell in enumerate(getattr(mod, "cells_%d" % d)):
                if zkey is None:
                    zkey = torch.zeros(1, H, dtype=torch.float32, device=dev)   # (no keys: derive_cell's attention slots stay unused)
                c = derive_cell(cell.weight_ih, cell.weight_hh, cell.bias_ih, cell.bias_hh, zkey, H, 0, i > 0, None, 0,
                                schedule="lockstep", pack=False, stacked=L)
                if c.Hp > 256 or not c.df_ok:
                    return None
                c.agg = 3 if (shared_flow and d == 1) else (1 if mod.agg == "add" else 2)
                c.agg_w = c.agg_b = None
                if wea and c.agg != 3:
                    w = torch.zeros(c.Hp, plan.R, dtype=torch.float32, device=dev)
                    w[:H] = conv.edge_encoder.weight.detach().float()
                    b = torch.zeros(c.Hp, dtype=torch.float32, device=dev)
                    b[:H] = conv.edge_encoder.bias.detach().float()
                    c.agg_w, c.agg_b = w, b
                out[(d, i)] = c
        pack_dataflow(out.values())
        return out

    cache = mod.__dict__.setdefault("_plain_df_cache", DerivedCache())
    cells = cache.get(srcs, make, fresh=mod.training)
    if cells is None:
        return None
    Hp = cells[(dirs[0], 0)].Hp
    groups = engine.dataflow_groups(dev, len(dirs), L, Hp, plan.B)
    if groups <= 0 or N * 3 * Hp >= (1 << 31):
        return None
    arena = mod._arena_for(x)
    arena.poll()
    gi0 = engine.gemm_nt_bias([x] * len(dirs), [cells[(d, 0)].w_ih for d in dirs], [cells[(d, 0)].b_ih for d in dirs])
    gi = [None, None]
    for q, d in enumerate(dirs):
        gi[d] = gi0[q]
    ld = engine.frontier_ld(Hp)
    h = [[torch.empty(N, ld, dtype=torch.float32, device=dev) if d in dirs else None for _ in range(L)] for d in range(2)]
    plan.wait_ready()
    engine.dataflow_run(plan, dirs, L, Hp, cells, gi, h, groups, arena=arena)
    return [[h[d][i][:, :H] if h[d][i] is not None else None for i in range(L)] for d in range(2)]


def run_hip(mod, G, x: torch.Tensor, plan, dataflow: bool = True) -> List[List[Optional[torch.Tensor]]]:
    """h[d][i] ([N, hidden]) through `dagnn_variant_run` (csrc/variants.hip) - or, for `add` / `max` with GRU cells, through the
    persistent dataflow kernel (`run_plain_dataflow`; `dataflow=False`: never - a training pass, whose reverse sweep reads
    dense [N, hidden] state rows).  `plan`: engine.PlanHandle of the batch (with the edge features when the model has an edge
    encoder)."""
    fast = run_plain_dataflow(mod, engine._dev(x.detach(), "node inputs", torch.float32), plan) if dataflow else None
    if fast is not None:
        return fast
    N, H, L, E = x.shape[0], mod.hidden_dim, mod.num_layers, mod.emb_dim
    x = engine._dev(x.detach(), "node inputs", torch.float32)
    dev = x.device
    lib = _lib.load()
    prm = _derive(mod)
    sched = plan.read_schedule()
    stream = engine._stream(x)
    mode = _MODES.get(mod.agg, _lib.AGG_MATTN if "mattn" in mod.agg else _lib.AGG_ATTN)
    shared_flow = mod.agg in ("add", "max") and getattr(mod, "shared_agg_flow", True)
    keep = []   # tensors the launches read: alive until the call returns (stream-ordered afterwards)
    h: List[List[Optional[torch.Tensor]]] = [[None] * L for _ in range(2)]
    args = _lib.VariantArgs()
    args.num_stacked, args.H, args.dir_mask = L, H, sum(1 << d for d in mod.dirs)
    with torch.no_grad():
        for d in mod.dirs:
            lands = 0 if (shared_flow and d == 1) else 1
            for i in range(L):
                h[d][i] = torch.empty(N, H, dtype=torch.float32, device=dev)
            given = None
            if mod.agg_x:
                # the aggregator runs on the inputs (dagnn.py:159-169): all layers in one launch, reused by every cell
                p = prm[(d, 0)]
                given = torch.zeros(N, H, dtype=torch.float32, device=dev)
                keep.append(given)
                T = len(sched[d]) - 1
                if lands and T > 1:
                    a = _lib.VariantAggregator()
                    a.mode, a.lands, a.val_dim, a.out_dim = mode, 1, E, H
                    a.vals, a.ld_vals, a.out, a.ld_out = x.data_ptr(), x.shape[1], given.data_ptr(), H
                    a.edge_mat0, a.edge_vec0 = _ptr(p.get("edge_mat0")), _ptr(p.get("edge_vec0"))
                    a.edge_mat1, a.edge_vec1 = _ptr(p.get("edge_mat1")), _ptr(p.get("edge_vec1"))
                    if mode == _lib.AGG_ATTN:
                        a.node0, a.ld_node, a.aux_dim = x.data_ptr(), x.shape[1], E
                    elif mode == _lib.AGG_MATTN:
                        kr = torch.addmm(p["kr_b"], x, p["kr_w"].t())
                        ql = torch.addmm(p["ql_b"], x, p["ql_w"].t())
                        keep += [kr, ql]
                        a.node0, a.node1, a.ld_node, a.aux_dim = kr.data_ptr(), ql.data_ptr(), kr.shape[1], kr.shape[1]
                    elif mode == _lib.AGG_GATED:
                        pq = torch.addmm(p["pq_b"], x, p["pq_w_t"])
                        keep.append(pq)
                        a.node0, a.node1, a.ld_node = pq.data_ptr(), pq.data_ptr() + 4 * E, 2 * E
                    engine.check(lib.dagnn_variant_aggregate(C.byref(plan.desc), C.byref(a), d, int(sched[d][1]),
                                                             int(sched[d][T]), stream), "dagnn_variant_aggregate")
            for i in range(L):
                p, vc = prm[(d, i)], args.cell[d][i]
                inp = x if i == 0 else h[d][i - 1]
                vc.recurrent, vc.in_dim = (1 if mod.recurr else 0), inp.shape[1]
                vc.input, vc.ld_input = inp.data_ptr(), inp.shape[1]
                vc.w_in_t, vc.w_agg_t = p["w_in_t"].data_ptr(), p["w_agg_t"].data_ptr()
                vc.b_in, vc.b_agg = _ptr(p["b_in"]), _ptr(p["b_agg"])
                vc.h, vc.ld_h = h[d][i].data_ptr(), H
                a = vc.agg
                a.out_dim, a.ld_out = H, H
                if given is not None:
                    a.mode, a.lands, a.out = _lib.AGG_GIVEN, 1, given.data_ptr()
                    continue
                scratch = torch.empty(N, H, dtype=torch.float32, device=dev)
                keep.append(scratch)
                a.mode, a.lands, a.val_dim, a.out = mode, lands, H, scratch.data_ptr()
                a.vals, a.ld_vals = h[d][i].data_ptr(), H
                a.edge_mat0, a.edge_vec0 = _ptr(p.get("edge_mat0")), _ptr(p.get("edge_vec0"))
                a.edge_mat1, a.edge_vec1 = _ptr(p.get("edge_mat1")), _ptr(p.get("edge_vec1"))
                nm = 0
                if mode == _lib.AGG_ATTN:
                    keys = x if mod.agg_attn_x else h[d][i]
                    a.node0, a.ld_node, a.aux_dim = keys.data_ptr(), keys.shape[1], keys.shape[1]
                elif mode == _lib.AGG_GATED:
                    pq = torch.empty(N, 2 * H, dtype=torch.float32, device=dev)
                    keep.append(pq)
                    a.node0, a.node1, a.ld_node = pq.data_ptr(), pq.data_ptr() + 4 * H, 2 * H
                    m = vc.map[nm]
                    m.w_t, m.bias, m.out, m.ld_out, m.out_dim = p["pq_w_t"].data_ptr(), p["pq_b"].data_ptr(), \
                        pq.data_ptr(), 2 * H, 2 * H
                    if p.get("pq_vid") is not None:
                        m.vid_mod, m.vid_bias = int(mod.vid_nodes), p["pq_vid"].data_ptr()
                    nm += 1
                elif mode == _lib.AGG_MATTN:
                    qd = p["kr_w"].shape[0]
                    if mod.agg_attn_x:     # keys are the inputs: one library GEMM
                        kr = torch.addmm(p["kr_b"], x, p["kr_w"].t())
                    else:                  # keys are this cell's states: projected as the rows are produced
                        kr = torch.empty(N, qd, dtype=torch.float32, device=dev)
                        m = vc.map[nm]
                        m.w_t, m.bias, m.out, m.ld_out, m.out_dim = p["kr_w_t"].data_ptr(), p["kr_b"].data_ptr(), \
                            kr.data_ptr(), qd, qd
                        nm += 1
                    if mod.agg_attn_x or i == 0:   # the query is the node input
                        ql = torch.addmm(p["ql_b"], x, p["ql_w"].t())
                    else:                          # the query is the state of the cell below: projected there
                        ql = torch.empty(N, qd, dtype=torch.float32, device=dev)
                        below = args.cell[d][i - 1]
                        m = below.map[below.num_maps]
                        m.w_t, m.bias, m.out, m.ld_out, m.out_dim = p["ql_w_t"].data_ptr(), p["ql_b"].data_ptr(), \
                            ql.data_ptr(), qd, qd
                        below.num_maps += 1
                    keep += [kr, ql]
                    a.node0, a.node1, a.ld_node, a.aux_dim = kr.data_ptr(), ql.data_ptr(), qd, qd
                vc.num_maps += nm
        ptrs = (C.POINTER(C.c_int32) * 2)()
        nl = (C.c_int32 * 2)()
        if mod.agg_x:   # the aggregate is an input: the layers are independent, every cell is one launch over all rows
            import numpy as np
            sched = [np.array([0, N], dtype=np.int32)] * 2
        for d in (0, 1):
            ptrs[d] = sched[d].ctypes.data_as(C.POINTER(C.c_int32))
            nl[d] = len(sched[d]) - 1
        with engine._span("variant_run", x):
            engine.check(lib.dagnn_variant_run(C.byref(plan.desc), C.byref(args), ptrs, nl, stream),
                         "dagnn_variant_run")
    del keep
    return h


# ---------------------------------------------------------------------------------------------- HIP training path
_BWD_MODES = {"gated_sum": _lib.AGG_GATED, "mattn_h": _lib.AGG_MATTN, "add": _lib.AGG_ADD, "max": _lib.AGG_MAX,
              "attn_h": _lib.AGG_ATTN, "attn_x": _lib.AGG_ATTN, "self_attn_h": _lib.AGG_ATTN, "self_attn_x": _lib.AGG_ATTN}


def _attn_slice(mod, i):
    """(offset of the key weights inside attn_lin.weight, key width) of stacked layer i - the rule of `_derive`."""
    E, H = mod.emb_dim, mod.hidden_dim
    off = 0 if "self_attn" in mod.agg else (E if (i == 0 or mod.agg_x) else (E if mod.agg_attn_x else H))
    kd = E if (mod.agg_x or mod.agg_attn_x) else H
    return off, kd


def hip_backward_supported(mod, G) -> bool:
    """Every constructor-string variant trains through HIP (csrc/variants_bwd.hip): `gated_sum`, `mattn_h`, `add`, `max`
    with GRU or Linear (`recurr=0`) cells, the additive-attention aggregators on the Linear cell (with GRU cells they
    are the tuned main path), and all of them with `agg_x`.  Left on the differentiable torch-ops path: more than 8
    cells, widths that are not multiples of 4, more than two edge features with an edge encoder."""
    if mod.agg not in _BWD_MODES or len(mod.dirs) * mod.num_layers > 8:
        return False
    if _BWD_MODES[mod.agg] == _lib.AGG_ATTN and mod.recurr and not mod.agg_x:
        return False   # (never reached: that is the main path)
    if mod.hidden_dim % 4 or mod.emb_dim % 4 or (mod.agg_x and mod.emb_dim > mod.hidden_dim):
        return False
    has_enc = getattr(mod.node_aggr_0[0], "wea", False)
    if has_enc and (getattr(G, "edge_attr", None) is None or G.edge_attr.view(G.edge_attr.shape[0], -1).shape[1] > 2):
        return False
    return True


def _cell_params(mod, d, i):
    """(name, parameter) of everything cell (d, i) reads, in a fixed order.  With `agg_x` only stacked layer 0 owns an
    aggregator (the reference calls `node_aggr[0]` once per direction, dagnn.py:159-169)."""
    c = getattr(mod, "cells_%d" % d)[i]
    if mod.recurr:
        out = [("w_ih", c.weight_ih), ("w_hh", c.weight_hh), ("b_ih", c.bias_ih), ("b_hh", c.bias_hh)]
    else:
        out = [("w_lin", c.weight), ("b_lin", c.bias)]
    if mod.agg_x and i > 0:
        return out
    a = getattr(mod, "node_aggr_%d" % d)[i]
    if mod.agg == "gated_sum":
        out += [("wg", a.gate[0].weight), ("bg", a.gate[0].bias), ("wm", a.mapper.weight)]
        if a.mapper.bias is not None:
            out.append(("bm", a.mapper.bias))
    elif mod.agg == "mattn_h":
        out += [("wl", a.attn_linl.weight), ("bl", a.attn_linl.bias), ("wr", a.attn_linr.weight), ("br", a.attn_linr.bias)]
    elif _BWD_MODES.get(mod.agg) == _lib.AGG_ATTN:
        out += [("attn_w", a.attn_lin.weight), ("attn_b", a.attn_lin.bias)]
    if getattr(a, "wea", False):
        out += [("we", a.edge_encoder.weight), ("be", a.edge_encoder.bias)]
    return out


def _setup_aggregator(mod, plan, bc, o, p, cp, mode, lands, d, i, vals, query, W, a_out, rows, stream, x):
    """Forward quantities of one aggregator for its reverse pass: the per-node projections (library GEMMs), the attention
    weights and the aggregate `a_out` [N, W] of the rows `rows` = (first slot, end slot) (HIP), and the fields of the C
    struct `bc` that describe it.  `vals` [N, W] are the aggregated values (the cell's states, or x with `agg_x`)."""
    lib = _lib.load()
    N, R = vals.shape[0], plan.R
    f32 = dict(dtype=torch.float32, device=vals.device)
    has_enc = "we" in cp
    bc.mode, bc.lands = mode, lands
    bc.edge_mat0, bc.edge_vec0 = _ptr(p.get("edge_mat0")), _ptr(p.get("edge_vec0"))
    bc.edge_mat1, bc.edge_vec1 = _ptr(p.get("edge_mat1")), _ptr(p.get("edge_vec1"))
    bc.h, bc.a = vals.data_ptr(), a_out.data_ptr()
    o["esum"] = None

    def aggregate(agg_mode, node0=None, node1=None, ld_node=0):
        if rows[1] <= rows[0] or not lands:
            return
        ag = _lib.VariantAggregator()
        ag.mode, ag.lands, ag.val_dim, ag.out_dim = agg_mode, 1, W, W
        ag.vals, ag.ld_vals, ag.out, ag.ld_out = vals.data_ptr(), W, a_out.data_ptr(), W
        ag.node0, ag.node1, ag.ld_node = node0, node1, ld_node
        ag.edge_mat0, ag.edge_vec0, ag.edge_mat1, ag.edge_vec1 = bc.edge_mat0, bc.edge_vec0, bc.edge_mat1, bc.edge_vec1
        engine.check(lib.dagnn_variant_aggregate(C.byref(plan.desc), C.byref(ag), d, rows[0], rows[1], stream),
                     "dagnn_variant_aggregate")

    if mode == _lib.AGG_GATED:
        pq = torch.addmm(p["pq_b"], vals, p["pq_w_t"])
        o["node0"], o["dnode0"] = pq, torch.zeros(N, 2 * W, **f32)
        o["w_node"] = torch.cat([cp["wg"], cp["wm"]], 0).contiguous()
        bc.node0, bc.dnode0, bc.w_node, bc.proj_dim = pq.data_ptr(), o["dnode0"].data_ptr(), o["w_node"].data_ptr(), W
        if has_enc and R > 0:
            o["esum"] = torch.zeros(N, 2 * R * W, **f32)
        aggregate(_lib.AGG_GATED, pq.data_ptr(), pq.data_ptr() + 4 * W, 2 * W)
    elif mode in (_lib.AGG_ADD, _lib.AGG_MAX):
        if has_enc and R > 0 and lands:
            o["esum"] = torch.zeros(N, (R + 1) * W, **f32)
        aggregate(mode)
    elif mode == _lib.AGG_ATTN:
        keys = x if (mod.agg_attn_x or mod.agg_x) else vals
        off, kd = _attn_slice(mod, i)
        o.update(node0=keys, dnode0=torch.zeros(N, **f32), alpha=torch.zeros(max(plan.E, 1), **f32))
        if has_enc and R > 0:
            o["esum"] = torch.zeros(N, R, **f32)
        bc.proj_dim = kd
        bc.reserved = 1 if (mod.agg_x or not mod.agg_attn_x) else 0   # the keys are the aggregated values themselves
        bc.node0, bc.dnode0, bc.alpha = keys.data_ptr(), o["dnode0"].data_ptr(), o["alpha"].data_ptr()
        bc.w_node = p["edge_vec0"].data_ptr()
        if rows[1] > rows[0]:
            engine.check(lib.dagnn_variant_mattn_prepare(C.byref(plan.desc), C.byref(bc), d, W, rows[0], rows[1], stream),
                         "dagnn_variant_mattn_prepare")
    else:   # mattn
        P = cp["wl"].shape[0]
        kr = torch.addmm(cp["br"], vals, cp["wr"].t())
        ql = torch.addmm(cp["bl"], query, cp["wl"].t())
        o.update(node0=kr, node1=ql, dnode0=torch.zeros(N, P, **f32), dnode1=torch.zeros(N, P, **f32),
                 alpha=torch.zeros(max(plan.E, 1), **f32), dlogit=torch.zeros(max(plan.E, 1), **f32))
        if has_enc and R > 0:
            o["esum"] = torch.zeros(N, R * P, **f32)
        bc.proj_dim = P
        bc.node0, bc.node1, bc.dnode0, bc.dnode1 = (o[n].data_ptr() for n in ("node0", "node1", "dnode0", "dnode1"))
        bc.alpha, bc.dlogit = o["alpha"].data_ptr(), o["dlogit"].data_ptr()
        bc.w_node, bc.w_query = cp["wr"].data_ptr(), cp["wl"].data_ptr()
        if rows[1] > rows[0]:
            engine.check(lib.dagnn_variant_mattn_prepare(C.byref(plan.desc), C.byref(bc), d, W, rows[0], rows[1], stream),
                         "dagnn_variant_mattn_prepare")
    bc.esum = _ptr(o["esum"])


def _aggregator_grads(mod, mode, o, cp, vals, query, R, W, i, dx_d):
    """{name: gradient} of one aggregator's parameters from the sweep's per-node outputs (transposed products and sums
    over all nodes); `dx_d` additionally receives the key gradient of the `*_x` attention aggregators."""
    g = {}
    es = o.get("esum")
    f32 = dict(dtype=torch.float32, device=vals.device)
    if mode == _lib.AGG_GATED:
        dpq = o["dnode0"]
        prod = dpq.t() @ vals                       # [2W, W]: dP^T h | dM^T h
        sums = dpq.sum(0)
        gwg, gwm, dPs, dMs = prod[:W], prod[W:], sums[:W], sums[W:]
        if "we" in cp:
            We, be = cp["we"], cp["be"]
            gwg = gwg + torch.outer(dPs, be)
            gwm = gwm + torch.outer(dMs, be)
            gbe = cp["wg"].t() @ dPs + cp["wm"].t() @ dMs
            gwe = torch.zeros_like(We)
            if es is not None:
                esm = es.sum(0).view(2, R, W)      # [gate | map][r][k]
                for r in range(R):
                    gwg = gwg + torch.outer(esm[0, r], We[:, r])
                    gwm = gwm + torch.outer(esm[1, r], We[:, r])
                    gwe[:, r] = cp["wg"].t() @ esm[0, r] + cp["wm"].t() @ esm[1, r]
            g["we"], g["be"] = gwe, gbe
        g["wg"], g["bg"], g["wm"] = gwg, dPs, gwm
        if "bm" in cp:
            g["bm"] = dMs
    elif mode == _lib.AGG_MATTN:
        dkr, dql = o["dnode0"], o["dnode1"]
        dkrs = dkr.sum(0)
        gwr = dkr.t() @ vals
        g["wl"], g["bl"] = dql.t() @ query, dql.sum(0)
        if "we" in cp:
            We, be = cp["we"], cp["be"]
            gwr = gwr + torch.outer(dkrs, be)
            gwe = torch.zeros_like(We)
            if es is not None:
                esm = es.sum(0).view(R, -1)
                for r in range(R):
                    gwr = gwr + torch.outer(esm[r], We[:, r])
                    gwe[:, r] = cp["wr"].t() @ esm[r]
            g["we"], g["be"] = gwe, cp["wr"].t() @ dkrs
        g["wr"], g["br"] = gwr, dkrs
    elif mode == _lib.AGG_ATTN:
        # logit_e = w_k . (key_j + W_e attr_e + b_e) (+ query / bias terms that cancel inside a segment: exact zeros);
        # sigma_v = sum of ds over the out-edges of v
        sig = o["dnode0"]
        off, kd = _attn_slice(mod, i)
        wk = cp["attn_w"][0, off:off + kd]
        g_key = (o["node0"] * sig[:, None]).sum(0)
        if mod.agg_attn_x and not mod.agg_x:   # the keys are x, the values are the states: the key gradient goes to x here
            dx_d += sig[:, None] * wk[None, :]
        if "we" in cp:
            m = es.sum(0) if es is not None else torch.zeros(R, **f32)
            ssum = sig.sum()
            g_key = g_key + cp["we"] @ m + cp["be"] * ssum
            g["we"], g["be"] = torch.outer(wk, m), wk * ssum
        g_attn = torch.zeros_like(cp["attn_w"])
        g_attn[0, off:off + kd] = g_key
        g["attn_w"], g["attn_b"] = g_attn, torch.zeros_like(cp["attn_b"])
    else:   # add / max: only the (shared) edge encoder has parameters
        if "we" in cp:
            gwe, gbe = torch.zeros_like(cp["we"]), torch.zeros_like(cp["be"])
            if es is not None:
                esm = es.sum(0).view(R + 1, W)
                for r in range(R):
                    gwe[:, r] = esm[r]
                gbe = esm[R]
            g["we"], g["be"] = gwe, gbe
    return g


class VariantRecurrence(torch.autograd.Function):
    """States h[d][i] of a constructor-string variant, differentiable: forward = `run_hip` (the generic lock-step kernels
    of csrc/variants.hip), backward = the reverse sweep of csrc/variants_bwd.hip + a parallel epilogue (weight gradients as
    transposed products over all nodes).  Inputs after `x`: the parameters of every cell in `_cell_params` order (a
    module shared by several cells - `add` / `max` - simply appears several times)."""

    @staticmethod
    def forward(ctx, mod, G, plan, x, *params):
        h = run_hip(mod, G, x, plan, dataflow=False)
        ctx.mod, ctx.plan, ctx.h = mod, plan, h
        ctx.save_for_backward(x, *params)
        return tuple(h[d][i] for d in mod.dirs for i in range(mod.num_layers))

    @staticmethod
    def backward(ctx, *gouts):
        mod, plan, h = ctx.mod, ctx.plan, ctx.h
        saved = list(ctx.saved_tensors)
        x, params = saved[0].detach(), [p.detach() for p in saved[1:]]
        N, H, L, E = x.shape[0], mod.hidden_dim, mod.num_layers, mod.emb_dim
        f32 = dict(dtype=torch.float32, device=x.device)
        lib = _lib.load()
        mode = _BWD_MODES[mod.agg]
        R = plan.R
        sched = plan.read_schedule()
        stream = engine._stream(x)
        prm = _derive(mod)
        shared_flow = mod.agg in ("add", "max") and getattr(mod, "shared_agg_flow", True)
        recurr, agg_x = bool(mod.recurr), bool(mod.agg_x)
        names, cellp, k = [], {}, 0
        for d in mod.dirs:
            for i in range(L):
                spec = _cell_params(mod, d, i)
                cellp[(d, i)] = {n: params[k + q] for q, (n, _) in enumerate(spec)}
                names += [(d, i, n) for n, _ in spec]
                k += len(spec)
        args = _lib.VariantBwdArgs()
        args.num_stacked, args.H, args.dir_mask = L, H, sum(1 << d for d in mod.dirs)
        g, res, aggx = {}, {}, {}
        q = 0
        for d in mod.dirs:
            for i in range(L):
                go = gouts[q]
                q += 1
                g[(d, i)] = go.detach().float().contiguous().clone() if go is not None else torch.zeros(N, H, **f32)
        dxd = {d: torch.zeros(N, E, **f32) for d in mod.dirs}
        with torch.no_grad():
            for d in mod.dirs:
                lands = 0 if (shared_flow and d == 1) else 1
                T = len(sched[d]) - 1
                rows = (int(sched[d][1]), int(sched[d][T])) if T > 1 else (0, 0)
                given = None
                if agg_x:
                    # the aggregator reads x only (dagnn.py:159-169): its output a_x [N, E] (zero-padded to H) is the
                    # aggregate of EVERY stacked cell of the direction
                    ao = dict(a=torch.zeros(N, E, **f32))
                    bca = _lib.VariantBwdCell()
                    bca.in_dim = E
                    _setup_aggregator(mod, plan, bca, ao, prm[(d, 0)], cellp[(d, 0)], mode, lands, d, 0, x, x, E, ao["a"], rows, stream, x)
                    given = torch.zeros(N, H, **f32)
                    given[:, :E] = ao["a"]
                    aggx[d] = (bca, ao)
                for i in range(L):
                    p, cp, bc = prm[(d, i)], cellp[(d, i)], args.cell[d][i]
                    hi = h[d][i]
                    u = x if i == 0 else h[d][i - 1]
                    in_dim = u.shape[1]
                    o = dict(u=u, dgi=torch.zeros(N, 3 * H, **f32), dgh=torch.zeros(N, 3 * H, **f32), da=torch.zeros(N, H, **f32))
                    bc.in_dim, bc.recurrent = in_dim, int(recurr)
                    if agg_x:
                        o["a"] = given
                        bc.mode, bc.lands = _lib.AGG_GIVEN, 1
                        bc.h, bc.a = hi.data_ptr(), given.data_ptr()
                    else:
                        o["a"] = torch.zeros(N, H, **f32)
                        _setup_aggregator(mod, plan, bc, o, p, cp, mode, lands, d, i, hi, u, H, o["a"], rows, stream, x)
                    a = o["a"]
                    if recurr:
                        o["gi"] = engine.gemm_nt_bias([u], [cp["w_ih"]], [cp["b_ih"]])[0]
                        o["gh"] = engine.gemm_nt_bias([a], [cp["w_hh"]], [cp["b_hh"]])[0]
                        bc.gi, bc.gh = o["gi"].data_ptr(), o["gh"].data_ptr()
                        bc.w_hh, bc.w_ih = cp["w_hh"].data_ptr(), cp["w_ih"].data_ptr()
                    else:   # Linear cell: W = [W_in | W_agg]
                        o["w_in"] = cp["w_lin"][:, :in_dim].contiguous()
                        o["w_agg"] = cp["w_lin"][:, in_dim:].contiguous()
                        bc.w_ih, bc.w_hh = o["w_in"].data_ptr(), o["w_agg"].data_ptr()
                    bc.g = g[(d, i)].data_ptr()
                    bc.g_in = (g[(d, i - 1)] if i > 0 else dxd[d]).data_ptr()
                    bc.da, bc.dgi, bc.dgh = o["da"].data_ptr(), o["dgi"].data_ptr(), o["dgh"].data_ptr()
                    res[(d, i)] = o
            import numpy as np
            ptrs = (C.POINTER(C.c_int32) * 2)()
            nl = (C.c_int32 * 2)()
            one = np.array([0, N], dtype=np.int32)   # agg_x: nothing couples the layers - one pseudo-layer of all rows
            for d in (0, 1):
                sd = one if agg_x else sched[d]
                ptrs[d] = sd.ctypes.data_as(C.POINTER(C.c_int32))
                nl[d] = len(sd) - 1
            with engine._span("variant_backward_run", x):
                engine.check(lib.dagnn_variant_backward_run(C.byref(plan.desc), C.byref(args), ptrs, nl, stream),
                             "dagnn_variant_backward_run")
                if agg_x:   # the aggregator's own reverse pass: one shot over all rows, into the gradient of x
                    for d in mod.dirs:
                        bca, ao = aggx[d]
                        if not bca.lands:
                            continue
                        dps = res[(d, 0)]["da"]
                        for i in range(1, L):
                            dps = dps + res[(d, i)]["da"]
                        ao["da"] = dps[:, :E].contiguous()
                        bca.da, bca.g, bca.g_in = ao["da"].data_ptr(), dxd[d].data_ptr(), dxd[d].data_ptr()
                        engine.check(lib.dagnn_variant_aggregator_backward(C.byref(plan.desc), C.byref(bca), d, E, 0, N, stream),
                                     "dagnn_variant_aggregator_backward")
            # ---- epilogue: parameter gradients (transposed products over all nodes) ----
            grads = {}
            jobs = []
            if recurr:
                for d in mod.dirs:
                    for i in range(L):
                        o = res[(d, i)]
                        jobs += [(o["dgi"], o["u"], True), (o["dgh"], o["a"], True)]
            wg = engine.wgrad(jobs, N, H, H) if (N > 0 and jobs) else None
            kq = 0
            for d in mod.dirs:
                for i in range(L):
                    o, cp = res[(d, i)], cellp[(d, i)]
                    if not recurr:   # Linear cell: dW = g^T [u ; a], db = sum g (g holds the total gradient of every row now)
                        gt = g[(d, i)]
                        grads[(d, i, "w_lin")] = torch.cat([gt.t() @ o["u"], gt.t() @ o["a"]], 1)
                        grads[(d, i, "b_lin")] = gt.sum(0)
                    else:
                        if wg is not None:
                            (gw_ih, gb_ih), (gw_hh, gb_hh) = wg[kq], wg[kq + 1]
                        else:
                            gw_ih, gb_ih, gw_hh, gb_hh = (torch.zeros_like(cp[n]) for n in ("w_ih", "b_ih", "w_hh", "b_hh"))
                        kq += 2
                        grads[(d, i, "w_ih")], grads[(d, i, "b_ih")] = gw_ih, gb_ih
                        grads[(d, i, "w_hh")], grads[(d, i, "b_hh")] = gw_hh, gb_hh
                    if agg_x:
                        if i == 0:
                            _, ao = aggx[d]
                            ag = _aggregator_grads(mod, mode, ao, cp, x, x, R, E, 0, dxd[d]) if "da" in ao else {}
                            for n in cp:
                                if n not in ("w_ih", "w_hh", "b_ih", "b_hh", "w_lin", "b_lin"):
                                    grads[(d, i, n)] = ag.get(n, torch.zeros_like(cp[n]))
                    else:
                        for n, v in _aggregator_grads(mod, mode, o, cp, h[d][i], o["u"], R, H, i, dxd[d]).items():
                            grads[(d, i, n)] = v
            dx = None
            if ctx.needs_input_grad[3]:
                dx = sum(dxd[d] for d in mod.dirs)
        return (None, None, None, dx) + tuple(grads[n] for n in names)
