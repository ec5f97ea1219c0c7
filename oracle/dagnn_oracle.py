"""CPU restatement of the reference's layer-by-layer DAGNN message-passing path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import this module; nothing under `dagnn_amd/` does.  It is the checker, never the thing
measured or shipped.

Parity status: **pinned**.  The reference has no tests or golden vectors of its own (SURVEY.md
§4), so the pin is against outputs of the reference itself: `tests/golden/make_golden.py` imports
the unmodified reference model files in the build container and writes `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function here against those files.

What is restated (plain torch on CPU, fp32 or fp64, no custom kernels), with the reference lines
each function follows:

* `ast_node_encoder`      - `ogbg-code/utils.py:26-28`
* `gru_cell`              - `torch.nn.GRUCell` as called at `ogbg-code/model/dagnn.py:181`
* `attn_aggregate`        - `AttnConv.forward/message` `dagnn.py:362-373` + PyG `propagate`,
                            `softmax` and scatter-add
* `recurrence_faithful`   - the three nested loops `dagnn.py:144-182` op for op: per-node scan of
                            the whole `edge_index` (:153-155), aggregation into all N rows (:179)
* `recurrence_csr`        - same math on a layer-sorted CSR (what the HIP path implements)
* `conv_variant`, `recurrence_variants` - the other aggregators / `agg_x` / `recurr=0` (`dagnn.py:159-169,232-313,379-409`)
* `code2_forward`         - `DAGNN.forward` `dagnn.py:128-215` (read-outs :184-202, heads :209-215)
* `code2_grads`           - one training step's loss + gradients, `ogbg-code/main_pyg.py:55-62`
* `dvae_forward/encode`   - `dvae/dagnn.py:99-184` (NA, `vids` key bias :130-139) and
                            `dvae/dagnn_bn.py:98-177` (BN); `dvae_grads`: gradients of the encoder
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

Tensor = torch.Tensor
ADDITIVE_ATTN = ("attn_h", "attn_x", "self_attn_h", "self_attn_x")


# ------------------------------------------------------------------------------ building blocks
def ast_node_encoder(sd: Dict[str, Tensor], x: Tensor, depth: Tensor, max_depth: int = 20,
                     prefix: str = "encoder.") -> Tensor:
    """type_emb[x0] + attr_emb[x1] + depth_emb[min(depth, max_depth)]; clamps `depth` IN PLACE
    like the reference does (`utils.py:27`)."""
    depth[depth > max_depth] = max_depth
    return (sd[prefix + "type_encoder.weight"][x[:, 0]] + sd[prefix + "attribute_encoder.weight"][x[:, 1]]
            + sd[prefix + "depth_encoder.weight"][depth])


def gru_cell(x: Tensor, h: Optional[Tensor], w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor) -> Tensor:
    """r,z = sigmoid(W_i x + b_i + W_h h + b_h); n = tanh(W_in x + b_in + r*(W_hn h + b_hn));
    h' = (1-z)*n + z*h.  `h=None` means zeros (`dagnn.py:172-173`).  Gate order (r, z, n)."""
    H = w_hh.shape[1]
    if h is None:
        h = x.new_zeros(x.shape[0], H)
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * h


def segment_softmax(logit: Tensor, seg: Tensor, num_seg: int) -> Tensor:
    """PyG-1.6 `softmax`: exp(x - segmax) / (segsum + 1e-16)."""
    mx = logit.new_full((num_seg,), float("-inf")).scatter_reduce_(0, seg, logit, "amax", include_self=True)
    ex = (logit - mx[seg]).exp()
    sm = logit.new_zeros(num_seg).scatter_add_(0, seg, ex)
    return ex / (sm[seg] + 1e-16)


def attn_aggregate(h_val: Tensor, key: Tensor, query: Optional[Tensor], tgt: Tensor, src: Tensor,
                   attn_w: Tensor, attn_b: Tensor, edge_emb: Optional[Tensor], num_rows: int) -> Tensor:
    """`AttnConv.message` (`dagnn.py:366-373`): logit = attn_lin([q_tgt ; key_src (+ e)]),
    alpha = softmax over edges sharing a target, out[tgt] += alpha * h_val[src].
    `query=None` is `SelfAttnConv.message` (`dagnn.py:297-310`): logit = attn_lin(key_src (+ e))."""
    k = key[src]
    if edge_emb is not None:
        k = k + edge_emb
    feat = k if query is None else torch.cat([query[tgt], k], dim=-1)
    logit = (feat @ attn_w.t() + attn_b).squeeze(-1)
    alpha = segment_softmax(logit, tgt, num_rows)
    out = h_val.new_zeros(num_rows, h_val.shape[1])
    return out.index_add_(0, tgt, h_val[src] * alpha.unsqueeze(-1))


def _vids(n: int, num_nodes: int, like: Tensor) -> Tensor:
    """one-hot of (node index mod num_nodes) (`dvae/dagnn.py:130-133`)."""
    v = like.new_zeros(n, num_nodes)
    v[torch.arange(n), torch.arange(n) % num_nodes] = 1
    return v


class _Cfg:
    def __init__(self, sd, dirs, L, H, cell_prefix, has_edge_enc, vid_nodes, agg="attn_h"):
        self.sd, self.dirs, self.L, self.H = sd, dirs, L, H
        self.cell_prefix, self.has_edge_enc, self.vid_nodes = cell_prefix, has_edge_enc, vid_nodes
        # aggregator strings of src/constants.py:12-27 handled here: attn_h, attn_x, self_attn_h, self_attn_x
        self.keys_from_x = "_x" in agg        # dagnn.py:175-177: h_attn = G.x
        self.use_query = "self_attn" not in agg

    def cell(self, d, i):
        p = "%s%d.%d." % (self.cell_prefix, d, i)
        return (self.sd[p + "weight_ih"], self.sd[p + "weight_hh"], self.sd[p + "bias_ih"], self.sd[p + "bias_hh"])

    def aggr(self, d, i):
        p = "node_aggr_%d.%d." % (d, i)
        ee = (self.sd[p + "edge_encoder.weight"], self.sd[p + "edge_encoder.bias"]) if self.has_edge_enc else None
        return self.sd[p + "attn_lin.weight"], self.sd[p + "attn_lin.bias"], ee


# ------------------------------------------------------------------------------ the recurrence
def recurrence_faithful(cfg: _Cfg, x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor],
                        layers: Sequence[Tensor]) -> List[List[Tensor]]:
    """Op-for-op mirror of `dagnn.py:141-182` (and `dvae/dagnn.py:106-145`): reproduces the
    reference's cost profile (O(F*E) edge scan per step, full [N,H] scatter per micro-step)."""
    N, H = x.shape[0], cfg.H
    ids = torch.arange(N)
    h = [[x.new_zeros(N, H) for _ in range(cfg.L)] for _ in cfg.dirs]
    T = int(layers[0].max()) + 1
    for d in cfg.dirs:
        tgt_row, src_row = (1, 0) if d == 0 else (0, 1)
        for t in range(T):
            layer = ids[layers[d] == t]
            inp = x[layer]
            if t > 0:
                le = torch.cat([(edge_index[1 - d] == n).nonzero().squeeze(-1) for n in layer], dim=-1)
                lp = edge_index[:, le]
            for i in range(cfg.L):
                if t == 0:
                    ps = None
                else:
                    aw, ab, ee = cfg.aggr(d, i)
                    emb = edge_attr[le] @ ee[0].t() + ee[1] if ee is not None else None
                    hv = h[d][i]
                    if cfg.vid_nodes:
                        vid = _vids(N, cfg.vid_nodes, x)
                        key = torch.cat([hv, vid], -1)
                        q = torch.cat([h[d][i - 1], vid], -1) if i > 0 else x
                    elif cfg.keys_from_x:
                        key, q = x, x
                    else:
                        key, q = hv, (h[d][i - 1] if i > 0 else x)
                    ps = attn_aggregate(hv, key, q if cfg.use_query else None, lp[tgt_row], lp[src_row], aw, ab, emb,
                                        N)[layer]
                inp = gru_cell(inp, ps, *cfg.cell(d, i))
                h[d][i][layer] += inp
    return h


def build_layer_csr(edge_index: Tensor, layer: Tensor, d: int):
    """Frontier-ordered edge lists for direction d: nodes sorted by (layer, id); the in-edges
    (d=0) / out-edges (d=1) of each node in original edge order - the order the reference's
    scan produces (`dagnn.py:153-156`)."""
    N = layer.shape[0]
    order = torch.argsort(layer * N + torch.arange(N))  # (layer, id)
    pos = torch.empty(N, dtype=torch.long)
    pos[order] = torch.arange(N)
    T = int(layer.max()) + 1 if N else 0
    lptr = torch.zeros(T + 1, dtype=torch.long)
    lptr[1:] = torch.bincount(layer, minlength=T).cumsum(0)
    tgt = edge_index[1 - d]
    other = edge_index[d]
    E = tgt.shape[0]
    eorder = torch.argsort(pos[tgt] * max(E, 1) + torch.arange(E))  # stable by target position
    rowptr = torch.zeros(N + 1, dtype=torch.long)
    rowptr[1:] = torch.bincount(pos[tgt], minlength=N).cumsum(0)
    return order, lptr, rowptr, other[eorder], eorder


def recurrence_csr(cfg: _Cfg, x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor],
                   layers: Sequence[Tensor]) -> List[List[Tensor]]:
    """Same math as `recurrence_faithful` on a layer-sorted CSR; the aggregate is computed for the
    F frontier rows only.  The full logit (query term, bias, edge embedding) is kept."""
    N, H = x.shape[0], cfg.H
    h = [[x.new_zeros(N, H) for _ in range(cfg.L)] for _ in cfg.dirs]
    vid = _vids(N, cfg.vid_nodes, x) if cfg.vid_nodes else None
    for d in cfg.dirs:
        order, lptr, rowptr, col, eid = build_layer_csr(edge_index, layers[d], d)
        for t in range(lptr.shape[0] - 1):
            p0, p1 = int(lptr[t]), int(lptr[t + 1])
            rows = order[p0:p1]
            inp = x[rows]
            if t > 0:
                e0, e1 = int(rowptr[p0]), int(rowptr[p1])
                src = col[e0:e1]
                seg = torch.repeat_interleave(torch.arange(p1 - p0), rowptr[p0 + 1:p1 + 1] - rowptr[p0:p1])
                ea = edge_attr[eid[e0:e1]] if edge_attr is not None else None
            for i in range(cfg.L):
                if t == 0:
                    ps = None
                else:
                    aw, ab, ee = cfg.aggr(d, i)
                    hv = h[d][i]
                    key = (x if cfg.keys_from_x else hv)[src]
                    qn = (x if cfg.keys_from_x else (h[d][i - 1] if i > 0 else x))[rows]
                    if vid is not None:
                        key = torch.cat([key, vid[src]], -1)
                        if i > 0:
                            qn = torch.cat([qn, vid[rows]], -1)
                    if ee is not None:
                        key = key + (ea @ ee[0].t() + ee[1])
                    feat = torch.cat([qn[seg], key], -1) if cfg.use_query else key
                    logit = (feat @ aw.t() + ab).squeeze(-1)
                    alpha = segment_softmax(logit, seg, p1 - p0)
                    ps = x.new_zeros(p1 - p0, H).index_add_(0, seg, hv[src] * alpha.unsqueeze(-1))
                inp = gru_cell(inp, ps, *cfg.cell(d, i))
                h[d][i][rows] = inp
    return h


# ------------------------------------------------------------------------------ variants (row a12)
def _propagate(msg: Tensor, index: Tensor, num_nodes: int, reduce: str) -> Tensor:
    """PyG-1.6 `propagate` tail: scatter the messages onto `index`; rows nothing lands on are zero."""
    out = msg.new_zeros(num_nodes, msg.shape[1])
    if reduce == "add":
        return out.index_add_(0, index, msg)
    return out.scatter_reduce_(0, index.view(-1, 1).expand_as(msg), msg, "amax", include_self=False)


def conv_variant(sd: Dict[str, Tensor], prefix: str, agg: str, vals: Tensor, lp: Tensor, reverse: bool,
                 edge_attr: Optional[Tensor], h_attn: Optional[Tensor], h_attn_q: Optional[Tensor]) -> Tensor:
    """One call of the reference's conv modules on the step's edge list `lp` [2, E_f], result for ALL N nodes:
    `AggConv` add|max (`dagnn.py:232-251`), `GatedSumConv` (:254-276), `SelfAttnConv` (:279-313), `AttnConv`
    (:347-376), `MultAttnConv` (:379-409).  `reverse` = the module was built with flow target_to_source (messages go
    from lp[1] to lp[0]); the shared `AggConv` never is (`dagnn.py:74-75`)."""
    N = vals.shape[0]
    i_row, j_row = (0, 1) if reverse else (1, 0)
    e = None
    if prefix + "edge_encoder.weight" in sd and edge_attr is not None:
        e = edge_attr @ sd[prefix + "edge_encoder.weight"].t() + sd[prefix + "edge_encoder.bias"]
    hj = vals[lp[j_row]]
    if agg in ("add", "max"):
        return _propagate(hj + e if e is not None else hj, lp[i_row], N, agg)
    if agg == "gated_sum":
        m = hj + e if e is not None else hj
        gate = torch.sigmoid(m @ sd[prefix + "gate.0.weight"].t() + sd[prefix + "gate.0.bias"])
        mapped = m @ sd[prefix + "mapper.weight"].t()
        if prefix + "mapper.bias" in sd:
            mapped = mapped + sd[prefix + "mapper.bias"]
        return _propagate(gate * mapped, lp[i_row], N, "add")
    k = (h_attn if h_attn is not None else vals)[lp[j_row]]
    if e is not None:
        k = k + e
    if "mattn" in agg:
        ql = h_attn_q[lp[i_row]] @ sd[prefix + "attn_linl.weight"].t() + sd[prefix + "attn_linl.bias"]
        kr = k @ sd[prefix + "attn_linr.weight"].t() + sd[prefix + "attn_linr.bias"]
        logit = (ql * kr).sum(1)
    elif "self_attn" in agg:
        logit = (k @ sd[prefix + "attn_lin.weight"].t() + sd[prefix + "attn_lin.bias"]).squeeze(-1)
    else:
        logit = (torch.cat([h_attn_q[lp[i_row]], k], -1) @ sd[prefix + "attn_lin.weight"].t()
                 + sd[prefix + "attn_lin.bias"]).squeeze(-1)
    alpha = segment_softmax(logit, lp[i_row], N)
    return _propagate(hj * alpha.unsqueeze(-1), lp[i_row], N, "add")


def recurrence_variants(sd: Dict[str, Tensor], x: Tensor, edge_index: Tensor, edge_attr: Optional[Tensor],
                        layers: Sequence[Tensor], *, dirs: Sequence[int], L: int, H: int, agg: str, agg_x: bool,
                        recurr: int) -> List[List[Tensor]]:
    """Op-for-op mirror of `dagnn.py:144-182` for every constructor string (aggregator, `agg_x`, `recurr=0`): per-node
    edge scan, convs evaluated for all N nodes, frontier rows taken afterwards."""
    N = x.shape[0]
    ids = torch.arange(N)
    agg_attn, agg_attn_x = "attn" in agg, "_x" in agg
    shared = agg in ("add", "max")   # one AggConv module for both directions: never reversed
    h = [[x.new_zeros(N, H) for _ in range(L)] for _ in dirs]
    T = int(layers[0].max()) + 1
    for d in dirs:
        for t in range(T):
            layer = ids[layers[d] == t]
            inp = x[layer]
            if t > 0:
                le = torch.cat([(edge_index[1 - d] == n).nonzero().squeeze(-1) for n in layer], dim=-1)
                lp, ea = edge_index[:, le], (edge_attr[le] if edge_attr is not None else None)
                if agg_x:
                    ps_x = conv_variant(sd, "node_aggr_%d.0." % d, agg, x, lp, d == 1 and not shared, ea,
                                        x if agg_attn else None, x if agg_attn else None)[layer]
                    if ps_x.shape[1] < H:
                        ps_x = torch.cat([ps_x, ps_x.new_zeros(ps_x.shape[0], H - ps_x.shape[1])], -1)
            for i in range(L):
                if t == 0:
                    ps = None if recurr else x.new_zeros(inp.shape[0], H)
                elif agg_x:
                    ps = ps_x
                else:
                    h_attn = h_attn_q = None
                    if agg_attn:
                        h_attn = x if agg_attn_x else h[d][i]
                        h_attn_q = x if agg_attn_x else (h[d][i - 1] if i > 0 else x)
                    ps = conv_variant(sd, "node_aggr_%d.%d." % (d, i), agg, h[d][i], lp, d == 1 and not shared, ea,
                                      h_attn, h_attn_q)[layer]
                p = "cells_%d.%d." % (d, i)
                if recurr:
                    inp = gru_cell(inp, ps, sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], sd[p + "bias_hh"])
                else:
                    inp = torch.cat([inp, ps], 1) @ sd[p + "weight"].t() + sd[p + "bias"]
                h[d][i][layer] += inp
    return h


# ------------------------------------------------------------------------------ read-outs
def _pool(x: Tensor, batch: Tensor, how: str) -> Tensor:
    B = int(batch.max()) + 1
    idx = batch.view(-1, 1).expand_as(x)
    if how == "max":
        return x.new_zeros(B, x.shape[1]).scatter_reduce_(0, idx, x, "amax", include_self=False)
    s = x.new_zeros(B, x.shape[1]).scatter_add_(0, idx, x)
    if how in ("add", "attn"):  # P_ATTN is a softmax over a size-1 dim == sum pooling (Appendix D)
        return s
    if how == "mean":
        cnt = torch.bincount(batch, minlength=B).clamp(min=1).to(x.dtype)
        return s / cnt.view(-1, 1)
    raise ValueError(how)


def _cast(sd, dtype, keep_graph=False):
    if keep_graph:  # code2_grads: the parameters are autograd leaves
        return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    return {k: (v.detach().to(dtype) if v.is_floating_point() else v.detach()) for k, v in sd.items()}


def code2_forward(sd: Dict[str, Tensor], G, *, num_layers: int = 2, bidirectional: bool = True,
                  out_wx: bool = False, out_pool_all: bool = False, out_pool: str = "max",
                  max_seq_len: int = 5, num_class: int = 0, mode: str = "csr",
                  dtype: torch.dtype = torch.float32, agg: str = "attn_h", keep_graph: bool = False,
                  agg_x: bool = False, recurr: int = 1):
    """`DAGNN.forward` of `ogbg-code/model/dagnn.py:128-215` for the additive-attention aggregators
    (`agg` in attn_h, attn_x, self_attn_h, self_attn_x), `recurr=1`, `agg_x=False`.  Reproduces the side effects on G (`G.x`, `G.h`, `G.bi_layer_index`, clamped
    `G.node_depth`, and `G.batch` on the unidirectional branch).  Returns a list of logits per
    head, or one tensor when `num_class > 0`."""
    sd = _cast(sd, dtype, keep_graph)
    dirs = [0, 1] if bidirectional else [0]
    H = sd["cells_0.0.weight_hh"].shape[1] if recurr else sd["cells_0.0.weight"].shape[0]
    G.bi_layer_index = torch.stack([torch.stack([G._bi_layer_idx0, G._bi_layer_index0], 0),
                                    torch.stack([G._bi_layer_idx1, G._bi_layer_index1], 0)], 0)
    G.x = ast_node_encoder(sd, G.x, G.node_depth.view(-1))
    layers = [G.bi_layer_index[0][0], G.bi_layer_index[1][0]]
    ea = G.edge_attr.to(dtype) if getattr(G, "edge_attr", None) is not None else None
    if agg in ADDITIVE_ATTN and not agg_x and recurr:
        cfg = _Cfg(sd, dirs, num_layers, H, "cells_", "node_aggr_0.0.edge_encoder.weight" in sd, 0, agg)
        rec = recurrence_faithful if mode == "faithful" else recurrence_csr
        G.h = rec(cfg, G.x, G.edge_index, ea, layers)
    else:   # the other constructor strings (row a12): one faithful mirror
        G.h = recurrence_variants(sd, G.x, G.edge_index, ea, layers, dirs=dirs, L=num_layers, H=H, agg=agg,
                                  agg_x=agg_x, recurr=recurr)

    def out_nodes(reverse):  # dagnn.py:119-126
        return G.bi_layer_index[0][1][G.bi_layer_index[0][0] == 0] if reverse else \
            G.bi_layer_index[1][1][G.bi_layer_index[1][0] == 0]

    if bidirectional and not out_pool_all:
        outs = []
        for d in (0, 1):
            idx = out_nodes(reverse=d)
            hd = torch.cat(([G.x] if out_wx else []) + [G.h[d][l] for l in range(num_layers)], -1)
            outs.append(_pool(hd[idx], G.batch[idx], out_pool))
        out = torch.cat(outs, -1)
    else:
        G.h = torch.cat(([G.x] if out_wx else []) + [G.h[d][l] for d in dirs for l in range(num_layers)], -1)
        if not out_pool_all:
            idx = out_nodes(reverse=0)
            G.h, G.batch = G.h[idx], G.batch[idx]
        out = _pool(G.h, G.batch, out_pool)
    if num_class > 0:
        return out @ sd["graph_pred_linear.weight"].t() + sd["graph_pred_linear.bias"]
    return [out @ sd["graph_pred_linear_list.%d.weight" % s].t() + sd["graph_pred_linear_list.%d.bias" % s]
            for s in range(max_seq_len)]


def code2_grads(sd: Dict[str, Tensor], G, y: Tensor, *, dtype: torch.dtype = torch.float32, **kw):
    """Loss and parameter gradients of one training step as `ogbg-code/main_pyg.py:55-62` computes them:
    `loss = mean_s CrossEntropy(pred_list[s], y_arr[:, s])`, `loss.backward()`.  Plain torch autograd
    through `code2_forward`; returns `(loss, {parameter name: gradient})` (zeros for unused parameters)."""
    leaves = {k: (v.detach().to(dtype).clone().requires_grad_(True) if v.is_floating_point() else v)
              for k, v in sd.items()}
    pred = code2_forward(leaves, G, dtype=dtype, keep_graph=True, **kw)
    pred = pred if isinstance(pred, (list, tuple)) else [pred]
    loss = sum(torch.nn.functional.cross_entropy(p, y[:, s]) for s, p in enumerate(pred)) / len(pred)
    names = [k for k, v in leaves.items() if v.is_floating_point()]
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return loss.detach(), {k: (torch.zeros_like(leaves[k]) if g is None else g) for k, g in zip(names, gs)}


def recurrence_dvae_plain(sd: Dict[str, Tensor], x: Tensor, edge_index: Tensor, layers: Sequence[Tensor], dirs, L: int,
                          H: int, agg: str, vid_nodes: int) -> List[List[Tensor]]:
    """The loop nest of `dvae/dagnn.py:106-145` (`dvae/dagnn_bn.py:104-137`) for agg in {gated_sum, add, max}, op for op:
    `GatedSumConv` (`dagnn.py:269-299`) on hs = [state ; one-hot vertex id] with the base class's mapper / gate of the
    direction (`:60-65`), `AggConv` (`:242-267`) on the plain states - one conv per direction (`reverse=True` for the
    second: messages flow to the frontier in both); rows nothing lands on are zero (PyG-1.6 propagate)."""
    N = x.shape[0]
    ids = torch.arange(N)
    h = [[x.new_zeros(N, H) for _ in range(L)] for _ in range(2)]
    T = int(layers[0].max()) + 1
    for d in dirs:
        tgt_row, src_row = (1, 0) if d == 0 else (0, 1)
        name = "forward" if d == 0 else "backward"
        for t in range(T):
            layer = ids[layers[d] == t]
            inp = x[layer]
            if t > 0:
                le = torch.cat([(edge_index[1 - d] == n).nonzero().squeeze(-1) for n in layer], dim=-1)
                lp = edge_index[:, le]
            for i in range(L):
                ps = None
                if t > 0:
                    hv = h[d][i]
                    if agg == "gated_sum":
                        hs = torch.cat([hv, _vids(N, vid_nodes, x)], -1) if vid_nodes else hv
                        hj = hs[lp[src_row]]
                        gate = torch.sigmoid(hj @ sd["gate_%s.%d.0.weight" % (name, i)].t() + sd["gate_%s.%d.0.bias" % (name, i)])
                        msg = gate * (hj @ sd["mapper_%s.%d.0.weight" % (name, i)].t())
                        ps = _propagate(msg, lp[tgt_row], N, "add")[layer]
                    else:
                        ps = _propagate(hv[lp[src_row]], lp[tgt_row], N, agg)[layer]
                p = "cells_%d.%d." % (d, i)
                inp = gru_cell(inp, ps, sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], sd[p + "bias_hh"])
                h[d][i] = h[d][i].index_add(0, layer, inp)
    return [h[d] for d in dirs]


def dvae_forward(sd: Dict[str, Tensor], G, *, num_layers: int = 2, bidirectional: bool = False,
                 num_nodes: int = 8, vids: bool = True, mode: str = "csr",
                 dtype: torch.dtype = torch.float32, keep_graph: bool = False, out_pool_all: bool = False,
                 out_pool: str = "max", agg: str = "attn_h") -> Tensor:
    """`DAGNN.forward` of `dvae/dagnn.py:99-175` (`vids=True`, NA) or `DAGNN_BN.forward` of
    `dvae/dagnn_bn.py:98-168` (`vids=False`), `out_pool_all=False`: read-out = the end vertex of
    every graph for d=0 and the start vertex for d=1 (fixed stride `num_nodes`).  `agg`: `attn_h` or `self_attn_h`
    (`dvae/dagnn.py:49-59`: `SelfAttnConv` scores the keys alone - no query half in `attn_lin`)."""
    sd = _cast(sd, dtype, keep_graph)
    dirs = [0, 1] if bidirectional else [0]
    H = sd["cells_0.0.weight_hh"].shape[1]
    x = G.x.to(dtype)
    layers = [G.bi_layer_index[0][0], G.bi_layer_index[1][0]]
    if agg in ("gated_sum", "add", "max"):
        h = recurrence_dvae_plain(sd, x, G.edge_index, layers, dirs, num_layers, H, agg, num_nodes if vids else 0)
    elif agg not in ("attn_h", "self_attn_h"):
        raise NotImplementedError(agg)
    else:
        cfg = _Cfg(sd, dirs, num_layers, H, "cells_", False, num_nodes if vids else 0, agg)
        rec = recurrence_faithful if mode == "faithful" else recurrence_csr
        h = rec(cfg, x, G.edge_index, None, layers)
    N = x.shape[0]
    if out_pool_all:   # dvae/dagnn.py:163-172: per-node projection, then pooling over all nodes of a graph
        G.h = torch.cat([h[q][l] for q in range(len(dirs)) for l in range(num_layers)], -1)
        if bidirectional:
            G.h = G.h @ sd["hg_unify.0.weight"].t() + sd["hg_unify.0.bias"]
        elif num_layers > 1:
            G.h = G.h @ sd["out_linear.weight"].t() + sd["out_linear.bias"]
        return _pool(G.h, G.batch, out_pool)
    first = torch.arange(0, N, num_nodes)
    last = first + (num_nodes - 1)
    if bidirectional:
        h0 = torch.cat([h[0][l][last] for l in range(num_layers)], -1)
        h1 = torch.cat([h[1][l][first] for l in range(num_layers)], -1)
        G.h = torch.cat([h0, h1], -1)
        G.batch = G.batch[first]
        return G.h @ sd["hg_unify.0.weight"].t() + sd["hg_unify.0.bias"]
    G.h = torch.cat([h[0][l][last] for l in range(num_layers)], -1)
    G.batch = G.batch[last]
    if num_layers > 1:
        return G.h @ sd["out_linear.weight"].t() + sd["out_linear.bias"]
    return G.h


def dvae_encode(sd, G, **kw):
    """`encode` (`dvae/dagnn.py:177-184`): (mu, logvar) = fc1(Hg), fc2(Hg)."""
    dtype = kw.get("dtype", torch.float32)
    Hg = dvae_forward(sd, G, **kw)
    sdc = _cast({k: sd[k] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")}, dtype,
                kw.get("keep_graph", False))
    return Hg @ sdc["fc1.weight"].t() + sdc["fc1.bias"], Hg @ sdc["fc2.weight"].t() + sdc["fc2.bias"]


def dvae_grads(sd: Dict[str, Tensor], G, r1: Tensor, r2: Tensor, *, dtype: torch.dtype = torch.float32, **kw):
    """Gradients of `loss = <mu, r1> + <logvar, r2>` through `dvae_encode` (plain torch autograd): the checker of the
    encoder's backward pass (the VAE loss itself needs the decoder, which is outside the path).  Returns
    `(loss, {parameter name: gradient})` for the parameters that take part."""
    leaves = {k: (v.detach().to(dtype).clone().requires_grad_(True) if v.is_floating_point() else v)
              for k, v in sd.items()}
    mu, logvar = dvae_encode(leaves, G, dtype=dtype, keep_graph=True, **kw)
    loss = (mu * r1.to(dtype)).sum() + (logvar * r2.to(dtype)).sum()
    names = [k for k, v in leaves.items() if v.is_floating_point()]
    gs = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    out = {k: g for k, g in zip(names, gs) if g is not None}
    # the D-VAE modules register their encoder GRUs under two names (cells_d == grue_forward / grue_backward,
    # dvae/dagnn.py:73-75): report a gradient under every name of the same storage
    by_ptr = {}
    for k, v in sd.items():
        by_ptr.setdefault(v.data_ptr(), []).append(k)
    for ks in by_ptr.values():
        have = [k for k in ks if k in out]
        for k in ks:
            if have and k not in out:
                out[k] = out[have[0]]
    return loss.detach(), out
