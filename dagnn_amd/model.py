"""`DAGNN` - drop-in for the reference's `ogbg-code/model/dagnn.py:16-215` on MI355X.

Same constructor signature, same `state_dict` keys and shapes (checkpoints written by the
reference's `utils2.create_checkpoint` load unchanged), same `forward(G)` contract on a PyG-style
batch including its side effects on `G` (`G.bi_layer_index`, `G.x` replaced by the embedding,
`G.node_depth` clamped, `G.h`).  The layer-by-layer gather -> attention-aggregate -> GRU path
runs in hand-written HIP (libdagnn_hip.so); there is no CPU or eager-PyTorch fallback for it.

Aggregators: the additive-attention family (`attn_h` - every BASELINE config - `attn_x`, `self_attn_h`,
`self_attn_x`) runs in HIP, forward and backward.  The reference's other constructor strings (`mattn_h`,
`gated_sum`, `add`, `max`, `agg_x=True`, `recurr=0`; SURVEY.md §8(a) row a12: exercised by no BASELINE
configuration) keep the same contract on the generic HIP kernels of `csrc/variants.hip` (forward) and
`csrc/variants_bwd.hip` (the reverse sweep of a training step), marshalled by `dagnn_amd/variants.py`; only what those
kernels do not take (more than 8 cells, widths that are not multiples of 4, more than two edge features behind an edge
encoder) trains on torch-ROCm ops, and says so once (`variants.warn_torch_path`).
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import constants as K
from . import engine
from .core import DerivedCache, built_marker, default_schedule, derive_cell, meet_built, num_graphs_of, pack_lockstep, run_stack


class ASTNodeEncoder(nn.Module):
    """Node embedding of `ogbg-code/utils.py:6-28` (same parameter names)."""

    def __init__(self, emb_dim, num_nodetypes, num_nodeattributes, max_depth):
        super().__init__()
        self.max_depth = max_depth
        self.type_encoder = nn.Embedding(num_nodetypes, emb_dim)
        self.attribute_encoder = nn.Embedding(num_nodeattributes, emb_dim)
        self.depth_encoder = nn.Embedding(self.max_depth + 1, emb_dim)

    def forward(self, x, depth):
        if x.is_cuda and self.type_encoder.weight.shape[1] % 4 == 0 and depth.dtype == torch.int64 \
                and depth.is_contiguous():
            tables = (self.type_encoder.weight, self.attribute_encoder.weight, self.depth_encoder.weight)
            if torch.is_grad_enabled() and any(t.requires_grad for t in tables):
                from .autograd import EncodeAST
                return EncodeAST.apply(x, depth, *tables, self.max_depth)
            return engine.encode_ast(x, depth, *tables, self.max_depth)
        # generic torch path (odd widths): same math as utils.py:26-28
        depth[depth > self.max_depth] = self.max_depth
        return self.type_encoder(x[:, 0]) + self.attribute_encoder(x[:, 1]) + self.depth_encoder(depth)


class _EdgeAttnParams(nn.Module):
    """Parameter holder with the names of the reference's `AttnConv` (`dagnn.py:347-359`)."""

    def __init__(self, attn_q_dim, emb_dim, attn_dim=0, num_relations=1, reverse=False):
        super().__init__()
        attn_dim = attn_dim if attn_dim > 0 else emb_dim
        self.wea = num_relations > 1
        if self.wea:
            self.edge_encoder = nn.Linear(num_relations, attn_dim)
        self.attn_lin = nn.Linear(attn_q_dim + attn_dim, 1)
        self.reverse = reverse


class _SelfAttnParams(nn.Module):  # dagnn.py:279-290
    def __init__(self, emb_dim, attn_dim=0, num_relations=1, reverse=False):
        super().__init__()
        attn_dim = attn_dim if attn_dim > 0 else emb_dim
        self.wea = num_relations > 1
        if self.wea:
            self.edge_encoder = nn.Linear(num_relations, attn_dim)
        self.attn_lin = nn.Linear(attn_dim, 1)


class _MultAttnParams(nn.Module):  # dagnn.py:379-392
    def __init__(self, attn_q_dim, emb_dim, attn_dim=0, num_relations=1, reverse=False):
        super().__init__()
        attn_dim = attn_dim if attn_dim > 0 else emb_dim
        self.wea = num_relations > 1
        if self.wea:
            self.edge_encoder = nn.Linear(num_relations, attn_dim)
        self.attn_linl = nn.Linear(attn_q_dim, attn_q_dim)
        self.attn_linr = nn.Linear(attn_dim, attn_q_dim)


class _GatedSumParams(nn.Module):  # dagnn.py:254-265
    def __init__(self, emb_dim, num_relations=1, mapper_bias=True, reverse=False):
        super().__init__()
        self.wea = num_relations > 1
        if self.wea:
            self.edge_encoder = nn.Linear(num_relations, emb_dim)
        self.mapper = nn.Linear(emb_dim, emb_dim, bias=mapper_bias)
        self.gate = nn.Sequential(nn.Linear(emb_dim, emb_dim), nn.Sigmoid())


class _AggParams(nn.Module):  # dagnn.py:232-241
    def __init__(self, agg, num_relations=1, emb_dim=0):
        super().__init__()
        self.wea = num_relations > 1
        if self.wea:
            self.edge_encoder = nn.Linear(num_relations, emb_dim)


def _init_encoder(word_vectors, emb_dims):  # dagnn.py:218-223
    if word_vectors is not None:
        return nn.EmbeddingBag.from_pretrained(word_vectors, freeze=True, mode="sum")
    if len(emb_dims) > 0:
        return nn.EmbeddingBag(emb_dims[0], emb_dims[1], mode="sum")
    return None


class _HeadsLinear(torch.autograd.Function):
    """`out @ Wcat^T + bcat` for the S heads at once (their parameters are views of `wcat` / `bcat`: `DAGNN._head_storage`)
    with the gradients of the S weights and biases as views of ONE product."""

    @staticmethod
    def forward(ctx, out, wcat, bcat, V, *params):
        ctx.save_for_backward(out)
        ctx.wcat, ctx.V, ctx.S = wcat, V, len(params) // 2
        ctx.needs = (out.requires_grad, any(p.requires_grad for p in params))
        return torch.addmm(bcat, out, wcat.t())

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        wcat, V, S = ctx.wcat, ctx.V, ctx.S
        if g.stride(1) != 1:
            g = g.contiguous()
        d_out = g @ wcat if ctx.needs_input_grad[0] else None
        if ctx.needs[1]:
            D = out.shape[1]
            if g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0 and D % 2 == 0 and out.is_contiguous() and g.dtype == torch.float32:
                # (rows pitched to a multiple of 4 floats - what train.seq_cross_entropy hands back: csrc/wgrad.hip's own tiles
                # for a product that reduces 128 rows into 25 010 x 1 024 outputs, bias sums on the way)
                dw = torch.empty(S * V, D, dtype=torch.float32, device=g.device)
                db = torch.empty(S * V, dtype=torch.float32, device=g.device)
                engine.check(engine._lib.load().dagnn_tn_product(g.data_ptr(), g.stride(0), out.data_ptr(), out.stride(0), g.shape[0],
                                                                S * V, D, dw.data_ptr(), db.data_ptr(), engine._stream(g)),
                             "dagnn_tn_product")
            else:
                dw = g.t() @ out            # [S V, D]: the S weight gradients, one product
                db = g.sum(0)
            gw, gb = list(dw.split(V, 0)), list(db.split(V, 0))
        else:
            gw, gb = [None] * S, [None] * S
        return (d_out, None, None, None, *gw, *gb)


class DAGNN(nn.Module):
    """See module docstring.  Constructor mirrors `dagnn.py:18-112` argument for argument."""

    _plain_dataflow_ok = True   # (variants.run_plain_dataflow: `add` / `max` of THIS class - one shared AggConv - on the dataflow kernel)

    def __init__(self, num_vocab, max_seq_len, emb_dim, hidden_dim, out_dim,
                 num_rels=2, w_edge_attr=True, num_layers=2, bidirectional=True, mapper_bias=True,
                 agg_x=False, agg=K.NA_ATTN_H, out_wx=True, out_pool_all=True, out_pool=K.P_MAX, encoder=None,
                 dropout=0.0, word_vectors=None, emb_dims=[], activation=None, num_class=0, recurr=1):
        super().__init__()
        self.num_class = num_class
        self.num_vocab = num_vocab
        self.max_seq_len = max_seq_len
        if agg_x and hidden_dim < emb_dim:
            raise ValueError('Hidden dimension too small for input.')

        self.agg = agg
        self.agg_x = agg_x
        self.agg_attn = "attn" in agg
        self.agg_attn_x = "_x" in agg
        self.bidirectional = bidirectional
        self.dirs = [0, 1] if bidirectional else [0]
        self.num_layers = num_layers
        self.out_wx = out_wx
        self.output_all = out_pool_all
        self.out_pool = out_pool
        self.recurr = recurr
        self.emb_dim = emb_dim
        self.hidden_dim = hidden_dim
        nd = len(self.dirs)
        self.out_hidden_dim = emb_dim * nd + hidden_dim * nd * num_layers if out_wx else hidden_dim * nd * num_layers

        self.encoder = encoder if encoder is not None else _init_encoder(word_vectors, emb_dims)

        # parameter creation order follows the reference so that seeded default inits coincide
        num_rels = num_rels if w_edge_attr else 1
        self.num_rels = num_rels
        pred_dim = emb_dim if agg_x else hidden_dim
        attn_dim = emb_dim if "_x" in agg else hidden_dim
        if "self_attn" in agg:
            self.node_aggr_0 = nn.ModuleList([_SelfAttnParams(attn_dim, num_relations=num_rels)
                                              for _ in range(num_layers)])
            self.node_aggr_1 = nn.ModuleList([_SelfAttnParams(attn_dim, num_relations=num_rels, reverse=True)
                                              for _ in range(num_layers)])
        elif "attn" in agg:
            op = _MultAttnParams if "mattn" in agg else _EdgeAttnParams
            self.node_aggr_0 = nn.ModuleList([
                op(emb_dim if l == 0 else attn_dim, pred_dim, num_relations=num_rels, attn_dim=attn_dim)
                for l in range(num_layers)])
            self.node_aggr_1 = nn.ModuleList([
                op(emb_dim if l == 0 else attn_dim, pred_dim, num_relations=num_rels, attn_dim=attn_dim, reverse=True)
                for l in range(num_layers)])
        elif agg == K.NA_GATED_SUM:
            self.node_aggr_0 = nn.ModuleList([_GatedSumParams(pred_dim, num_rels, mapper_bias=mapper_bias)
                                              for _ in range(num_layers)])
            self.node_aggr_1 = nn.ModuleList([_GatedSumParams(pred_dim, num_rels, mapper_bias=mapper_bias,
                                                              reverse=True) for _ in range(num_layers)])
        else:
            node_aggr = _AggParams(agg, num_rels, pred_dim)
            self.node_aggr_0 = self.node_aggr_1 = nn.ModuleList([node_aggr for _ in range(num_layers)])

        for d in self.dirs:
            if recurr:
                cells = [nn.GRUCell(emb_dim if l == 0 else hidden_dim, hidden_dim) for l in range(num_layers)]
            else:
                cells = [nn.Linear((emb_dim if l == 0 else hidden_dim) + hidden_dim, hidden_dim)
                         for l in range(num_layers)]
            setattr(self, "cells_{}".format(d), nn.ModuleList(cells))

        if out_pool == K.P_ATTN:
            dd = int(self.out_hidden_dim / 2) if self.bidirectional and not self.output_all else self.out_hidden_dim
            self.self_attn_linear_out = nn.Linear(dd, 1)

        self.dropout = nn.Dropout(dropout)
        if self.num_class > 0:
            self.graph_pred_linear = nn.Linear(self.out_hidden_dim, self.num_class)
        else:
            self.graph_pred_linear_list = nn.ModuleList()
            if self.num_vocab == 1:
                self.graph_pred_linear_list.append(nn.Sequential(nn.Linear(self.out_hidden_dim, self.num_vocab),
                                                                 nn.ReLU()))
            else:
                for _ in range(max_seq_len):
                    self.graph_pred_linear_list.append(nn.Linear(self.out_hidden_dim, self.num_vocab))

        self._derived = {}
        self._head_cache = DerivedCache()
        self._arenas = {}  # per device: granule buffers of the persistent tail kernel
        self.schedule = default_schedule()  # 'lockstep' (frontier launches) or 'pergraph' (persistent workgroups)
        self.variant_backend = "hip"       # constructor-string variants (a12): 'hip' = csrc/variants.hip forward and
                                            # csrc/variants_bwd.hip reverse sweep; 'torch' = always the differentiable
                                            # torch-ROCm ops (the checker of the HIP sweep in the GPU tests)

    # ------------------------------------------------------------------------------ helpers
    # additive-attention aggregators: the logit is w . [query ; key (+ edge)] (+ b); query and bias cancel
    # inside the segment softmax, so all four reduce to "score of the key + edge gain"
    _HIP_AGGS = (K.NA_ATTN_H, K.NA_ATTN_X, K.NA_SELF_ATTN_H, K.NA_SELF_ATTN_X)

    def _hip_supported(self) -> bool:
        return self.agg in self._HIP_AGGS and not self.agg_x and bool(self.recurr)

    def _attn_geometry(self, i: int):
        """(offset of the key weights inside attn_lin.weight, key width) for stacked layer i."""
        key_dim = self.emb_dim if self.agg_attn_x else self.hidden_dim
        if "self_attn" in self.agg:
            return 0, key_dim                                   # SelfAttnConv: Linear(attn_dim, 1), no query
        attn_dim = self.emb_dim if self.agg_attn_x else self.hidden_dim
        return (self.emb_dim if i == 0 else attn_dim), key_dim  # AttnConv: Linear(attn_q_dim + attn_dim, 1)

    def _cells(self, fresh: bool = False):
        srcs: List[torch.Tensor] = []
        for d in self.dirs:
            for i in range(self.num_layers):
                c = getattr(self, "cells_%d" % d)[i]
                a = getattr(self, "node_aggr_%d" % d)[i]
                srcs += [c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh, a.attn_lin.weight]
                if a.wea:
                    srcs.append(a.edge_encoder.weight)

        def make():
            out = {}
            for d in self.dirs:
                for i in range(self.num_layers):
                    c = getattr(self, "cells_%d" % d)[i]
                    a = getattr(self, "node_aggr_%d" % d)[i]
                    dq, kd = self._attn_geometry(i)
                    out[(d, i)] = derive_cell(c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh, a.attn_lin.weight,
                                              self.hidden_dim, dq, i > 0, a.edge_encoder.weight if a.wea else None, 0,
                                              schedule=self.schedule, key_dim=kd, pack=False, stacked=self.num_layers)
            if self.schedule == "lockstep":
                pack_lockstep(out.values())
            return out

        return self._derived.setdefault(self.schedule, DerivedCache()).get(srcs, make, fresh=fresh or self.training)

    def _folded_tables(self, cells):
        """Input side of stacked layer 0 by constant folding (evaluation only).  The node embedding is a sum of three table
        rows (utils.py:26-28) and nothing non-linear sits between it and `GRUCell`'s `W_ih x + b_ih` (dagnn.py:139,148,181), so
        `W_ih x_v + b_ih = (T W_ih^T)[type_v] + (A W_ih^T)[attr_v] + (D W_ih^T + b_ih)[depth_v]`: the three products depend on
        the PARAMETERS only and are made once per weight version (kept on the derived cell, like the packed matrices); per batch the
        [N, emb] x [emb, 3H] GEMM of every direction becomes the encoder's own row kernel on the folded tables.  Per
        direction (type, attribute, depth) tables of width 3Hp, or None where it does not apply (other encoders, odd widths;
        training passes never ask)."""
        if not engine.FOLD_INPUT or type(self.encoder) is not ASTNodeEncoder or self.schedule != "lockstep" or self.agg_x:
            return None
        if self.training:
            # a no-grad pass on a module left in train() mode (validation without eval(), MC dropout): the derived cells are
            # rebuilt on every pass there, and re-folding three tables per direction each time costs what the fold saves
            return None
        enc = self.encoder
        tabs = [enc.type_encoder.weight, enc.attribute_encoder.weight, enc.depth_encoder.weight]
        if tabs[0].shape[1] % 4 or not tabs[0].is_cuda:
            return None
        key = tuple((p.data_ptr(), p._version) for p in tabs)
        out = []
        with torch.no_grad():
            for d in self.dirs:
                c = cells[(d, 0)]   # (the folded tables live and die with the derived cell: same invalidation rules, same width)
                if c.fold is None or c.fold[0] != key:
                    t, a, dp = (engine.gemm_nt_bias([tab.detach()], [c.w_ih], [b])[0]
                                for tab, b in zip(tabs, (None, None, c.b_ih)))   # (every node takes exactly one depth row: the bias rides on it)
                    c.fold = (key, (t, a, dp), built_marker(t))
                else:
                    meet_built(c.fold[2])   # (built by a pass on another stream, perhaps still in flight)
                out.append(c.fold[1])
        self.__dict__["fold_passes"] = self.__dict__.get("fold_passes", 0) + 1   # (tests assert the path a pass took)
        return out

    @staticmethod
    def _rows_ok(x_idx, depth) -> bool:
        return bool(x_idx.is_cuda and x_idx.dtype == torch.int64 and x_idx.dim() == 2 and x_idx.shape[1] == 2
                    and x_idx.is_contiguous() and depth.is_cuda and depth.dtype == torch.int64 and depth.is_contiguous())

    def _folded_gi0(self, x_idx, depth, cells):
        """gi0 of every direction from the folded tables (`_folded_tables`), or None."""
        if not self._rows_ok(x_idx, depth):
            return None
        folded = self._folded_tables(cells)
        if folded is None:
            return None
        return [engine.encode_ast(x_idx, depth, t, a, dp, self.encoder.max_depth) for (t, a, dp) in folded]

    def _prepare_fused(self, G, B, cells):
        """Evaluation passes: plan + dataflow schedule + encoder rows (+ folded gi0 rows) + side effect 1 as ONE pipeline of 7
        launches (`dagnn_prepare`, csrc/prepare.hip).  Returns (plan, gi0 or None) with `G.x` / `G.bi_layer_index` set, or None
        where the pipeline does not apply (the caller then takes the separate calls)."""
        if not engine.PREPARE_FUSED or self.schedule != "lockstep" or type(self.encoder) is not ASTNodeEncoder or \
                getattr(G, "_dagnn_plan", None) is not None:
            return None
        x_idx, depth = G.x, G.node_depth.view(-1, )
        enc = self.encoder
        if not (self._rows_ok(x_idx, depth) and enc.type_encoder.weight.shape[1] % 4 == 0 and G.edge_index.is_cuda):
            return None
        has_edge_enc = getattr(self.node_aggr_0[0], "wea", False)
        ea = G.edge_attr if has_edge_enc else None
        dev, N = x_idx.device, x_idx.shape[0]
        R = 0 if ea is None else int(ea.numel() // max(1, G.edge_index.shape[1]))
        wide_ok = has_edge_enc and not (self.agg_x or self.agg_attn_x)
        Hp = engine.state_width(self.hidden_dim, self.num_layers, R, wide_ok=wide_ok)
        groups = engine.dataflow_groups(dev, len(self.dirs), self.num_layers, Hp, B, training=False)
        tables = [(enc.type_encoder.weight, enc.attribute_encoder.weight, enc.depth_encoder.weight,
                   torch.empty(N, enc.type_encoder.weight.shape[1], dtype=torch.float32, device=dev))]
        folded = self._folded_tables(cells)
        gi0 = None
        if folded is not None:
            gi0 = [torch.empty(N, t.shape[1], dtype=torch.float32, device=dev) for (t, _, _) in folded]
            tables += [(t, a, dp, o) for (t, a, dp), o in zip(folded, gi0)]
        srcs = [G._bi_layer_idx0, G._bi_layer_index0, G._bi_layer_idx1, G._bi_layer_index1]
        lidx = torch.empty(4, N, dtype=torch.int64, device=dev)
        plan = engine.build_plan(G.edge_index, G._bi_layer_idx0, G._bi_layer_idx1, G.batch, B, ea, launch=False)
        plan.launch_prepare(groups, enc=(x_idx, depth, enc.max_depth, tables), stack=(srcs, lidx))
        G.bi_layer_index = lidx.view(2, 2, -1)   # side effect 1 (dagnn.py:130-133)
        G.x = tables[0][3]                       # side effects 2 + 3 (dagnn.py:139, utils.py:27)
        return plan, gi0

    def check(self) -> None:
        """Blocking check for device-side failures of every pass launched so far (`core.check_arenas`): call it where
        the outputs of the LAST forward of a loop are consumed - the non-blocking poll inside `forward` only reports
        earlier passes."""
        from .core import check_arenas
        check_arenas(self)

    def _arena_for(self, x, role="forward"):
        # one arena per stream: passes issued on different streams keep their own granule buffers, epochs and error words
        # (their plan / encoder / GEMM / head kernels may overlap; the all-resident persistent launches themselves are
        # ordered device-wide by engine.persistent_launch - two of them in flight would deadlock on each other's CUs)
        key = (role, x.device, engine._stream(x))
        arena = self._arenas.get(key)
        if arena is None:
            arena = engine.GranuleArena()
            # passes of further streams (micro-batches in flight) start their workgroup packing two XCDs further on
            arena.xcd_first = 2 * sum(1 for k in self._arenas if k[0] == role and k[1] == x.device) % 8
            self._arenas[key] = arena
        return arena

    def _training_pass(self) -> bool:
        """True when this call must be differentiable.  The HIP backward (csrc/backward.hip) covers what the
        reference's training scripts run (scripts/ogb_tok.sh: attn_h, bidirectional, max-pool over the output
        nodes); other combinations raise instead of silently detaching."""
        if not (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return False
        ok = self._hip_supported() and self.schedule == "lockstep" and self.emb_dim % 4 == 0
        if not ok:
            raise NotImplementedError(
                "the HIP backward pass covers the aggregators %s (any read-out) with the lock-step schedule; call "
                "this configuration under torch.no_grad() (evaluation) or freeze its parameters" % (self._HIP_AGGS,))
        return True

    # hooks of autograd.Recurrence
    _vid_nodes = 0

    def _key_offset(self, i: int) -> int:
        return self._attn_geometry(i)[0]

    def _static_scores(self, x, cells):
        """`*_x` aggregators: the keys are the node inputs, one score per node and cell (dagnn.py:175-177)."""
        if not self.agg_attn_x:
            return None
        return {k: torch.mv(x.detach(), c.key_raw) for k, c in cells.items()}

    def _readout(self, plan, B, x, h):
        """Max-pool over the output nodes of both directions (dagnn.py:184-193), columns [d][x?, layer 0.. L-1]."""
        out = torch.empty(B, self.out_hidden_dim, dtype=torch.float32, device=x.device)
        col, jobs = 0, []
        for d in (0, 1):
            for t in ([x] if self.out_wx else []) + [h[d][i] for i in range(self.num_layers)]:
                jobs.append((t, d, col))
                col += t.shape[1]
        if len(jobs) <= 16:
            engine.readout_max_batch(plan, jobs, out)   # one launch for all (direction, stacked layer) columns
        else:
            for t, d, c in jobs:
                engine.readout_max(plan, t, d, out, c)
        return out

    def _readout_backward(self, plan, x, h, gout, g_ext, dx):
        col, jobs = 0, []
        for d in (0, 1):  # to the arg-max output node of every (graph, column)
            if self.out_wx:   # (both directions add into dx: one after the other, not in one launch)
                engine.readout_max_backward(plan, x, d, gout, col, dx)
                col += x.shape[1]
            for i in range(self.num_layers):
                jobs.append((h[d][i], d, col, g_ext[d][i]))
                col += self.hidden_dim
        engine.readout_max_backward_batch(plan, jobs, gout)   # one launch for all (direction, stacked layer) state buffers

    def _train_params(self):
        flat = []
        for d in self.dirs:
            for i in range(self.num_layers):
                c = getattr(self, "cells_%d" % d)[i]
                a = getattr(self, "node_aggr_%d" % d)[i]
                flat += [c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh, a.attn_lin.weight, a.attn_lin.bias,
                         a.edge_encoder.weight if a.wea else None, a.edge_encoder.bias if a.wea else None]
        return flat

    def _pool(self, h, batch, B):
        """`global_{max,mean,add}_pool` / P_ATTN read-outs on torch (variants outside BASELINE)."""
        how = self.out_pool
        if how == K.P_ATTN:  # dagnn.py:114-117: softmax over a size-1 dim == 1 -> sum pooling
            w = F.softmax(self.self_attn_linear_out(h), dim=-1)
            h, how = w * h, K.P_ADD
        idx = batch.view(-1, 1).expand_as(h)
        out = h.new_zeros(B, h.shape[1])
        if how == K.P_MAX:
            return out.scatter_reduce_(0, idx, h, "amax", include_self=False)
        out.scatter_add_(0, idx, h)
        if how == K.P_MEAN:
            cnt = torch.bincount(batch, minlength=B).clamp(min=1).to(h.dtype).view(-1, 1)
            out = out / cnt
        return out

    def train(self, mode: bool = True):
        """Mode switches drop the derived-weight caches (core.DerivedCache: an optimizer may have updated the
        parameters without bumping their version counters)."""
        self.invalidate_caches()
        return super().train(mode)

    def _plan_of(self, G, B, overlap: bool = False):
        """The batch's plan.  With `overlap` the plan kernels (and the dataflow schedule's) run on a side stream next
        to the encoder and the batched input GEMM, which do not depend on them; the recurrence waits for `plan.ready`."""
        has_edge_enc = getattr(self.node_aggr_0[0], "wea", False)
        if getattr(G, "_dagnn_plan", None) is not None:  # built by the loader (dagnn_amd.host_plan.attach_plan)
            if has_edge_enc or int(G._dagnn_plan_meta.get("R", 0)) == 0:
                if overlap:
                    self._set_layer_index(G)
                return engine.PlanHandle.from_words(G._dagnn_plan, G._dagnn_plan_meta, getattr(G, "_dagnn_df", None))
            # the loader packed edge features this model has no encoder for (w_edge_attr=False): the kernels would
            # want an edge gain per feature - build the plan without them here instead
        ea = G.edge_attr if has_edge_enc else None
        if not (overlap and G.edge_index.is_cuda and self.schedule == "lockstep"):
            if overlap:
                self._set_layer_index(G)
            return engine.build_plan(G.edge_index, G._bi_layer_idx0, G._bi_layer_idx1, G.batch, B, ea)
        dev = G.edge_index.device
        # the same (width, group count) key `run_stack_lockstep` will ask for: hidden sizes 257..320 run 320 wide on the
        # dataflow kernel (`wide_ok`), a training pass under an active communicator reserves CUs for the collective
        R = 0 if ea is None else int(ea.numel() // max(1, G.edge_index.shape[1]))
        wide_ok = has_edge_enc and not (self.agg_x or self.agg_attn_x)
        Hp = engine.state_width(self.hidden_dim, self.num_layers, R, wide_ok=wide_ok)
        groups = engine.dataflow_groups(dev, len(self.dirs), self.num_layers, Hp, B, training=self._training_pass())
        # every buffer comes from the CALLER's pool (no `record_stream`, no foreign-pool blocks that cannot be reused: a forward
        # that allocated under a side stream cost seven hipMalloc calls per batch); only the launches may go to a side stream
        plan = engine.build_plan(G.edge_index, G._bi_layer_idx0, G._bi_layer_idx1, G.batch, B, ea, launch=False)
        if groups > 0:
            plan.dataflow_schedule(groups, launch=False)
        srcs = [G._bi_layer_idx0, G._bi_layer_index0, G._bi_layer_idx1, G._bi_layer_index1]
        stack = None
        if engine.PREPARE_FUSED and all(t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() for t in srcs):
            stack = (srcs, torch.empty(4, srcs[0].shape[0], dtype=torch.int64, device=dev))

        def launch():
            if engine.PREPARE_FUSED:   # plan + schedule (+ side effect 1 on its second launch) as one pipeline (csrc/prepare.hip)
                plan.launch_prepare(groups, stack=stack)
            else:
                plan.launch_build()
                if groups > 0:
                    plan.dataflow_schedule(groups)

        if engine.PLAN_OVERLAP:
            # ... on the arena's side stream, next to the encoder and the input GEMM of a training pass, which do not depend on
            # them; the side stream forks here and joins at `plan.wait_ready()`
            cur = torch.cuda.current_stream(dev)
            side = self._arena_for(G.edge_index).side_stream(dev)
            side.wait_stream(cur)
            with engine.launch_on(side):
                launch()
            plan.ready = torch.cuda.Event()
            plan.ready.record(side)
        else:
            launch()
        if stack is not None:
            G.bi_layer_index = stack[1].view(2, 2, -1)   # side effect 1 (dagnn.py:130-133); on the caller's stream after `plan.wait_ready()`
        else:
            self._set_layer_index(G)
        return plan

    @staticmethod
    def _set_layer_index(G, out=None):
        """side effect 1 (dagnn.py:130-133): `G.bi_layer_index` [2, 2, N] (one copy kernel instead of three)"""
        srcs = [G._bi_layer_idx0, G._bi_layer_index0, G._bi_layer_idx1, G._bi_layer_index1]
        G.bi_layer_index = (torch.stack(srcs, dim=0) if out is None else torch.stack(srcs, dim=0, out=out)).view(2, 2, -1)

    # ------------------------------------------------------------------------------ forward
    def forward(self, G):
        L, H, dirs = self.num_layers, self.hidden_dim, self.dirs
        if not self._hip_supported():
            # constructor strings outside every BASELINE configuration: same contract, torch-ROCm ops
            if not G.x.is_cuda:
                raise engine.DagnnHipError("DAGNN.forward needs its batch on a ROCm GPU (there is no CPU path)")
            from . import variants
            G.bi_layer_index = torch.stack([G._bi_layer_idx0, G._bi_layer_index0, G._bi_layer_idx1, G._bi_layer_index1],
                                        dim=0).view(2, 2, -1)   # (one copy kernel instead of three)
            B = num_graphs_of(G)
            G.x = self.encoder(G.x, G.node_depth.view(-1, ))
            if self.variant_backend == "torch" or (torch.is_grad_enabled()
                                                    and any(p.requires_grad for p in self.parameters())):
                for c in (self._head_cache, self.__dict__.get("_variant_cache"), self.__dict__.get("_plain_df_cache")):
                    if c is not None:
                        c.invalidate()
                if self.variant_backend != "torch" and variants.hip_backward_supported(self, G):
                    # gated_sum / mattn_h / add with GRU cells: forward AND reverse sweep in HIP (csrc/variants_bwd.hip)
                    plan = self._plan_of(G, B)
                    flat_params = [p for d in self.dirs for i in range(L) for _, p in variants._cell_params(self, d, i)]
                    flat = variants.VariantRecurrence.apply(self, G, plan, G.x, *flat_params)
                    h = [[None] * L for _ in range(2)]
                    for q, d in enumerate(dirs):
                        for i in range(L):
                            h[d][i] = flat[q * L + i]
                    return self._finish(G, None, G.x, h, B)
                if self.variant_backend != "torch":   # (an explicit 'torch' backend is a choice, not a cliff)
                    variants.warn_torch_path(self, G)
                return self._finish(G, None, G.x, variants.run(self, G, G.x), B)   # training: differentiable torch ops
            plan = self._plan_of(G, B)
            return self._finish(G, plan, G.x, variants.run_hip(self, G, G.x, plan), B)
        train = self._training_pass()

        B = num_graphs_of(G)
        if not train:
            cells = self._cells()
            fused = self._prepare_fused(G, B, cells)   # (plan + schedule + encoder + side effect 1: one pipeline, csrc/prepare.hip)
            if fused is not None:
                plan, gi0 = fused
                x = G.x
                sscore = self._static_scores(x, cells)
                h = run_stack(plan, x, cells, dirs, L, H, schedule=self.schedule, static_score=sscore,
                              arena=self._arena_for(x), gi0=gi0)
                self._guard_params(x)
                return self._finish(G, plan, x, h, B)
        # side effect 1 (dagnn.py:130-133) + the plan: on a side stream next to the encoder and the input GEMM, which do not
        # depend on them (`engine.PLAN_OVERLAP`); the caller's stream meets it again in front of the recurrence
        plan = self._plan_of(G, B, overlap=True)
        # side effects 2+3 (dagnn.py:139, utils.py:27): embedding replaces G.x, depth clamped in place
        x_idx, depth = G.x, G.node_depth.view(-1, )
        G.x = self.encoder(x_idx, depth)
        x = G.x
        fused_readout = self.bidirectional and not self.output_all and self.out_pool == K.P_MAX
        if train:
            self._head_cache.invalidate()   # the optimizer step that follows may not bump version counters
            # differentiable call: HIP read-out + its backward for the configuration the reference trains
            # (scripts/ogb_tok.sh), otherwise differentiable states and the torch read-outs below
            from .autograd import Recurrence
            res = Recurrence.apply(self, plan, B, fused_readout, x, *self._train_params())
            flat = res[1:] if fused_readout else res
            h = [[None] * L for _ in range(2)]
            for q, d in enumerate(dirs):
                for i in range(L):
                    h[d][i] = flat[q * L + i]
            if fused_readout:
                G.h = [[h[d][i] for i in range(L)] for d in dirs]
                return self._heads(self.dropout(res[0]))
            return self._finish(G, plan, x, h, B)
        cells = self._cells()
        sscore = self._static_scores(x, cells)
        h = run_stack(plan, x, cells, dirs, L, H, schedule=self.schedule, static_score=sscore,
                      arena=self._arena_for(x), gi0=self._folded_gi0(x_idx, depth, cells))
        self._guard_params(x)
        return self._finish(G, plan, x, h, B)

    def _guard_params(self, x) -> None:
        """Evaluation passes: the parameters behind the derived-weight caches still are what the caches were built from
        (`core.ParamGuard`; a training-mode pass rebuilds everything anyway)."""
        if self.training or not engine.PARAM_GUARD:
            return
        from .core import guard_params
        guard_params(self, self._arena_for(x).err)

    def invalidate_caches(self) -> None:
        """Drop every tensor derived from the parameters (what `train()` / `eval()` do): call it after updating parameters in
        evaluation mode through a path the version counters do not see (`.data`, a fused optimizer)."""
        for c in list(self.__dict__.get("_derived", {}).values()) + [self.__dict__.get("_head_cache"),
                                                                      self.__dict__.get("_variant_cache"),
                                                                      self.__dict__.get("_plain_df_cache")]:
            if c is not None:
                c.invalidate()
        from .core import drop_guard
        drop_guard(self)

    def _head_storage(self):
        """The S vocabulary heads' weights and biases as ONE [S V, D] / [S V] pair: each head's parameter is (made) a VIEW of
        it - same names, shapes and values in `state_dict`, same `Parameter` objects for the optimizer - so the heads are one
        matrix for the library GEMMs without a concatenation per pass (102 MB at the headline shape) and without a cache that
        could go stale: an in-place update of a head IS an update of the matrix.  Re-established when something re-seated the
        parameters' storage (`module.to()`, `deepcopy`)."""
        heads = list(self.graph_pred_linear_list)
        V, D = heads[0].weight.shape
        st = self.__dict__.get("_heads_flat")
        w0 = heads[0].weight
        ok = st is not None and st[0].device == w0.device and st[0].dtype == w0.dtype
        if ok:
            for i, hd in enumerate(heads):
                if hd.weight.data_ptr() != st[0].data_ptr() + i * V * D * st[0].element_size() or \
                        hd.bias.data_ptr() != st[1].data_ptr() + i * V * st[1].element_size():
                    ok = False
                    break
        if not ok:
            with torch.no_grad():
                w = torch.cat([hd.weight.data for hd in heads], 0)
                b = torch.cat([hd.bias.data for hd in heads], 0)
                for i, hd in enumerate(heads):
                    hd.weight.data = w[i * V:(i + 1) * V]
                    hd.bias.data = b[i * V:(i + 1) * V]
            st = self.__dict__["_heads_flat"] = (w, b)
        return st

    def _heads(self, out):
        """The prediction heads on the pooled graph vectors (dagnn.py:204-215)."""
        if self.num_class > 0:
            return self.graph_pred_linear(out)
        if self.num_vocab > 1 and self.max_seq_len > 1 and out.is_cuda and out.dtype == torch.float32 and \
                all(isinstance(hd, nn.Linear) for hd in self.graph_pred_linear_list):
            # the S vocabulary heads as ONE library GEMM (dagnn.py:212-215 runs S): their parameters are views of one matrix
            # (`_head_storage`), the list entries are views of its output.  Under autograd the GEMM and its two backward
            # products are one node (`_HeadsLinear`); a caller that takes the loss through `train.seq_cross_entropy` meets
            # the logits as ONE tensor there too (fifteen 25-us GEMMs + ~50 loss launches -> 3 + 2).
            wcat, bcat = self._head_storage()
            if torch.is_grad_enabled() and (out.requires_grad or any(p.requires_grad for p in self.graph_pred_linear_list.parameters())):
                heads = list(self.graph_pred_linear_list)
                logits = _HeadsLinear.apply(out, wcat, bcat, self.num_vocab, *[hd.weight for hd in heads], *[hd.bias for hd in heads])
            else:
                logits = torch.addmm(bcat, out, wcat.t())
            return list(logits.split(self.num_vocab, dim=1))
        return [self.graph_pred_linear_list[i](out) for i in range(self.max_seq_len)]

    def _finish(self, G, plan, x, h, B):
        """Read-out + heads (dagnn.py:184-215) on the states h[d][i]."""
        L, dirs = self.num_layers, self.dirs
        G.h = [[h[d][i] for i in range(L)] for d in dirs]  # side effect 4 (dagnn.py:141-142,182)

        differentiable = torch.is_grad_enabled() and (x.requires_grad or any(h[d][i].requires_grad
                                                                             for d in dirs for i in range(L)))
        hip_pool = plan is not None and not differentiable
        # P_ATTN (dagnn.py:114-117) is a softmax over a size-1 dimension: weights of exactly 1, i.e. add-pooling
        how = K.P_ADD if self.out_pool in (K.P_ATTN, K.P_SUM) else self.out_pool
        if self.bidirectional and not self.output_all:
            if hip_pool and how == K.P_MAX:
                out = self._readout(plan, B, x, h)   # HIP max-pool over the output nodes
            elif hip_pool:
                out = torch.empty(B, self.out_hidden_dim, dtype=torch.float32, device=x.device)
                col = 0
                for d in (0, 1):
                    for t in ([x] if self.out_wx else []) + [h[d][i] for i in range(L)]:
                        engine.readout_pool(plan, t, d, how, out, col)
                        col += t.shape[1]
            else:
                outs = []
                for d in (0, 1):
                    idx = G.bi_layer_index[1 - d][1][G.bi_layer_index[1 - d][0] == 0]
                    hd = torch.cat(([x] if self.out_wx else []) + [h[d][i] for i in range(L)], dim=-1)
                    outs.append(self._pool(hd[idx], G.batch[idx], B))
                out = torch.cat(outs, dim=-1)
        else:
            G.h = torch.cat(([x] if self.out_wx else []) + [h[d][i] for d in dirs for i in range(L)], dim=-1)
            if hip_pool:   # over all nodes, or over the output nodes of direction 0 (dagnn.py:124-126,194-202)
                out = torch.empty(B, G.h.shape[1], dtype=torch.float32, device=x.device)
                engine.readout_pool(plan, G.h, 2 if self.output_all else 0, how, out, 0)
            if not self.output_all:
                idx = G.bi_layer_index[1][1][G.bi_layer_index[1][0] == 0]  # dagnn.py:124-126
                G.h, G.batch = G.h[idx], G.batch[idx]
            if not hip_pool:
                out = self._pool(G.h, G.batch, B)

        return self._heads(self.dropout(out))
