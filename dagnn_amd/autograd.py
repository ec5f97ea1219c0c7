"""Training path: `torch.autograd.Function`s around the HIP kernels.

The reference trains through torch autograd over its Python loops (`loss.backward()`,
ogbg-code/main_pyg.py:62): one autograd node per (direction, topological layer, stacked layer)
micro-step.  Here the forward pass is the same HIP path as inference and keeps only the states
h[d][i]; `Recurrence.backward` recomputes aggregates / pre-activations in parallel, walks the dependent
chain once in reverse lock-step (csrc/backward.hip) and finishes with library GEMMs for the weight
gradients.  Everything around it (dropout, the vocabulary heads, the loss, the optimizer) stays
ordinary torch autograd, as in the reference.
"""
from __future__ import annotations

import torch

from . import engine
from .core import run_stack_lockstep


class EncodeAST(torch.autograd.Function):
    """`ASTNodeEncoder.forward` (ogbg-code/utils.py:26-28) on the HIP kernel; backward = the three
    embedding-table gradients (row sums of the incoming gradient by index)."""

    @staticmethod
    def forward(ctx, x, depth, type_w, attr_w, depth_w, max_depth):
        out = engine.encode_ast(x, depth, type_w, attr_w, depth_w, max_depth)  # clamps `depth` in place
        ctx.save_for_backward(x, depth.clone())
        ctx.rows = (type_w.shape[0], attr_w.shape[0], depth_w.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        x, depth = ctx.saved_tensors
        g = g.contiguous()
        grads = []
        for rows, idx in zip(ctx.rows, (x[:, 0], x[:, 1], depth)):
            grads.append(torch.zeros(rows, g.shape[1], dtype=g.dtype, device=g.device).index_add_(0, idx, g))
        return (None, None, grads[0], grads[1], grads[2], None)


def _wgrad(dg: torch.Tensor, u: torch.Tensor, splits: int = 32) -> torch.Tensor:
    """dg^T @ u for [N, J] x [N, K] with N >> J, K (a weight gradient): the reduction dimension is the long one,
    so it is split into `splits` batches (one library bmm fills the GPU; a plain GEMM here has J*K/128^2 = 12
    output tiles for 256 CUs) and the partial products are summed in a fixed order."""
    N = dg.shape[0]
    n = N // splits * splits
    if n == 0:
        return dg.t() @ u
    out = torch.bmm(dg[:n].view(splits, n // splits, dg.shape[1]).transpose(1, 2),
                    u[:n].reshape(splits, n // splits, u.shape[1])).sum(0)
    if n < N:
        out = out + dg[n:].t() @ u[n:]
    return out


class Recurrence(torch.autograd.Function):
    """Plan + node inputs -> graph read-out, differentiable (dagnn.py:144-193; dvae/dagnn.py:99-175).

    `mod` is the calling module; it supplies `num_layers`, `hidden_dim`, `dirs`, `_cells()`, `_arena_for(x, role)`,
    `_vid_nodes` (node count per graph when the keys carry a vertex-id one-hot, else 0), `_key_offset(i)` (position of
    the key weights inside attn_lin.weight of stacked layer i) and the two read-out hooks
    `_readout(plan, B, x, h) -> out`, `_readout_backward(plan, x, h, grad_out, g_ext, dx)`.
    Inputs after `x` are the cells' parameters, 8 per (direction, stacked layer): weight_ih, weight_hh, bias_ih,
    bias_hh, attn_lin.weight, attn_lin.bias, edge_encoder.weight, edge_encoder.bias (the last two None without edge
    features)."""

    PER_CELL = 8

    @staticmethod
    def forward(ctx, mod, plan, B, fused, x, *params):
        """fused: returns (read-out [B, .], states...) with the read-out done by the module's HIP hooks and the states
        not differentiable; otherwise returns the states h[d][i] ([N, H] each, d over `dirs`, i over layers) as
        differentiable outputs - any torch read-out can follow (other pools, all nodes, unidirectional)."""
        L, H, dirs = mod.num_layers, mod.hidden_dim, mod.dirs
        ctx.set_materialize_grads(False)   # (outputs nobody differentiates arrive as None in backward, not as [N, H] zero fills)
        cells = mod._cells(fresh=True)   # a differentiable pass: the optimizer changes the parameters every step
        keep = {}
        sscore = mod._static_scores(x, cells)   # keys from the inputs (`*_x` aggregators): one score per node and cell
        h = run_stack_lockstep(plan, x, cells, dirs, L, H, vid_nodes=mod._vid_nodes, arena=mod._arena_for(x), keep=keep,
                               static_score=sscore)
        ctx.mod, ctx.plan, ctx.cells, ctx.keep, ctx.h, ctx.fused = mod, plan, cells, keep, h, bool(fused)
        ctx.sscore = sscore
        ctx.save_for_backward(x, *[p for p in params if p is not None])
        ctx.present = [p is not None for p in params]
        flat = [h[d][i] for d in dirs for i in range(L)]
        if not fused:
            return tuple(flat)
        out = mod._readout(plan, B, x, h)
        ctx.mark_non_differentiable(*flat)
        return (out,) + tuple(flat)

    @staticmethod
    def backward(ctx, *gouts):
        mod, plan, cells, keep, h = ctx.mod, ctx.plan, ctx.cells, ctx.keep, ctx.h
        saved = list(ctx.saved_tensors)
        x, it = saved[0], iter(saved[1:])
        params = [next(it) if present else None for present in ctx.present]
        L, H, dirs, Hp = mod.num_layers, mod.hidden_dim, mod.dirs, keep["Hp"]
        N, dev = x.shape[0], x.device
        g_all = torch.zeros(len(dirs) * L, N, Hp, dtype=torch.float32, device=dev)   # (one fill instead of one per cell)
        g_ext = [[g_all[dirs.index(d) * L + i] if d in dirs else None for i in range(L)] for d in range(2)]
        dx = torch.zeros_like(x)
        if ctx.fused:
            if gouts[0] is not None:
                mod._readout_backward(plan, x, h, gouts[0].contiguous().float(), g_ext, dx)
        else:   # gradients of the states themselves, from whatever torch read-out followed
            for q, d in enumerate(dirs):
                for i in range(L):
                    if gouts[q * L + i] is not None:
                        g_ext[d][i][:, :H] = gouts[q * L + i]
        groups = keep.get("groups", 0)
        if groups > 0 and engine.bwd_dataflow_groups(dev, len(dirs), L, Hp, plan.B) == groups and \
                engine.bwd_dataflow_fits(dev, N, len(dirs) * L):
            res = engine.bwd_dataflow_sweep(plan, dirs, L, Hp, cells, keep["h_buf"], keep["gi0"], g_ext, groups,
                                            arena=mod._arena_for(x, "backward"), vid_mod=mod._vid_nodes,
                                            static_score=ctx.sscore, preact=keep.get("preact"))
        else:
            res = engine.backward_sweep(plan, dirs, L, Hp, cells, keep["h_buf"], keep["gi0"], g_ext,
                                        arena=mod._arena_for(x, "backward"), vid_mod=mod._vid_nodes,
                                        static_score=ctx.sscore)

        ep = engine._span("backward_epilogue", x)
        ep.__enter__()

        def gates(t):  # [N, 3Hp] in gate blocks of Hp -> [N, 3H]
            return t if Hp == H else t.view(N, 3, Hp)[:, :, :H].reshape(N, 3 * H)

        # weight / bias gradients of every cell: ONE batch of split transposed products in HIP (csrc/wgrad.hip)
        jobs = []
        for d in dirs:
            for i in range(L):
                r = res[(d, i)]
                jobs.append((r["dgi"], x if i == 0 else h[d][i - 1], True))
                jobs.append((r["dgh"], r["a"][:, :H], True))
        wg = engine.wgrad(jobs, N, Hp, H) if N > 0 else None
        # the small reductions of the attention / edge-encoder gradients, all cells in one batch (csrc/wgrad.hip):
        # sum_v sigma_v keys_v, sum_v (edge-feature sums)_v, sum_v sigma_v
        cs_jobs, cs_at = [], {}
        if N > 0:
            for d in dirs:
                for i in range(L):
                    r = res[(d, i)]
                    keys = x if ctx.sscore is not None else h[d][i]
                    cs_at[(d, i)] = len(cs_jobs)
                    cs_jobs.append((keys, r["sigma"]))
                    if r["edge_feat_grad"] is not None:
                        cs_jobs += [(r["edge_feat_grad"], None), (r["sigma"], None)]
        cs = engine.colsums(cs_jobs, N) if cs_jobs else None
        # gradients of attn_lin.weight and of the edge encoder, every cell in ONE launch (csrc/wgrad.hip).  The attention logit is
        # s_e = w_key . (h_p + W_e feat_e + b_e) [+ w_vid[p mod n]] (+ query and bias terms that cancel in the segment
        # softmax: their gradients are exact zeros)
        ag = None
        if cs is not None:
            ag_jobs = []
            for k, (d, i) in enumerate([(d, i) for d in dirs for i in range(L)]):
                w_ih, w_hh, b_ih, b_hh, attn_w, attn_b, edge_w, edge_b = params[k * Recurrence.PER_CELL:(k + 1) * Recurrence.PER_CELL]
                at = cs_at[(d, i)]
                has_e = edge_w is not None and res[(d, i)]["edge_feat_grad"] is not None
                ag_jobs.append({"key_sum": cs[at], "feat_sum": cs[at + 1] if has_e else None,
                                "sigma_sum": cs[at + 2] if has_e else None, "edge_w": edge_w if has_e else None, "edge_b": edge_b,
                                "attn_w": attn_w, "dq": mod._key_offset(i)})
            ag = engine.attn_grads(ag_jobs)
        grads = []
        k = 0
        for d in dirs:
            for i in range(L):
                w_ih, w_hh, b_ih, b_hh, attn_w, attn_b, edge_w, edge_b = params[k:k + Recurrence.PER_CELL]
                r = res[(d, i)]
                ci = k // Recurrence.PER_CELL
                if wg is not None:
                    (g_wih, g_bih), (g_whh, g_bhh) = wg[2 * ci], wg[2 * ci + 1]
                else:
                    g_wih, g_bih, g_whh, g_bhh = (torch.zeros_like(t) for t in (w_ih, b_ih, w_hh, b_hh))
                k += Recurrence.PER_CELL
                if i == 0 and x.requires_grad:
                    dx.addmm_(gates(r["dgi"]), w_ih)   # (dx is this call's own buffer: accumulate in place, no add kernel)
                dq = mod._key_offset(i)
                sigma = r["sigma"]
                kd = attn_w.shape[1] - dq - mod._vid_nodes     # key width: H, or the input width for the `*_x` aggregators
                if ctx.sscore is not None and x.requires_grad:   # the score of node v is w_key . x_v
                    dx = dx + sigma[:, None] * attn_w[0, dq:dq + kd][None, :]
                if ag is not None:
                    g_attn, g_edge_w, g_edge_b = ag[ci]
                    if edge_w is not None and g_edge_w is None:
                        g_edge_w, g_edge_b = torch.zeros_like(edge_w), torch.zeros_like(edge_b)
                else:   # an empty batch
                    g_attn = torch.zeros_like(attn_w)
                    g_edge_w = None if edge_w is None else torch.zeros_like(edge_w)
                    g_edge_b = None if edge_w is None else torch.zeros_like(edge_b)
                if mod._vid_nodes:   # node v carries the one-hot of (v mod n): d w_vid[j] = sum of sigma over those nodes
                    g_attn[0, dq + kd:dq + kd + mod._vid_nodes] = sigma.view(-1, mod._vid_nodes).sum(0)
                grads += [g_wih, g_whh, g_bih, g_bhh, g_attn, None if attn_b is None else torch.zeros_like(attn_b),
                          g_edge_w, g_edge_b]
        ep.__exit__(None, None, None)
        return (None, None, None, None, dx if x.requires_grad else None) + tuple(grads)
