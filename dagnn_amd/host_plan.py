"""Loader-side plan: the layer-sorted CSR + lock-step schedule of a batch, built on the host.

`dagnn_plan_build` (csrc/plan.hip) derives the plan on the device inside every `forward`, and the
lock-step driver then reads the layer offsets back (one device->host sync per batch, the counterpart of
the reference's `.item()` at `ogbg-code/model/dagnn.py:137`).  Both are pure functions of the batch's
integer arrays, so a `DataLoader` worker can do them while the GPU is busy with the previous batch -
where the reference's `Collater` (`ogbg-code/tg/dataloader.py:13-35`) already runs:

    batch = collate_with_plan(data_list)        # in the worker: GraphBatch + int32 plan + host schedule
    batch = batch.to(device)                    # the plan travels as one int32 tensor
    model(batch)                                # no plan kernels, no device->host sync

`build_plan_host` writes exactly the words the device kernels write (same `PlanLayout`; the GPU test
compares the two word for word), with vectorised numpy - no Python loop over nodes or edges.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

PLAN_MAGIC = 0x44414731  # "DAG1", csrc/common.h
THIN_ROWS = 14            # DAGNN_PLAN_THIN_ROWS, csrc/common.h


def _align4(w: int) -> int:
    return (w + 3) & ~3


def plan_layout(N: int, E: int, B: int, R: int) -> Dict[str, int]:
    """Word offsets of the plan arrays: mirror of `dagnn_plan_layout_words` (csrc/common.h)."""
    o = 16
    L: Dict[str, int] = {}

    def take(name: str, n: int) -> None:
        nonlocal o
        L[name] = o
        o = _align4(o + n)

    take("node_ptr", B + 1)
    take("edge_ptr", B + 1)
    for name, n in (("depth", B), ("order", N), ("lstart", N + B), ("rowptr", N + B), ("col", E),
                    ("eattr", E * max(R, 0))):
        for d in (0, 1):
            take("%s%d" % (name, d), n)
    take("items", 2 * B)
    for name, n in (("slot", N), ("cursor", N + B), ("eidx", E), ("blptr", N + 2), ("blsplit", N + 2),
                    ("lbase", N + B), ("rowrec", 16 * N), ("brec", 16 * N)):
        for d in (0, 1):
            take("%s%d" % (name, d), n)
    L["total"] = o
    return L


def _np(t) -> np.ndarray:
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def build_plan_host(edge_index, layer_fwd, layer_bwd, batch, num_graphs: int, edge_attr=None,
                    return_written: bool = False):
    """-> (plan words int32 [total], [layer offsets of direction 0, of direction 1], [first deep slot of every
    layer of direction 0, of direction 1]) - the last two as int32 arrays, what `PlanHandle.read_schedule` reads back.

    Raises ValueError where the device build would flag `status` (unsorted batch vector, edges not grouped
    by graph or crossing graphs).  With `return_written` a bool mask of the words this function defines is
    returned as a fourth value (the device leaves the others uninitialised)."""
    ei = _np(edge_index).astype(np.int64).reshape(2, -1)
    batch = _np(batch).astype(np.int64).reshape(-1)
    layers = [_np(layer_fwd).astype(np.int64).reshape(-1), _np(layer_bwd).astype(np.int64).reshape(-1)]
    N, E, B = batch.shape[0], ei.shape[1], int(num_graphs)
    R = 0
    ea = None
    if edge_attr is not None:
        ea = np.ascontiguousarray(_np(edge_attr), dtype=np.float32).reshape(E, -1)
        R = ea.shape[1]
    if N and (np.any(np.diff(batch) < 0) or batch[0] < 0 or batch[-1] >= B):
        raise ValueError("plan contract violated: batch vector not sorted / out of range")
    if E:
        if ei.min() < 0 or ei.max() >= N or np.any(batch[ei[0]] != batch[ei[1]]):
            raise ValueError("plan contract violated: edge crosses graphs / out of range")
        if np.any(np.diff(batch[ei[0]]) < 0):
            raise ValueError("plan contract violated: edges not grouped by graph")
    L = plan_layout(N, E, B, R)
    ws = np.zeros(L["total"], dtype=np.int32)
    written = np.zeros(L["total"], dtype=bool) if return_written else None

    def put(name: str, values, index=None) -> None:
        values = np.asarray(values)
        if index is None:
            index = np.arange(values.shape[0])
        ws[L[name] + index] = values.astype(np.int32, copy=False) if values.dtype != np.int32 else values
        if written is not None:
            written[L[name] + index] = True

    ws[0:5] = (N, E, B, R, PLAN_MAGIC)
    if written is not None:
        written[0:5] = True
    gids = np.arange(B + 1)
    node_ptr = np.searchsorted(batch, gids, side="left")
    edge_ptr = np.searchsorted(batch[ei[0]], gids, side="left") if E else np.zeros(B + 1, dtype=np.int64)
    put("node_ptr", node_ptr)
    put("edge_ptr", edge_ptr)
    n_of = np.diff(node_ptr)
    ids = np.arange(N)
    sched: List[np.ndarray] = []
    splits: List[np.ndarray] = []
    depths = []
    for d in (0, 1):
        layer = np.clip(layers[d], 0, np.maximum(n_of[batch] - 1, 0)) if N else layers[d]
        depth = np.zeros(B, dtype=np.int64)
        if N:
            np.maximum.at(depth, batch, layer + 1)
        depths.append(depth)
        put("depth%d" % d, depth)
        # nodes sorted by (graph, layer, id): every frontier of every graph is a contiguous range
        order = np.lexsort((ids, layer, batch))
        pos = np.empty(N, dtype=np.int64)
        pos[order] = ids
        put("order%d" % d, order)
        # lstart: per graph (block at node_ptr[g] + g), entries 0..depth[g]: absolute positions into order
        big = int(n_of.max()) + 2 if B else 2
        key_sorted = batch[order] * big + layer[order]
        g_rep = np.repeat(np.arange(B), depth + 1)
        t_rep = np.arange(g_rep.shape[0]) - np.repeat(np.cumsum(depth + 1) - (depth + 1), depth + 1)
        ls_idx = node_ptr[g_rep] + g_rep + t_rep
        ls_val = np.searchsorted(key_sorted, g_rep * big + t_rep, side="left")
        put("lstart%d" % d, ls_val, ls_idx)
        # CSR rows: the node an edge feeds is its target (d = 0) or its source (d = 1); original edge order kept
        feed, other = (ei[1], ei[0]) if d == 0 else (ei[0], ei[1])
        p_e = pos[feed]
        perm = np.argsort(p_e, kind="stable")
        col = other[perm]
        put("col%d" % d, col)
        put("eidx%d" % d, perm)
        if R:
            put("eattr%d" % d, ea[perm].reshape(-1).view(np.int32))
        p_sorted = p_e[perm]
        row_begin = np.searchsorted(p_sorted, ids, side="left")   # indexed by sorted position
        g_of_pos = batch[order]
        rp = np.zeros(N + B, dtype=np.int64)
        rp[ids + g_of_pos] = row_begin
        rp[node_ptr[1:] + np.arange(B)] = edge_ptr[1:]
        put("rowptr%d" % d, rp)
        # batch-level layers: offsets, T at index N + 1
        T = int(depth.max()) if B else 0
        width = np.bincount(layer, minlength=T) if N else np.zeros(0, dtype=np.int64)
        bl = np.zeros(N + 2, dtype=np.int64)
        bl[1:T + 1] = np.cumsum(width[:T])
        bl[N + 1] = T
        put("blptr%d" % d, bl)
        sched.append(bl[:T + 1].astype(np.int32))
        # deep graphs of this direction: deeper than thr = 1 + the last layer with more than THIN_ROWS rows
        fat = np.flatnonzero(width[:T] > THIN_ROWS)
        thr = int(fat[-1]) + 1 if fat.size else 0
        ws[5 + d] = thr
        if written is not None:
            written[5 + d] = True
        deep = depth > thr
        # slots: nodes ordered by (layer, shallow before deep, graph, id); lbase = first slot of every (graph, layer)
        by_layer = np.lexsort((ids, deep[batch], layer)) if N else ids
        slot = np.empty(N, dtype=np.int64)
        slot[by_layer] = ids
        put("slot%d" % d, slot)
        split = np.zeros(N + 2, dtype=np.int64)
        if N:
            split[:T] = bl[:T] + np.bincount(layer[~deep[batch]], minlength=T)[:T]
        put("blsplit%d" % d, split)
        splits.append(split[:T].astype(np.int32))
        g2 = np.repeat(np.arange(B), depth)
        t2 = np.arange(g2.shape[0]) - np.repeat(np.cumsum(depth) - depth, depth)
        base2 = node_ptr[g2] + g2 + t2
        ls_full = np.zeros(N + B + 1, dtype=np.int64)
        ls_full[ls_idx] = ls_val
        cnt2 = ls_full[base2 + 1] - ls_full[base2]
        by_tg = np.lexsort((g2, deep[g2], t2))
        c = cnt2[by_tg]
        lb = np.empty(g2.shape[0], dtype=np.int64)
        lb[by_tg] = np.cumsum(c) - c
        put("lbase%d" % d, lb, base2)
        # 64-byte row records in slot order
        rec = np.zeros((N, 16), dtype=np.int32)
        eb = row_begin[pos]
        ee = np.empty(N, dtype=np.int64)
        ee_sorted = rp[ids + g_of_pos + 1]
        ee[order] = ee_sorted
        rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3] = ids, eb, ee, batch
        eattr_bits = ea[perm].view(np.int32) if R else None
        for q in range(4):
            ok = eb + q < ee
            src = np.minimum(eb + q, max(E - 1, 0))
            if E:
                rec[:, 4 + q] = np.where(ok, col[src], 0)
                if R >= 1:
                    rec[:, 8 + 2 * q] = np.where(ok, eattr_bits[src, 0], 0)
                if R >= 2:
                    rec[:, 9 + 2 * q] = np.where(ok, eattr_bits[src, 1], 0)
        out = np.empty((N, 16), dtype=np.int32)
        out[slot] = rec
        put("rowrec%d" % d, out.reshape(-1))
    # work items (g * 2 + d) sorted deepest first, ties by index
    key = np.stack(depths, 1).reshape(-1) if B else np.zeros(0, dtype=np.int64)
    put("items", np.argsort(-key, kind="stable"))
    if return_written:
        return ws, sched, splits, written
    return ws, sched, splits


DF_MAGIC = 0x44463031   # "DF01", csrc/dataflow.hip
DF_RB = 4               # rows per block


def dataflow_layout(N: int, B: int, G: int) -> Dict[str, int]:
    """Word offsets of the dataflow schedule workspace: mirror of `df_layout_words` (csrc/dataflow.hip)."""
    o = 16
    L: Dict[str, int] = {}

    def take(name: str, n: int) -> None:
        nonlocal o
        L[name] = o
        o = _align4(o + n)

    take("grp_of", B)
    take("gdepth", G)
    take("gload", G)
    take("loff", G + 1)
    for name, n in (("gtab", 2 * G), ("lcnt", N + G + 1), ("glbase", N + B), ("grec", 16 * (4 * N + 4))):
        for d in (0, 1):
            take("%s%d" % (name, d), n)
    L["total"] = o
    return L


def build_dataflow_schedule_host(plan_words, N: int, E: int, B: int, R: int, groups: int, cost_layer: int = 4,
                                 cost_row: int = 1) -> np.ndarray:
    """The dataflow kernel's schedule (`dagnn_dataflow_schedule`, csrc/dataflow.hip) from a plan, on the host: graphs
    dealt to `groups` groups longest-processing-time first (integer costs, ties to the lowest group), row records
    re-sorted by (group, layer, graph, node) with every group-layer padded to whole blocks of 4.  Word for word what
    the device kernels write (the GPU test compares them)."""
    ws = _np(plan_words).astype(np.int32, copy=False)
    P = plan_layout(N, E, B, R)
    S = dataflow_layout(N, B, groups)
    out = np.zeros(S["total"], dtype=np.int32)
    out[S["grec0"]:] = -1
    if B == 0 or N == 0:
        return out
    G = int(groups)
    node_ptr = ws[P["node_ptr"]:P["node_ptr"] + B + 1].astype(np.int64)
    n_of = np.diff(node_ptr)
    depth = [ws[P["depth%d" % d]:P["depth%d" % d] + B].astype(np.int64) for d in (0, 1)]
    items = ws[P["items"]:P["items"] + 2 * B]
    # ---- LPT assignment (df_assign_kernel)
    load = np.zeros(G, dtype=np.int64)
    gdepth = np.zeros(G, dtype=np.int64)
    empty = np.ones(G, dtype=bool)
    grp = np.zeros(B, dtype=np.int64)
    order = [int(it) >> 1 for it in items if not (it & 1)]   # graphs, deepest first (direction-0 entries of `items`)
    if B <= 4096 and len(order) > 0 and int(n_of.min()) == int(n_of.max()):
        # every graph has the same node count (the D-VAE batches): dealt round-robin in depth order, no sequential chain
        for j, g in enumerate(order):
            grp[g] = j % G
        for k in range(min(G, len(order))):
            dg = int(max(depth[0][order[k]], depth[1][order[k]]))
            gdepth[k] = dg
            load[k] = cost_layer * dg + cost_row * int(n_of[0]) * ((len(order) - k + G - 1) // G)
        order = []
    for g in order:
        dg = int(max(depth[0][g], depth[1][g]))
        cand = load + cost_row * int(n_of[g]) + np.where(empty, cost_layer * dg, 0)
        k = int(np.argmin(cand))   # first minimum = lowest group
        load[k] = cand[k]
        if empty[k]:
            gdepth[k], empty[k] = dg, False
        grp[g] = k
    out[0:3] = (G, DF_MAGIC, DF_RB)
    out[S["grp_of"]:S["grp_of"] + B] = grp
    out[S["gdepth"]:S["gdepth"] + G] = gdepth
    out[S["gload"]:S["gload"] + G] = np.minimum(load, 0x7fffffff)
    loff = np.concatenate([[0], np.cumsum(gdepth + 1)])
    out[S["loff"]:S["loff"] + G + 1] = loff
    for d in (0, 1):
        # per (graph, layer) row counts from lstart
        g2 = np.repeat(np.arange(B), depth[d])
        t2 = np.arange(g2.shape[0]) - np.repeat(np.cumsum(depth[d]) - depth[d], depth[d])
        base2 = node_ptr[g2] + g2 + t2
        ls = ws[P["lstart%d" % d]:P["lstart%d" % d] + N + B].astype(np.int64)
        cnt2 = ls[base2 + 1] - ls[base2]
        k2 = grp[g2]
        # rows per (group, layer), padded exclusive prefix (df_count_kernel, df_prefix_kernel)
        cnt = np.zeros(int(loff[G]), dtype=np.int64)
        np.add.at(cnt, loff[k2] + t2, cnt2)
        padded = (cnt + DF_RB - 1) // DF_RB * DF_RB
        pref = np.zeros_like(cnt)
        nblk = np.zeros(G, dtype=np.int64)
        for k in range(G):
            a, b = int(loff[k]), int(loff[k + 1])
            seg = padded[a:b].copy()
            seg[-1] = 0                      # the table has depth + 1 entries; the last holds the total
            c = np.cumsum(seg) - seg
            c[-1] = seg[:-1].sum()
            pref[a:b] = c
            nblk[k] = c[-1] // DF_RB
        out[S["lcnt%d" % d]:S["lcnt%d" % d] + int(loff[G])] = pref
        base = np.cumsum(nblk * DF_RB) - nblk * DF_RB
        gt = np.empty(2 * G, dtype=np.int64)
        gt[0::2], gt[1::2] = base, nblk
        out[S["gtab%d" % d]:S["gtab%d" % d] + 2 * G] = gt
        # first record of every (graph, layer) inside its group (df_lbase_kernel): graphs of a group in id order
        order2 = np.lexsort((g2, t2, k2))
        c = cnt2[order2]
        run_key = k2[order2] * (int(gdepth.max()) + 2) + t2[order2]
        start = np.concatenate([[True], run_key[1:] != run_key[:-1]])
        csum = np.cumsum(c) - c
        run_base = np.maximum.accumulate(np.where(start, csum, 0))
        within = csum - run_base
        glb = np.empty(g2.shape[0], dtype=np.int64)
        glb[order2] = pref[loff[k2[order2]] + t2[order2]] + within
        out[S["glbase%d" % d] + base2] = glb
        # records (df_records_kernel)
        order = ws[P["order%d" % d]:P["order%d" % d] + N].astype(np.int64)
        slot = ws[P["slot%d" % d]:P["slot%d" % d] + N].astype(np.int64)
        rowrec = ws[P["rowrec%d" % d]:P["rowrec%d" % d] + 16 * N].reshape(N, 16)
        p = np.arange(N)
        v = order[p]
        recs = rowrec[slot[v]]
        g = recs[:, 3].astype(np.int64)
        # layer of sorted position p inside its graph: ls[t] <= p < ls[t + 1]
        glb_full = np.zeros(N + B + 1, dtype=np.int64)
        glb_full[base2] = glb
        # expand per (graph, layer) -> per position
        reps = cnt2
        t_of_p = np.empty(N, dtype=np.int64)
        b_of_p = np.empty(N, dtype=np.int64)
        lsv = ls[base2]
        idx = np.repeat(np.arange(base2.shape[0]), reps)
        pos_sorted = np.repeat(lsv, reps) + (np.arange(idx.shape[0]) - np.repeat(np.cumsum(reps) - reps, reps))
        t_of_p[pos_sorted] = t2[idx]
        b_of_p[pos_sorted] = base2[idx]
        rec_idx = base[grp[g]] + glb_full[b_of_p] + (p - ls[b_of_p])
        dst = out[S["grec%d" % d]:S["grec%d" % d] + 16 * (4 * N + 4)].reshape(-1, 16)
        dst[rec_idx] = recs
    return out


def attach_plan(batch, num_graphs: Optional[int] = None, dataflow_groups: int = 0, cost_layer: int = 4, cost_row: int = 1):
    """Build the host plan of a collated batch and attach it (`_dagnn_plan`: int32 tensor that moves with
    `batch.to(device)`; `_dagnn_plan_meta`: sizes + the host schedule).  `DAGNN.forward` uses it when present.  With
    `dataflow_groups` the schedule of the persistent dataflow kernel is built here as well (`_dagnn_df`)."""
    B = int(num_graphs if num_graphs is not None else getattr(batch, "num_graphs", int(batch.batch[-1]) + 1))
    ea = getattr(batch, "edge_attr", None)
    ws, sched, splits = build_plan_host(batch.edge_index, batch._bi_layer_idx0, batch._bi_layer_idx1, batch.batch, B, ea)
    batch._dagnn_plan = torch.from_numpy(ws)
    N, E = int(batch.batch.shape[0]), int(batch.edge_index.shape[1])
    R = 0 if ea is None else int(ea.reshape(E, -1).shape[1])
    batch._dagnn_plan_meta = dict(N=N, E=E, B=B, R=R, schedule=sched, splits=splits)
    if dataflow_groups > 0:   # the dataflow kernel's schedule too (what `engine.dataflow_groups` gives for the model)
        batch._dagnn_df = torch.from_numpy(build_dataflow_schedule_host(ws, N, E, B, R, dataflow_groups, cost_layer, cost_row))
        batch._dagnn_plan_meta["dataflow_key"] = (int(dataflow_groups), int(cost_layer), int(cost_row))
    return batch
