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


def attach_plan(batch, num_graphs: Optional[int] = None):
    """Build the host plan of a collated batch and attach it (`_dagnn_plan`: int32 tensor that moves with
    `batch.to(device)`; `_dagnn_plan_meta`: sizes + the host schedule).  `DAGNN.forward` uses it when present."""
    B = int(num_graphs if num_graphs is not None else getattr(batch, "num_graphs", int(batch.batch[-1]) + 1))
    ea = getattr(batch, "edge_attr", None)
    ws, sched, splits = build_plan_host(batch.edge_index, batch._bi_layer_idx0, batch._bi_layer_idx1, batch.batch, B, ea)
    batch._dagnn_plan = torch.from_numpy(ws)
    batch._dagnn_plan_meta = dict(N=int(batch.batch.shape[0]), E=int(batch.edge_index.shape[1]), B=B,
                                  R=0 if ea is None else int(ea.reshape(batch.edge_index.shape[1], -1).shape[1]),
                                  schedule=sched, splits=splits)
    return batch
