"""Topological layering of DAGs (the layer ids the message-passing path consumes).

Restates the contract of the reference's `src/utils_dag.py:8-52` (`top_sort`,
`add_order_info_01`, `add_order_info`): layer(v) = length of the longest path from any source
to v.  The reference peels frontiers with O(depth * (n + e)) numpy work and then re-verifies
in O(n * e) python (`assert_order`, :55-67); here it is one Kahn pass, O(n + e).
"""
from __future__ import annotations

import numpy as np
import torch


def longest_path_layers(edge_index: np.ndarray, num_nodes: int) -> np.ndarray:
    """layer[v] = longest-path distance from any source (same ids as reference `top_sort`).

    edge_index: int array [2, e], row 0 = source, row 1 = target.  Raises on cycles.
    """
    src = np.asarray(edge_index[0], dtype=np.int64)
    dst = np.asarray(edge_index[1], dtype=np.int64)
    n = int(num_nodes)
    layer = np.zeros(n, dtype=np.int64)
    if n == 0:
        return layer
    indeg = np.bincount(dst, minlength=n).astype(np.int64)
    # CSR of out-edges
    order = np.argsort(src, kind="stable")
    nbr = dst[order]
    ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=n), out=ptr[1:])
    frontier = np.flatnonzero(indeg == 0)
    done = 0
    level = 0
    while frontier.size:
        layer[frontier] = level
        done += frontier.size
        # all out-edges of the frontier
        starts, ends = ptr[frontier], ptr[frontier + 1]
        cnt = ends - starts
        if cnt.sum() == 0:
            break
        idx = np.repeat(starts - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(cnt.sum())
        tgt = nbr[idx]
        dec = np.bincount(tgt, minlength=n)
        indeg -= dec
        frontier = np.flatnonzero((indeg == 0) & (dec > 0))
        level += 1
    if done != n:
        raise ValueError("graph has a cycle: %d of %d nodes layered" % (done, n))
    return layer


def check_layers(edge_index: np.ndarray, layer: np.ndarray) -> bool:
    """Validity check the reference does in `assert_order` (src/utils_dag.py:55-67), vectorised:
    every edge goes from a strictly lower layer to a higher one."""
    return bool(np.all(layer[edge_index[0]] < layer[edge_index[1]]))


def add_order_info_01(graph) -> None:
    """Attach `_bi_layer_idx0/1` (layer ids) and `_bi_layer_index0/1` (= arange(n)) to a graph
    object, as `src/utils_dag.py:39-52` does.  Keys containing "index" get node offsets when a
    PyG-style collation concatenates graphs; "idx" keys do not (`dagnn.py:129`)."""
    ei = graph.edge_index.cpu().numpy()
    n = int(graph.num_nodes)
    l0 = longest_path_layers(ei, n)
    l1 = longest_path_layers(ei[::-1], n)
    ns = torch.arange(n, dtype=torch.long)
    graph._bi_layer_idx0 = torch.from_numpy(l0)
    graph._bi_layer_index0 = ns
    graph._bi_layer_idx1 = torch.from_numpy(l1)
    graph._bi_layer_index1 = ns.clone()


def add_order_info(graph) -> None:
    """`bi_layer_index [2, 2, n]` = [[layer_fwd, ids], [layer_bwd, ids]] (`src/utils_dag.py:70-76`)."""
    ei = graph.edge_index.cpu().numpy()
    n = int(graph.num_nodes)
    ns = torch.arange(n, dtype=torch.long)
    l0 = torch.from_numpy(longest_path_layers(ei, n))
    l1 = torch.from_numpy(longest_path_layers(ei[::-1], n))
    graph.bi_layer_index = torch.stack([torch.stack([l0, ns]), torch.stack([l1, ns])], dim=0)


def add_order_info_batch(batch, num_graphs=None, check: bool = False):
    """`add_order_info_01` (src/utils_dag.py:39-52) for a whole collated batch that is already on the GPU:
    sets `_bi_layer_idx0/1` (HIP kernel, csrc/toposort.hip) and `_bi_layer_index0/1` (= arange(N), which is what
    PyG collation turns the per-graph aranges into).  `check=True` synchronises and raises on a cyclic graph."""
    from . import engine
    B = int(num_graphs if num_graphs is not None else getattr(batch, "num_graphs", None) or int(batch.batch[-1]) + 1)
    lf, lb, status = engine.topo_layers(batch.edge_index, batch.batch, B)
    if check and int(status):
        raise ValueError("a graph of the batch has a cycle")
    ids = torch.arange(batch.batch.numel(), device=batch.batch.device)
    batch._bi_layer_idx0, batch._bi_layer_index0 = lf, ids
    batch._bi_layer_idx1, batch._bi_layer_index1 = lb, ids.clone()
    return batch
