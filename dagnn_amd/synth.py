"""Synthetic inputs of the benchmark configurations.

The real datasets are not available offline (`ogbg-code2` downloads at first use,
`ogb/graphproppred/dataset_pyg.py:106-118`; `asia_200k.txt` is missing from the checkout), so
every measured figure uses these generators:

* code2-like AST batches (cfg 2/3/5): random DFS-preorder trees + next-token edges between
  consecutive leaves, exactly the edge layout `augment_edge2` produces
  (`ogbg-code/utils2.py:30-78`);
* ENAS rows -> 8-node DAGs (cfg 1), decoded as `decode_ENAS_to_pygraph` (`dvae/util.py:343-385`);
* Bayesian-network rows -> 10-node DAGs (cfg 4), decoded as `decode_BN_to_pygraph`
  (`dvae/util.py:290-339`).
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .dag_utils import add_order_info, add_order_info_01
from .data import GraphBatch, GraphData


# --------------------------------------------------------------------------- code2-like ASTs
def gen_ast(rng: np.random.Generator, n: int) -> dict:
    """One random AST in DFS pre-order with next-token edges (Appendix E, verbatim draw order)."""
    parent = np.full(n, -1)
    depth = np.zeros(n, dtype=np.int64)
    stack = [0]
    for i in range(1, n):
        k = min(len(stack) - 1, rng.geometric(0.45) - 1)
        for _ in range(k):
            stack.pop()
        parent[i] = stack[-1]
        depth[i] = depth[stack[-1]] + 1
        stack.append(i)
    haschild = np.zeros(n, bool)
    haschild[parent[1:]] = True
    leaves = np.where(~haschild)[0]
    ast = np.stack([parent[1:], np.arange(1, n)])  # parent -> child
    nt = np.stack([leaves[:-1], leaves[1:]])  # next-token
    ei = np.concatenate([ast, nt], 1)
    ea = np.concatenate([np.zeros((ast.shape[1], 2)),
                         np.stack([np.ones(nt.shape[1]), np.zeros(nt.shape[1])], 1)], 0).astype(np.float32)
    x = np.stack([rng.integers(0, 98, n), rng.integers(0, 10030, n)], 1)
    return dict(n=n, ei=ei, ea=ea, x=x, depth=depth)


def code2_graphs(seed: int, num_graphs: int, mean_n: int = 125, max_n: int = 1000) -> List[GraphData]:
    """`num_graphs` code2-like graphs drawn one after another from `default_rng(seed)`."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num_graphs):
        n = int(np.clip(rng.lognormal(np.log(mean_n) - 0.18, 0.6), 11, max_n))
        g = gen_ast(rng, n)
        d = GraphData(
            x=torch.from_numpy(g["x"]).long(),
            node_depth=torch.from_numpy(g["depth"]).long().view(-1, 1),
            edge_index=torch.from_numpy(g["ei"]).long(),
            edge_attr=torch.from_numpy(g["ea"]),
        )
        add_order_info_01(d)
        out.append(d)
    return out


def code2_batch(seed: int = 0, num_graphs: int = 128, mean_n: int = 125, max_n: int = 1000) -> GraphBatch:
    """The headline batch: seed 0, B=128 -> N=16 561, E=25 377, T=374."""
    return GraphBatch.from_data_list(code2_graphs(seed, num_graphs, mean_n, max_n))


# --------------------------------------------------------------------------- D-VAE graphs
def _adj_to_graph(adj: np.ndarray, types: Sequence[int], n_types: int) -> GraphData:
    # networkx.DiGraph(adj).edges enumerates the non-zeros row by row (source-major), which
    # is what the reference's decoders hand to torch (util.py:325,372)
    src, dst = np.nonzero(adj)
    x = torch.zeros(len(types), n_types)
    x[torch.arange(len(types)), torch.tensor(list(types))] = 1.0
    g = GraphData(x=x, edge_index=torch.from_numpy(np.stack([src, dst])).long())
    add_order_info(g)
    g.vs = [{"type": int(t)} for t in types]
    return g


def decode_enas_row(row, n_types: int = 6) -> GraphData:
    """ENAS row [[type, conn_0..conn_{i-1}], ...] -> 8-node DAG with start(0)/end(1) vertices:
    chain i -> i+1 plus the listed skip connections (`dvae/util.py:343-385`)."""
    n_types += 2
    n = len(row)
    adj = np.zeros((n_types, n_types))
    types = [0]
    for i, node in enumerate(row):
        types.append(node[0] + 2)
        adj[i, i + 1] = 1
        for j, e in enumerate(node[1:]):
            if e == 1:
                adj[j, i + 1] = 1
    types.append(1)
    adj[n, n + 1] = 1
    return _adj_to_graph(adj, types, n_types)


def decode_bn_row(row, n_types: int = 8) -> GraphData:
    """BN row -> 10-node DAG: parentless nodes hang off the start vertex, loose ends feed the end
    vertex (`dvae/util.py:290-339`)."""
    n_types += 2
    n = len(row)
    adj = np.zeros((n_types, n_types))
    loose = [True] * n
    types = [0]
    for i, node in enumerate(row):
        types.append(node[0] + 2)
        if sum(node[1:]) == 0:
            adj[0, i + 1] = 1
        else:
            for j, e in enumerate(node[1:]):
                if e == 1:
                    adj[j + 1, i + 1] = 1
                    loose[j] = False
    types.append(1)
    for j, flag in enumerate(loose):
        if flag:
            adj[j + 1, n + 1] = 1
    return _adj_to_graph(adj, types, n_types)


def enas_rows(seed: int, num_graphs: int) -> list:
    """ENAS-shaped rows (6 op types, 6 layers, random skips) when the real file is not at hand."""
    rng = np.random.default_rng(seed)
    return [[[int(rng.integers(0, 6))] + [int(rng.random() < 0.4) for _ in range(i)] for i in range(6)]
            for _ in range(num_graphs)]


def bn_rows(seed: int, num_graphs: int) -> list:
    """Synthetic Bayesian-network rows (Appendix E): permutation of 8 types, parent w.p. 0.3."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(num_graphs):
        perm = rng.permutation(8)
        out.append([[int(perm[i])] + [int(rng.random() < 0.3) for _ in range(i)] for i in range(8)])
    return out


def dvae_batch(graphs: Sequence[GraphData]) -> GraphBatch:
    """Collate D-VAE graphs the way `dvae/batch.py:26-146` does (`bi_layer_index` row 1 shifted)."""
    return GraphBatch.from_data_list([g for g in graphs])
