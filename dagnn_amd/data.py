"""Minimal graph containers with the PyG `Data` / `Batch` attribute contract.

`DAGNN.forward(G)` only does attribute access on `G`, so a real
`torch_geometric.data.Batch`, a `types.SimpleNamespace` or the `GraphBatch` below all work.  PyG
is not installable in the build image, hence this stand-alone container; collation follows the
PyG-1.6 rule the reference relies on (`dagnn.py:129`, `dvae/batch.py:54-59`): tensors whose key
contains "index" or "face" are concatenated along the last dim and shifted by the running node
count, everything else is concatenated along dim 0 unshifted.
"""
from __future__ import annotations

import re
from typing import Iterable, List, Sequence

import torch

_INDEX_KEY = re.compile("(index|face)")
# dvae/batch.py:54-59: only row [:, 1] (node ids) of these [2, 2, n] tensors is shifted
_BI_KEYS = ("bi_layer_index", "bi_layer_parent_index")


class GraphData(object):
    """One graph: an attribute bag (`x`, `edge_index`, `edge_attr`, ... any tensor attribute)."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    # -- PyG-like protocol -------------------------------------------------------------
    @property
    def keys(self) -> List[str]:
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("__")]

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    @property
    def num_nodes(self):
        n = self.__dict__.get("__num_nodes__")
        if n is not None:
            return n
        x = self.__dict__.get("x")
        if x is not None:
            return x.size(0)
        ei = self.__dict__.get("edge_index")
        if ei is not None and ei.numel() > 0:
            return int(ei.max()) + 1
        return 0

    @num_nodes.setter
    def num_nodes(self, n):
        self.__dict__["__num_nodes__"] = n

    def _apply(self, fn):
        for k in self.keys:
            v = self[k]
            if isinstance(v, torch.Tensor):
                self[k] = fn(v)
        return self

    def to(self, device, *args, **kwargs):
        if not args and not kwargs and isinstance(device, torch.device):   # already there: nothing to convert (a
            x = self.__dict__.get("x")                                     # forward pass calls this on every batch)
            if isinstance(x, torch.Tensor) and x.device == device and all(
                    v.device == device for v in self.__dict__.values() if isinstance(v, torch.Tensor)):
                return self
        return self._apply(lambda t: t.to(device, *args, **kwargs))

    def contiguous(self):
        return self._apply(lambda t: t.contiguous())

    def clone(self):
        out = self.__class__()
        for k, v in self.__dict__.items():
            out.__dict__[k] = v.clone() if isinstance(v, torch.Tensor) else v
        return out


class GraphBatch(GraphData):
    """A batch of graphs as one big disconnected graph plus the `batch` assignment vector."""

    @staticmethod
    def from_data_list(data_list: Sequence[GraphData]) -> "GraphBatch":
        keys: List[str] = []
        for d in data_list:
            for k in d.keys:
                if k not in keys:
                    keys.append(k)
        out = GraphBatch()
        cols = {k: [] for k in keys}
        bvec, ptr, cum = [], [0], 0
        for i, d in enumerate(data_list):
            n = int(d.num_nodes)
            for k in keys:
                item = d[k]
                if isinstance(item, torch.Tensor):
                    if k in _BI_KEYS:
                        item = item.clone()
                        item[:, 1] = item[:, 1] + cum
                    elif _INDEX_KEY.search(k) and item.dtype != torch.bool and cum != 0:
                        item = item + cum
                    if item.dim() == 0:
                        item = item.unsqueeze(0)
                cols[k].append(item)
            bvec.append(torch.full((n,), i, dtype=torch.long))
            cum += n
            ptr.append(cum)
        for k in keys:
            items = cols[k]
            if isinstance(items[0], torch.Tensor):
                out[k] = torch.cat(items, -1 if _INDEX_KEY.search(k) else 0)
            elif isinstance(items[0], (int, float)):
                out[k] = torch.tensor(items)
            else:
                out[k] = items
        out.batch = torch.cat(bvec, 0) if bvec else torch.zeros(0, dtype=torch.long)
        out.ptr = torch.tensor(ptr, dtype=torch.long)
        out.num_graphs = len(data_list)
        return out.contiguous()

    @property
    def num_nodes(self):
        b = self.__dict__.get("batch")
        if b is not None:
            return b.numel()
        return GraphData.num_nodes.fget(self)


def shard_by_nodes(num_nodes_per_graph: Iterable[int], ndevices: int) -> List[int]:
    """Contiguous, node-balanced split of one loader batch over `ndevices` devices.

    Restates the rule of the reference's `Collater.collate` (`ogbg-code/tg/dataloader.py:17-27`):
    graph g goes to device floor(ndev * midpoint(cumsum[g], cumsum[g+1]) / total); devices that
    receive no graph are dropped.  Returns the split points [0, ..., len(graphs)].
    """
    count = torch.tensor(list(num_nodes_per_graph))
    if count.numel() == 0:
        return [0]
    cumsum = torch.cat([count.new_zeros(1), count.cumsum(0)], dim=0)
    device_id = ndevices * cumsum.to(torch.float) / cumsum[-1].item()
    device_id = ((device_id[:-1] + device_id[1:]) / 2.0).to(torch.long)
    split = torch.cat([device_id.new_zeros(1), device_id.bincount().cumsum(0)], dim=0)
    return torch.unique(split, sorted=True).tolist()


def collate_sharded(data_list: Sequence[GraphData], ndevices: int) -> List[GraphBatch]:
    """`Collater.collate` (`tg/dataloader.py:13-35`): one `GraphBatch` per non-empty device."""
    split = shard_by_nodes([d.num_nodes for d in data_list], ndevices)
    return [GraphBatch.from_data_list(data_list[split[i]:split[i + 1]]) for i in range(len(split) - 1)]


def augment_edge2(data):
    """Edge augmentation of the ogbg-code2 transform (`ogbg-code/utils2.py:30-78`): keep the AST edges with
    `edge_attr = [0, 0]` and append one next-token edge `[1, 0]` between consecutive attributed nodes (the
    nodes are already in DFS order, so consecutive means consecutive indices of `node_is_attributed == 1`).
    No inverse edges (the reference has them commented out).  Modifies and returns `data`."""
    ast = data.edge_index
    tok = torch.where(data.node_is_attributed.view(-1) == 1)[0]
    nxt = torch.stack([tok[:-1], tok[1:]], dim=0)
    attr_ast = torch.zeros((ast.size(1), 2))
    attr_nxt = torch.cat([torch.ones(nxt.size(1), 1), torch.zeros(nxt.size(1), 1)], dim=1)
    data.edge_index = torch.cat([ast, nxt], dim=1)
    data.edge_attr = torch.cat([attr_ast, attr_nxt], dim=0)
    return data


def collate_with_plan(data_list: Sequence[GraphData]) -> "GraphBatch":
    """Collate (PyG rule, as `ogbg-code/tg/dataloader.py:13-35` does per device chunk) and build the batch's
    plan on the host, in the loader worker: `forward` then launches no plan kernels and
    needs no device->host read."""
    from .host_plan import attach_plan
    return attach_plan(GraphBatch.from_data_list(data_list))
