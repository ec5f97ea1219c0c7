import inspect

import torch
from torch_scatter import scatter

_SPECIAL = ("index", "ptr", "size_i", "size_j")


class MessagePassing(torch.nn.Module):
    """propagate() = gather per-edge arguments, message(), scatter-reduce (PyG-1.6 contract).

    flow='source_to_target': `x_j` is taken at edge_index[0], `x_i` at edge_index[1] and the
    reduction runs over edge_index[1]; 'target_to_source' swaps the two rows. The result has as
    many rows as the node tensors that were passed in, zero where no edge lands.
    """

    def __init__(self, aggr="add", flow="source_to_target", node_dim=0):
        super().__init__()
        assert flow in ("source_to_target", "target_to_source")
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim
        self._msg_args = [p for p in inspect.signature(self.message).parameters]

    def propagate(self, edge_index, size=None, **kwargs):
        tgt, src = (1, 0) if self.flow == "source_to_target" else (0, 1)
        n_rows = [None, None]  # [rows seen through *_j, rows seen through *_i]
        call = {}
        for name in self._msg_args:
            if name in _SPECIAL:
                continue
            if name.endswith("_i") or name.endswith("_j"):
                value = kwargs.get(name[:-2])
                side = 0 if name.endswith("_j") else 1
                if isinstance(value, torch.Tensor):
                    if n_rows[side] is None:
                        n_rows[side] = value.size(0)
                    value = value.index_select(0, edge_index[src if side == 0 else tgt])
                call[name] = value
            else:
                call[name] = kwargs.get(name)
        size_i = n_rows[1] if n_rows[1] is not None else n_rows[0]
        if "index" in self._msg_args:
            call["index"] = edge_index[tgt]
        if "ptr" in self._msg_args:
            call["ptr"] = None
        if "size_i" in self._msg_args:
            call["size_i"] = size_i
        if "size_j" in self._msg_args:
            call["size_j"] = n_rows[0] if n_rows[0] is not None else n_rows[1]
        msg = self.message(**call)
        out = scatter(msg, edge_index[tgt], dim=0, dim_size=size_i, reduce=self.aggr)
        return self.update(out)

    def message(self, x_j):  # pragma: no cover - always overridden
        return x_j

    def update(self, aggr_out):
        return aggr_out
