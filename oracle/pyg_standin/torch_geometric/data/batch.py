import torch

from . import Data


class Batch(Data):
    """PyG-1.6 collation: concatenate per key; keys matching (index|face) are concatenated along
    the last dim and offset by the running node count; `batch[v]` = graph id."""

    def __init__(self, batch=None, **kwargs):
        super().__init__(**kwargs)
        self.batch = batch

    @staticmethod
    def from_data_list(data_list, follow_batch=[]):
        keys = []
        for d in data_list:
            for k in d.keys:
                if k not in keys:
                    keys.append(k)
        out = Batch()
        cols = {k: [] for k in keys}
        bvec, cum = [], 0
        for i, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                item = d[k]
                if isinstance(item, torch.Tensor):
                    inc = d.__inc__(k, item)
                    if inc and item.dtype != torch.bool and cum != 0:
                        item = item + cum
                    if item.dim() == 0:
                        item = item.unsqueeze(0)
                cols[k].append(item)
            bvec.append(torch.full((n,), i, dtype=torch.long))
            cum += n
        ref = data_list[0]
        for k in keys:
            items = cols[k]
            if isinstance(items[0], torch.Tensor):
                out[k] = torch.cat(items, ref.__cat_dim__(k, items[0]))
            elif isinstance(items[0], (int, float)):
                out[k] = torch.tensor(items)
            else:
                out[k] = items
        out.batch = torch.cat(bvec, 0)
        out.num_graphs = len(data_list)
        return out.contiguous()
