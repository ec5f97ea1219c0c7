import torch
from torch_scatter import scatter


def softmax(src, index, ptr=None, num_nodes=None):
    """PyG-1.6 segment softmax: exp(src - segmax[index]) / (segsum[index] + 1e-16)."""
    if num_nodes is None:
        num_nodes = int(index.max()) + 1
    seg_max = scatter(src, index, dim=0, dim_size=num_nodes, reduce="max")
    out = (src - seg_max.index_select(0, index)).exp()
    seg_sum = scatter(out, index, dim=0, dim_size=num_nodes, reduce="sum")
    return out / (seg_sum.index_select(0, index) + 1e-16)
