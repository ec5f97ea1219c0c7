"""Stand-in for the two torch-scatter entry points the reference imports (test infrastructure).

Semantics restated from the torch-scatter documentation: reduce `src` rows into `dim_size` rows
selected by `index` along `dim`; rows nothing lands on are zero.
"""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    if dim < 0:
        dim += src.dim()
    if dim != 0:
        raise NotImplementedError("stand-in only supports dim=0")
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    shape = (dim_size,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    if reduce in ("sum", "add"):
        return res.scatter_add_(0, idx, src)
    if reduce == "mean":
        res.scatter_add_(0, idx, src)
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.scatter_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        return res / cnt.clamp(min=1).view(-1, *([1] * (src.dim() - 1)))
    if reduce == "max":
        return res.scatter_reduce_(0, idx, src, "amax", include_self=False)
    if reduce == "min":
        return res.scatter_reduce_(0, idx, src, "amin", include_self=False)
    raise ValueError(reduce)


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")
