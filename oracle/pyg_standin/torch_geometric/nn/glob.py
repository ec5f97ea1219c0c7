from torch_scatter import scatter

__all__ = ["global_add_pool", "global_mean_pool", "global_max_pool"]


def _rows(batch, size):
    return int(batch.max()) + 1 if size is None else size


def global_add_pool(x, batch, size=None):
    return scatter(x, batch, dim=0, dim_size=_rows(batch, size), reduce="add")


def global_mean_pool(x, batch, size=None):
    return scatter(x, batch, dim=0, dim_size=_rows(batch, size), reduce="mean")


def global_max_pool(x, batch, size=None):
    return scatter(x, batch, dim=0, dim_size=_rows(batch, size), reduce="max")
