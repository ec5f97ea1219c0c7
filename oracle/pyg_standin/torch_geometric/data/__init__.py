import re

import torch


class Data(object):
    """Attribute bag with the small protocol dvae/batch.py's collation relies on."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        self.x, self.edge_index, self.edge_attr, self.y, self.pos = x, edge_index, edge_attr, y, pos
        for k, v in kwargs.items():
            setattr(self, k, v)

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items()
                if v is not None and not (k.startswith("__") and k.endswith("__"))]

    def __contains__(self, key):
        return key in self.keys

    def __cat_dim__(self, key, value):
        return -1 if re.search("(index|face)", key) else 0

    def __inc__(self, key, value):
        return self.num_nodes if re.search("(index|face)", key) else 0

    @property
    def num_nodes(self):
        if getattr(self, "__num_nodes__", None) is not None:
            return self.__num_nodes__
        if self.x is not None:
            return self.x.size(0)
        if self.edge_index is not None and self.edge_index.numel() > 0:
            return int(self.edge_index.max()) + 1
        return None

    @num_nodes.setter
    def num_nodes(self, n):
        self.__num_nodes__ = n

    def _apply(self, fn):
        for k in self.keys:
            v = self[k]
            if isinstance(v, torch.Tensor):
                self[k] = fn(v)
        return self

    def to(self, device, *a, **k):
        return self._apply(lambda t: t.to(device))

    def contiguous(self):
        return self._apply(lambda t: t.contiguous())

    def debug(self):
        pass


from .batch import Batch  # noqa: E402,F401
