"""Empty stubs: dvae/batch.py imports these names but the encoder path never builds one."""


class SparseTensor:  # pragma: no cover - never instantiated on the hot path
    pass


def cat(*a, **k):  # pragma: no cover
    raise NotImplementedError
