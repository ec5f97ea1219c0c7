"""Empty stub: only the D-VAE decoder (out of scope) touches igraph."""


class Graph:  # pragma: no cover
    def __init__(self, *a, **k):
        raise NotImplementedError("igraph stand-in: decoder paths are out of scope")
