"""Minimal stand-in for the python-igraph symbols the D-VAE single-vertex step touches
(`dvae/dagnn.py:187-239`, `dvae/dagnn_bn.py:179-238`): a directed graph whose vertices carry attribute dictionaries.

    g = Graph(directed=True); g.add_vertices(n); g.add_edge(u, v)
    g.vcount(); g.vs[v]['type']; g.vs[v]['H_forward0'] = tensor; g.predecessors(v)   # ascending vertex ids

Everything else of igraph (the decoder's graph surgery) stays unimplemented: out of scope."""


class _Vertex(dict):
    pass


class _VertexSeq(list):
    def __setitem__(self, key, value):  # g.vs['type'] = [...]  (attribute for all vertices)
        if isinstance(key, str):
            for vtx, val in zip(self, value):
                vtx[key] = val
        else:
            list.__setitem__(self, key, value)

    def __getitem__(self, key):
        if isinstance(key, str):
            return [vtx[key] for vtx in self]
        return list.__getitem__(self, key)


class Graph:
    def __init__(self, directed=True, **_):
        if not directed:
            raise NotImplementedError("igraph stand-in: directed graphs only")
        self.vs = _VertexSeq()
        self._pred = []

    def add_vertices(self, n):
        for _ in range(n):
            self.add_vertex()

    def add_vertex(self, **attrs):
        self.vs.append(_Vertex(attrs))
        self._pred.append([])

    def add_edge(self, u, v):
        self._pred[v].append(u)

    def add_edges(self, edges):
        for u, v in edges:
            self.add_edge(u, v)

    def vcount(self):
        return len(self.vs)

    def predecessors(self, v):
        return sorted(self._pred[v])
