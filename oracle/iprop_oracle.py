"""TEST INFRASTRUCTURE - not part of the product path.  Only tests/ may import this module.

CPU restatement (dense torch ops) of the decoder-side single-vertex step of the D-VAE models,
`_ipropagate_to` (`/root/reference/dvae/dagnn.py:187-239`, `/root/reference/dvae/dagnn_bn.py:179-238`): the checker
for `dagnn_amd.dvae._DvaeDagnn._ipropagate_to`, which runs the same step as one HIP launch (`dagnn_iprop_step`).
Pinned against the reference's own function on the `iprop_*` fixtures (tests/test_cpu_host.py::test_ipropagate_to_oracle_matches_reference).

`self` is any module with the D-VAE attributes the reference's function touches (`nvt`, `hs`, `vs`, `max_n`,
`num_layers`, `node_aggr_0[l].attn_lin`, `_use_vids`); the tests pass a `dagnn_amd.dvae` model kept on the CPU - only
its parameters are read."""
import torch
import torch.nn.functional as F


def _zeros(self, n, length):
    return torch.zeros(n, length)


def _one_hot(idx, length):
    """`models_pyg.py:98-107`: a list gives one row per entry (None for an empty list), an int one row."""
    if type(idx) in (list, range):
        if len(idx) == 0:
            return None
        ids = torch.tensor(list(idx), dtype=torch.long).view(-1, 1)
    else:
        ids = torch.tensor([[int(idx)]], dtype=torch.long)
    return torch.zeros(ids.shape[0], length).scatter_(1, ids, 1)


def ipropagate_to(self, G, v, propagator, H=None, reverse=False):
    """New states at vertex `v` of every graph in `G` that has one, from the states of its predecessors
    (`dvae/dagnn.py:187-239`, `dvae/dagnn_bn.py:179-238`; called by the decoder as `_update_iv`,
    `models_pyg.py:247-250`, with `propagator = self.grud`).  `G` holds igraph-style graphs: `g.vcount()`,
    `g.predecessors(v)`, `g.vs[x]['type']`, `g.vs[x]['H_forward<l>']` ([1, hs] tensors, written for `v`).

    Reproduced as the reference computes it, quirks included: the predecessor lists are padded to the longest one
    with zero rows and the attention soft-max runs over the padding as well (a zero key scores `w_q.q + b`, so
    padded slots take weight away from the real predecessors) - which is why this is NOT the encoder's aggregate;
    and the aggregate `H` is computed for stacked layer 0 only and then reused by every layer above (`H` is no longer
    None in the later iterations of the reference's loop).  Dense torch ops, op for op as the reference."""
    assert not reverse
    G = [g for g in G if g.vcount() > v]
    if len(G) == 0:
        return None
    if H is not None:
        H = H[list(range(len(G)))]   # the reference indexes with the positions of the already filtered list
    X = _one_hot([g.vs[v]["type"] for g in G], self.nvt)
    Hv = X
    for l in range(self.num_layers):
        name = "H_forward%d" % l
        if H is None:
            preds = [g.predecessors(v) for g in G]
            P = max(len(p) for p in preds)
            if P == 0:
                H = _zeros(self, len(G), self.hs)
            else:
                def padded(rows, width):   # [len(G), P, width], zero rows behind the real ones
                    return torch.stack([torch.cat(r + [_zeros(self, P - len(r), width)], 0) for r in rows], 0)
                states = [[g.vs[x][name] for x in p] for g, p in zip(G, preds)]
                values = padded(states, self.hs)
                if self._use_vids:     # keys = [state ; one-hot of the predecessor's vertex id] (`dagnn.py:208-214`)
                    keys = padded([[torch.cat([h, _one_hot(x, self.max_n)], 1) for h, x in zip(st, p)]
                                   for st, p in zip(states, preds)], self.vs)
                else:
                    keys = values
                query = X if l == 0 else torch.cat([g.vs[v]["H_forward%d" % (l - 1)] for g in G], 0)
                lin = self.node_aggr_0[l].attn_lin     # AttnConv.forward with edge_index=None (`dagnn.py:391-399`)
                scores = lin(torch.cat([query[:, None, :].expand(-1, P, -1), keys], -1)).view(len(G), P)
                H = torch.einsum("bi,bij->bj", F.softmax(scores, dim=-1), values)
        Hv = propagator[l](Hv, H)
        for i, g in enumerate(G):
            g.vs[v][name] = Hv[i:i + 1]
    return Hv
