"""Fixture loading and model construction shared by the CPU and GPU test tiers."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np
import torch

import dagnn_amd
from dagnn_amd import DAGNN, DAGNN_BN, DAGNN_NA, ASTNodeEncoder
from oracle.seeding import seeded_fill

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CODE2 = ["code2_h32_bidir", "code2_h256_bidir", "code2_h512_L5", "code2_h300_L3", "code2_h64_unidir",
         "code2_h64_numclass", "code2_h128_deep", "code2_h64_attn_x", "code2_h64_self_attn_h",
         "code2_h64_self_attn_x"]
VARIANTS = ["var_h64_" + t for t in ("gated_sum", "gated_nobias", "mattn_h", "add", "max", "aggx_attn_h", "aggx_add",
                                        "recurr0", "recurr0_gated")]
GRAD = ["grad_h32_bidir", "grad_h256_bidir", "grad_h128_deep", "grad_h64_L3_wx", "grad_h300_bidir", "grad_h64_unidir",
        "grad_h64_mean_all"]
GRAD_VAR = ["grad_var_h64_" + t for t in ("gated_sum", "gated_nobias", "mattn_h", "add", "mattn_h_L3", "max", "recurr0_gated",
                                           "recurr0_mattn", "recurr0_attn_h", "recurr0_attn_x", "recurr0_self_attn_h",
                                           "aggx_attn_h", "aggx_add", "aggx_gated", "aggx_mattn", "aggx_max_recurr0")]
DVAE_GRAD = ["grad_na_h64_unidir", "grad_bn_h64_bidir", "grad_na_h501_unidir", "grad_bn_h501_bidir",
             "grad_na_h64_self_attn_h", "grad_bn_h64_self_attn_h",
             "grad_na_h64_gated_sum", "grad_na_h64_add", "grad_na_h64_max", "grad_bn_h64_add", "grad_bn_h64_max"]
DVAE = ["na_h128_unidir", "na_h64_bidir", "bn_h256_bidir", "bn_h64_unidir", "na_h64_poolall_max", "bn_h64_poolall_mean",
        "na_h501_unidir", "bn_h501_bidir",   # the reference's default width (dvae/train.py:55)
        "na_h64_self_attn_h", "bn_h128_self_attn_h",   # agg='self_attn_h' (dvae/dagnn.py:49-54)
        "na_h64_gated_sum", "na_h64_add", "na_h64_max", "bn_h64_add", "bn_h64_max"]   # dvae/dagnn.py:60-70


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, {k: z[k] for k in z.files if k != "meta"}


def code2_model(meta):
    H = meta["H"]
    enc = ASTNodeEncoder(H, 98, meta["n_attr"], 20)
    model = DAGNN(num_vocab=meta["V"], max_seq_len=meta["S"], emb_dim=H, hidden_dim=H, out_dim=None, encoder=enc,
                  **meta["ctor"]).eval()
    seeded_fill(model, meta["w_seed"])
    return model


def code2_batch(arr, device="cpu"):
    t = lambda k, dt=None: torch.from_numpy(arr[k].copy()).to(device)  # noqa: E731
    N = arr["x"].shape[0]
    ids = torch.arange(N, device=device)
    return SimpleNamespace(x=t("x"), node_depth=t("node_depth"), edge_index=t("edge_index"), edge_attr=t("edge_attr"),
                           batch=t("batch"), _bi_layer_idx0=t("layer0"), _bi_layer_index0=ids,
                           _bi_layer_idx1=t("layer1"), _bi_layer_index1=ids.clone(),
                           num_graphs=int(arr["batch"].max()) + 1)


def dvae_graphs(meta, arr):
    """The fixture's graphs, decoded from its stored rows (ENAS / BN encodings) by our own decoders."""
    import json as _json
    from dagnn_amd import synth
    rows = [_json.loads(r) for r in arr["rows"]]
    return [(synth.decode_enas_row if meta["kind"] == "na" else synth.decode_bn_row)(r) for r in rows]


def dvae_model(meta):
    cls, nn_ = (DAGNN_NA, 8) if meta["kind"] == "na" else (DAGNN_BN, 10)
    hs = meta["hs"]
    model = cls(nn_, hs, hs, nn_, nn_, 0, 1, hs=hs, nz=56, num_nodes=nn_, agg=meta.get("agg", "attn_h"), num_layers=meta["L"],
                bidirectional=meta["bidir"], out_wx=False, out_pool_all=meta.get("out_pool_all", False),
                out_pool=meta.get("out_pool", "max"), dropout=0.0).eval()
    seeded_fill(model, meta["w_seed"])
    return model, nn_


def dvae_batch(arr, device="cpu"):
    t = lambda k: torch.from_numpy(arr[k].copy()).to(device)  # noqa: E731
    return dagnn_amd.GraphBatch(x=t("x"), edge_index=t("edge_index"), bi_layer_index=t("bi_layer_index"),
                                batch=t("batch"))


def maxdiff(a, b):
    a = a.detach().cpu().double() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    return float((a - b).abs().max()) if a.numel() else 0.0


def grad_view(meta, name, g):
    """The rows of a full gradient that a gradient fixture stores (tests/golden/make_golden.sample_grad)."""
    return g[::meta["grad_stride"][name]]


def check_grads(meta, arr, grads, rtol=2e-4, atol=2e-7, verbose=False):
    """Compare {name: gradient} with a gradient fixture: stored rows and the float64 sum of every parameter,
    relative to the largest entry of that gradient (+ `atol`: the gradients of the attention query weights,
    attention bias and edge-encoder bias are mathematically zero - they cancel inside the segment softmax - and
    come out of the reference's autograd as ~1e-9 rounding noise).  Returns the worst relative error."""
    worst = 0.0
    for key in arr:
        if not key.startswith("g::"):
            continue
        name = key[3:]
        ref = torch.from_numpy(arr[key]).double()
        got = grad_view(meta, name, grads[name].detach().cpu()).double()
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        scale = float(ref.abs().max())
        aerr = float((got - ref).abs().max())
        abs_sum = float(arr["gsum::" + name][1])
        serr = abs(float(grads[name].detach().cpu().double().sum()) - float(arr["gsum::" + name][0]))
        if verbose:
            print("%-44s max|g| %.3e  err %.3e  sum err %.3e / %.3e" % (name, scale, aerr, serr, abs_sum))
        assert aerr <= rtol * scale + atol, "%s: max abs err %.3g at scale %.3g" % (name, aerr, scale)
        assert serr <= rtol * abs_sum + atol * got.numel() ** 0.5 * 10, "%s: sum err %.3g of %.3g" % (name, serr, abs_sum)
        worst = max(worst, aerr / max(scale, 1e-30) if scale > 100 * atol else 0.0)
    return worst


# ------------------------------------------------------------------ decoder-side single-vertex step
class VertexGraph(object):
    """The igraph surface `_ipropagate_to` touches: `vcount()`, `predecessors(v)` (ascending), `vs[x][attr]`."""

    def __init__(self, n):
        self.vs = [dict() for _ in range(n)]
        self._pred = [[] for _ in range(n)]

    def vcount(self):
        return len(self.vs)

    def predecessors(self, v):
        return sorted(self._pred[v])


def iprop_graphs(meta, arr, device):
    gs = []
    for k in range(meta["K"]):
        g = VertexGraph(int(arr["counts"][k]))
        for v in range(g.vcount()):
            g.vs[v]["type"] = int(arr["types"][k, v])
            for l in range(meta["L"]):
                g.vs[v]["H_forward%d" % l] = torch.from_numpy(arr["states"][k, v, l][None].copy()).to(device)
            g._pred[v] = [u for u in range(v) if arr["adj"][k, u, v]]
        gs.append(g)
    return gs


def check_ipropagate(name, step, device, tol):
    """`step(model, G, v, H=None)` against the `iprop_*` fixture generated from the reference's `_ipropagate_to`."""
    meta, arr = load(name)
    meta = dict(meta, bidir=False)
    model, nvt = dvae_model(meta)
    model = model.to(device)
    K, n, L = meta["K"], meta["n"], meta["L"]
    with torch.no_grad():
        for v in meta["vs"]:
            G = iprop_graphs(meta, arr, device)
            Hv = step(model, G, v)
            alive = [k for k in range(K) if arr["counts"][k] > v]
            assert alive == list(arr["v%d_alive" % v])
            assert maxdiff(Hv, arr["v%d_Hv" % v]) < tol
            got = np.stack([np.stack([G[k].vs[v]["H_forward%d" % l][0].cpu().numpy() for l in range(L)]) for k in alive])
            assert np.abs(got - arr["v%d_states" % v]).max() < tol
            Hg = step(model, iprop_graphs(meta, arr, device), v, H=torch.from_numpy(arr["H_given"].copy()).to(device))
            assert maxdiff(Hg, arr["v%d_Hv_given" % v]) < tol
        assert step(model, iprop_graphs(meta, arr, device), n + 3) is None   # no graph has that vertex
