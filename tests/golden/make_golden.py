#!/usr/bin/env python
"""Generate the golden fixtures in this directory from the REAL reference.

Runs only in the build container (needs `/root/reference`).  It imports the reference's model
files *unmodified* (`ogbg-code/model/dagnn.py`, `ogbg-code/utils.py`, `dvae/dagnn.py`,
`dvae/dagnn_bn.py`, `dvae/util.py`, `dvae/batch.py`, `src/utils_dag.py`) on top of our PyG
stand-in (`oracle/pyg_standin`, see its README), runs them on seeded inputs and writes inputs +
expected outputs as small `.npz` files.  Weights are not stored: `oracle/seeding.seeded_fill`
regenerates them from the seed on both sides.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The fixtures are data (inputs and expected outputs); no reference source travels.
"""
from __future__ import annotations

import copy
import importlib
import importlib.util
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("DAGNN_REFERENCE", "/root/reference")

sys.path.insert(0, REPO)
from oracle.seeding import seeded_fill  # noqa: E402
from dagnn_amd import synth  # noqa: E402


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _setup_paths():
    sys.argv = sys.argv[:1]  # dvae/util.py parses argv at import
    for p in (os.path.join(REF, "dvae"), REF, os.path.join(REPO, "oracle", "pyg_standin")):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)


def _np(t):
    return t.detach().cpu().numpy()


def _save(name, meta, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------- ogbg-code
def make_code2(ref_dagnn, ref_utils, ref_dagutils, name, *, data_seed, B, mean_n, H, L, bidir,
               V, S, n_attr, w_seed, row_stride, max_n=1000, **ctor):
    from types import SimpleNamespace
    graphs = synth.code2_graphs(data_seed, B, mean_n, max_n)
    # layer ids from the REFERENCE's top_sort (src/utils_dag.py:8-52), not ours
    for g in graphs:
        g.x[:, 1] %= n_attr
        ns = SimpleNamespace(edge_index=g.edge_index, num_nodes=g.num_nodes)
        ref_dagutils.add_order_info_01(ns)
        for k in ("_bi_layer_idx0", "_bi_layer_index0", "_bi_layer_idx1", "_bi_layer_index1"):
            setattr(g, k, getattr(ns, k))
    b = synth.GraphBatch.from_data_list(graphs)
    enc = ref_utils.ASTNodeEncoder(H, 98, n_attr, 20)
    kw = dict(w_edge_attr=True, num_layers=L, bidirectional=bidir, agg="attn_h", out_wx=False,
              out_pool_all=False, out_pool="max", dropout=0.0)
    kw.update(ctor)
    model = ref_dagnn.DAGNN(num_vocab=V, max_seq_len=S, emb_dim=H, hidden_dim=H, out_dim=None,
                            encoder=enc, **kw).eval()
    seeded_fill(model, w_seed)
    G = SimpleNamespace(x=b.x.clone(), node_depth=b.node_depth.clone(), edge_index=b.edge_index.clone(),
                        edge_attr=b.edge_attr.clone(), batch=b.batch.clone(),
                        _bi_layer_idx0=b._bi_layer_idx0.clone(), _bi_layer_index0=b._bi_layer_index0.clone(),
                        _bi_layer_idx1=b._bi_layer_idx1.clone(), _bi_layer_index1=b._bi_layer_index1.clone())
    with torch.no_grad():
        out = model(G)
    out = out if isinstance(out, (list, tuple)) else [out]
    N = b.x.shape[0]
    rows = np.arange(0, N, row_stride)
    arrays = dict(
        x=_np(b.x), node_depth=_np(b.node_depth), edge_index=_np(b.edge_index), edge_attr=_np(b.edge_attr),
        batch=_np(b.batch), layer0=_np(b._bi_layer_idx0), layer1=_np(b._bi_layer_idx1),
        pred=np.stack([_np(o) for o in out]), rows=rows, x_emb=_np(G.x)[rows],
        node_depth_after=_np(G.node_depth),
    )
    if isinstance(G.h, list):
        for d, hd in enumerate(G.h):
            for i, h in enumerate(hd):
                arrays["h_%d_%d" % (d, i)] = _np(h)[rows]
    else:
        arrays["h_cat"] = _np(G.h)
        arrays["batch_after"] = _np(G.batch)
    meta = dict(kind="code2", data_seed=data_seed, B=B, mean_n=mean_n, max_n=max_n, H=H, L=L, bidir=bool(bidir),
                V=V, S=S, n_attr=n_attr, w_seed=w_seed, ctor=kw, N=int(N), E=int(b.edge_index.shape[1]),
                T=int(b._bi_layer_idx0.max()) + 1,
                state_dict={k: list(v.shape) for k, v in model.state_dict().items()})
    _save(name, meta, **arrays)


def sample_grad(name, g):
    """What a gradient fixture keeps of one parameter gradient: everything when it is small, else every
    `stride`-th row (tests/helpers.grad_view applies the same rule) - plus float64 sum / abs-sum of the full array."""
    g = np.asarray(g)
    stride = 1 if g.size <= 30000 or g.ndim < 2 else -(-g.shape[0] // 48)
    return g[::stride], stride, np.array([g.astype(np.float64).sum(), np.abs(g.astype(np.float64)).sum()])


def make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, name, *, data_seed, B, mean_n, H, L, V, S, n_attr, w_seed,
                    y_seed, max_n=1000, **ctor):
    """One training step's gradients from the reference: `DAGNN.forward` under autograd, the loss of
    ogbg-code/main_pyg.py:55-62 (mean over the S heads of CrossEntropy against y_arr[:, s]), `.backward()`."""
    from types import SimpleNamespace
    graphs = synth.code2_graphs(data_seed, B, mean_n, max_n)
    for g in graphs:
        g.x[:, 1] %= n_attr
        ns = SimpleNamespace(edge_index=g.edge_index, num_nodes=g.num_nodes)
        ref_dagutils.add_order_info_01(ns)
        for k in ("_bi_layer_idx0", "_bi_layer_index0", "_bi_layer_idx1", "_bi_layer_index1"):
            setattr(g, k, getattr(ns, k))
    b = synth.GraphBatch.from_data_list(graphs)
    enc = ref_utils.ASTNodeEncoder(H, 98, n_attr, 20)
    kw = dict(w_edge_attr=True, num_layers=L, bidirectional=True, agg="attn_h", out_wx=False,
              out_pool_all=False, out_pool="max", dropout=0.0)
    kw.update(ctor)
    model = ref_dagnn.DAGNN(num_vocab=V, max_seq_len=S, emb_dim=H, hidden_dim=H, out_dim=None,
                            encoder=enc, **kw).train()
    seeded_fill(model, w_seed)
    y = torch.from_numpy(np.random.default_rng(y_seed).integers(0, V, size=(B, S)))
    G = SimpleNamespace(x=b.x.clone(), node_depth=b.node_depth.clone(), edge_index=b.edge_index.clone(),
                        edge_attr=b.edge_attr.clone(), batch=b.batch.clone(),
                        _bi_layer_idx0=b._bi_layer_idx0.clone(), _bi_layer_index0=b._bi_layer_index0.clone(),
                        _bi_layer_idx1=b._bi_layer_idx1.clone(), _bi_layer_index1=b._bi_layer_index1.clone())
    pred = model(G)
    ce = torch.nn.CrossEntropyLoss()
    loss = 0
    for s_ in range(len(pred)):
        loss = loss + ce(pred[s_].to(torch.float32), y[:, s_])
    loss = loss / len(pred)
    loss.backward()
    arrays = dict(
        x=_np(b.x), node_depth=_np(b.node_depth), edge_index=_np(b.edge_index), edge_attr=_np(b.edge_attr),
        batch=_np(b.batch), layer0=_np(b._bi_layer_idx0), layer1=_np(b._bi_layer_idx1), y=_np(y),
        loss=np.array(float(loss.detach())), pred=np.stack([_np(o) for o in pred]))
    strides = {}
    for k, p_ in model.named_parameters():
        g = np.zeros(tuple(p_.shape), np.float32) if p_.grad is None else _np(p_.grad)
        arrays["g::" + k], strides[k], arrays["gsum::" + k] = sample_grad(k, g)
    meta = dict(kind="code2_grad", data_seed=data_seed, B=B, mean_n=mean_n, max_n=max_n, H=H, L=L,
                bidir=bool(kw["bidirectional"]),
                V=V, S=S, n_attr=n_attr, w_seed=w_seed, y_seed=y_seed, ctor=kw, N=int(b.x.shape[0]),
                E=int(b.edge_index.shape[1]), T=int(b._bi_layer_idx0.max()) + 1, grad_stride=strides,
                state_dict={k: list(v.shape) for k, v in model.state_dict().items()})
    _save(name, meta, **arrays)


# ------------------------------------------------------------------------------- dvae
def _dvae_case(model, graphs, ref_batch_mod):
    data = [copy.deepcopy(g) for g in graphs]  # models_pyg.py:114-115 (collation mutates)
    b = ref_batch_mod.Batch.from_data_list(data)
    b.batch_before = b.batch.clone()  # forward overwrites G.batch with the read-out rows' ids
    with torch.no_grad():
        Hg = model(b)
        mu, logvar = model.fc1(Hg), model.fc2(Hg)
        # encode() itself must agree (dvae/dagnn.py:177-184)
        mu2, lv2 = model.encode([copy.deepcopy(g) for g in graphs])
    assert torch.equal(mu, mu2) and torch.equal(logvar, lv2)
    return b, Hg, mu, logvar


def make_na(ref_na, ref_util, ref_batch_mod, name, *, hs, L, bidir, w_seed, nrows=64, out_pool_all=False,
            out_pool="max", agg="attn_h"):
    rows = []
    with open(os.path.join(REF, "dvae", "data", "final_structures6.txt")) as f:
        for i, line in enumerate(f):
            if i < 1000:  # burn_in of load_ENAS_graphs (dvae/util.py:67)
                continue
            rows.append(eval(line)[0])
            if len(rows) == nrows:
                break
    graphs = [ref_util.decode_ENAS_to_pygraph(r)[0] for r in rows]
    model = ref_na.DAGNN(8, hs, hs, 8, 8, 0, 1, hs=hs, nz=56, num_nodes=8, agg=agg, num_layers=L,
                         bidirectional=bidir, out_wx=False, out_pool_all=out_pool_all, out_pool=out_pool,
                         dropout=0.0).eval()
    seeded_fill(model, w_seed)
    b, Hg, mu, logvar = _dvae_case(model, graphs, ref_batch_mod)
    meta = dict(kind="na", hs=hs, L=L, bidir=bool(bidir), w_seed=w_seed, nrows=nrows, out_pool_all=out_pool_all,
                out_pool=out_pool, agg=agg,
                state_dict={k: list(v.shape) for k, v in model.state_dict().items()})
    _save(name, meta, rows=np.array([json.dumps(r) for r in rows]),
          x=_np(b.x), edge_index=_np(b.edge_index), bi_layer_index=_np(b.bi_layer_index), batch=_np(b.batch_before), batch_after=_np(b.batch),
          Hg=_np(Hg), mu=_np(mu), logvar=_np(logvar))


def make_bn(ref_bn, ref_util, ref_batch_mod, name, *, hs, L, bidir, w_seed, data_seed, nrows, out_pool_all=False,
            out_pool="max", agg="attn_h"):
    rows = synth.bn_rows(data_seed, nrows)
    graphs = [ref_util.decode_BN_to_pygraph(r)[0] for r in rows]
    model = ref_bn.DAGNN_BN(10, hs, hs, 10, 10, 0, 1, hs=hs, nz=56, num_nodes=10, agg=agg, num_layers=L,
                            bidirectional=bidir, out_wx=False, out_pool_all=out_pool_all, out_pool=out_pool,
                            dropout=0.0).eval()
    seeded_fill(model, w_seed)
    b, Hg, mu, logvar = _dvae_case(model, graphs, ref_batch_mod)
    meta = dict(kind="bn", hs=hs, L=L, bidir=bool(bidir), w_seed=w_seed, data_seed=data_seed, nrows=nrows,
                out_pool_all=out_pool_all, out_pool=out_pool, agg=agg,
                state_dict={k: list(v.shape) for k, v in model.state_dict().items()})
    _save(name, meta, rows=np.array([json.dumps(r) for r in rows]),
          x=_np(b.x), edge_index=_np(b.edge_index), bi_layer_index=_np(b.bi_layer_index), batch=_np(b.batch_before), batch_after=_np(b.batch),
          Hg=_np(Hg), mu=_np(mu), logvar=_np(logvar))


# ------------------------------------------------------------------------------- D-VAE decoder-side single-vertex step
def make_ipropagate(ref_mod, cls_name, name, *, kind, hs, L, w_seed, data_seed, K, n):
    """`_ipropagate_to(G, v, propagator)` of the reference (`dvae/dagnn.py:187-239`, `dvae/dagnn_bn.py:179-238`) on
    K stand-in igraph graphs of up to n vertices: the new states at vertex v from the states of its predecessors
    (dense padded attention over the predecessor lists), for two vertices, without and with a given H."""
    import igraph
    nvt = 8 if kind == "na" else 10
    model = getattr(ref_mod, cls_name)(nvt, hs, hs, nvt, nvt, 0, 1, hs=hs, nz=56, num_nodes=nvt, agg="attn_h",
                                       num_layers=L, bidirectional=False, out_wx=False, out_pool_all=False,
                                       out_pool="max", dropout=0.0).eval()
    seeded_fill(model, w_seed)
    rng = np.random.default_rng(data_seed)
    counts = rng.integers(max(3, n - 3), n + 1, size=K)
    types = np.full((K, n), -1, dtype=np.int64)
    adj = np.zeros((K, n, n), dtype=np.int64)      # adj[k, u, v] = 1: edge u -> v (u < v)
    states = rng.standard_normal((K, n, L, hs)).astype(np.float32) * 0.5
    for k in range(K):
        types[k, :counts[k]] = rng.integers(0, nvt, size=counts[k])
        for v in range(1, counts[k]):
            for u in range(v):
                adj[k, u, v] = int(rng.random() < (0.9 if u == v - 1 else 0.35))
    adj[0, :, 2] = 0                               # a vertex without predecessors among the cases

    def graphs():
        gs = []
        for k in range(K):
            g = igraph.Graph(directed=True)
            g.add_vertices(int(counts[k]))
            for v in range(counts[k]):
                g.vs[v]["type"] = int(types[k, v])
                for l in range(L):
                    g.vs[v]["H_forward%d" % l] = torch.from_numpy(states[k, v, l][None].copy())
                for u in range(v):
                    if adj[k, u, v]:
                        g.add_edge(u, v)
            gs.append(g)
        return gs

    out = {}
    H_given = torch.from_numpy(rng.standard_normal((K, hs)).astype(np.float32) * 0.5)
    with torch.no_grad():
        for v in (2, n - 2):
            G = graphs()
            Hv = model._ipropagate_to(G, v, model.grud)
            alive = [k for k in range(K) if counts[k] > v]
            out["v%d_alive" % v] = np.array(alive)
            out["v%d_Hv" % v] = _np(Hv)
            out["v%d_states" % v] = np.stack([np.stack([_np(G[k].vs[v]["H_forward%d" % l])[0] for l in range(L)])
                                              for k in alive])
            G = graphs()
            out["v%d_Hv_given" % v] = _np(model._ipropagate_to(G, v, model.grud, H=H_given.clone()))
    meta = dict(kind=kind, hs=hs, L=L, w_seed=w_seed, K=K, n=n, vs=[2, n - 2],
                state_dict={k: list(v.shape) for k, v in model.state_dict().items()})
    _save(name, meta, counts=counts, types=types, adj=adj, states=states, H_given=_np(H_given), **out)



def make_augment(name):
    """`augment_edge2` of the reference (ogbg-code/utils2.py:30-78) on seeded ASTs: inputs and outputs."""
    from types import SimpleNamespace
    ref_utils2 = _load_file("ref_ogbg_utils2", os.path.join(REF, "ogbg-code", "utils2.py"))
    rng = np.random.default_rng(41)
    arrays = {}
    for k, n in enumerate((12, 40, 3, 90)):
        g = synth.gen_ast(rng, n)
        parent_edges = g["ei"][:, g["ea"][:, 0] == 0]          # the AST edges only
        attributed = np.zeros(n, dtype=np.int64)
        attributed[np.setdiff1d(np.arange(n), parent_edges[0])] = 1   # leaves carry the tokens
        d = SimpleNamespace(edge_index=torch.from_numpy(parent_edges).long(),
                            node_is_attributed=torch.from_numpy(attributed).view(-1, 1))
        out = ref_utils2.augment_edge2(d)
        arrays["in_edge_index_%d" % k] = parent_edges
        arrays["in_attributed_%d" % k] = attributed
        arrays["out_edge_index_%d" % k] = _np(out.edge_index)
        arrays["out_edge_attr_%d" % k] = _np(out.edge_attr)
    _save(name, dict(kind="augment_edge2", cases=4), **arrays)


def _dvae_grad_case(model, graphs, name, meta, seed):
    """Gradients of the reference encoder: `(mu, logvar) = model.encode(graphs)` (dvae/dagnn.py:177-184) under
    autograd, loss = <mu, R1> + <logvar, R2> with seeded weights R (a stand-in for the VAE loss of dvae/train.py, whose
    decoder is outside the path), `.backward()`.  Only parameters that received a gradient are stored."""
    model.train()
    mu, logvar = model.encode([copy.deepcopy(g) for g in graphs])
    rng = np.random.default_rng(seed)
    r1 = torch.from_numpy(rng.standard_normal(tuple(mu.shape)).astype(np.float32))
    r2 = torch.from_numpy(rng.standard_normal(tuple(mu.shape)).astype(np.float32))
    loss = (mu * r1).sum() + (logvar * r2).sum()
    loss.backward()
    arrays = dict(r1=_np(r1), r2=_np(r2), loss=np.array(float(loss.detach())), mu=_np(mu), logvar=_np(logvar))
    strides = {}
    for k, p_ in model.named_parameters():
        if p_.grad is None:
            continue
        arrays["g::" + k], strides[k], arrays["gsum::" + k] = sample_grad(k, _np(p_.grad))
    meta = dict(meta, grad_stride=strides, loss_seed=seed)
    return meta, arrays


def make_na_grad(ref_na, ref_util, name, *, hs, L, bidir, w_seed, nrows, agg="attn_h"):
    rows = []
    with open(os.path.join(REF, "dvae", "data", "final_structures6.txt")) as f:
        for i, line in enumerate(f):
            if i < 1000:
                continue
            rows.append(eval(line)[0])
            if len(rows) == nrows:
                break
    graphs = [ref_util.decode_ENAS_to_pygraph(r)[0] for r in rows]
    model = ref_na.DAGNN(8, hs, hs, 8, 8, 0, 1, hs=hs, nz=56, num_nodes=8, agg=agg, num_layers=L,
                         bidirectional=bidir, out_wx=False, out_pool_all=False, out_pool="max", dropout=0.0)
    seeded_fill(model, w_seed)
    meta, arrays = _dvae_grad_case(model, graphs, name, dict(kind="na", hs=hs, L=L, bidir=bool(bidir), w_seed=w_seed,
                                                                nrows=nrows, agg=agg), 401)
    _save(name, meta, rows=np.array([json.dumps(r) for r in rows]), **arrays)


def make_bn_grad(ref_bn, ref_util, name, *, hs, L, bidir, w_seed, data_seed, nrows, agg="attn_h"):
    rows = synth.bn_rows(data_seed, nrows)
    graphs = [ref_util.decode_BN_to_pygraph(r)[0] for r in rows]
    model = ref_bn.DAGNN_BN(10, hs, hs, 10, 10, 0, 1, hs=hs, nz=56, num_nodes=10, agg=agg, num_layers=L,
                            bidirectional=bidir, out_wx=False, out_pool_all=False, out_pool="max", dropout=0.0)
    seeded_fill(model, w_seed)
    meta, arrays = _dvae_grad_case(model, graphs, name, dict(kind="bn", hs=hs, L=L, bidir=bool(bidir), w_seed=w_seed,
                                                                data_seed=data_seed, nrows=nrows, agg=agg), 402)
    _save(name, meta, rows=np.array([json.dumps(r) for r in rows]), **arrays)


VARIANTS = (("gated_sum", 31, dict(agg="gated_sum")), ("gated_nobias", 32, dict(agg="gated_sum", mapper_bias=False)),
            ("mattn_h", 33, dict(agg="mattn_h")), ("add", 34, dict(agg="add")), ("max", 35, dict(agg="max")),
            ("aggx_attn_h", 36, dict(agg="attn_h", agg_x=True)), ("aggx_add", 37, dict(agg="add", agg_x=True)),
            ("recurr0", 38, dict(agg="attn_h", recurr=0)), ("recurr0_gated", 39, dict(agg="gated_sum", recurr=0)))


def _variants_only(ref_dagnn, ref_utils, ref_dagutils, common):
    for tag, seed, extra in VARIANTS:
        make_code2(ref_dagnn, ref_utils, ref_dagutils, "var_h64_" + tag, data_seed=seed, B=4, mean_n=25, H=64, L=2,
                   bidir=1, w_seed=100 + seed, row_stride=2, **extra, **common)


def _ipropagate_only():
    importlib.import_module("util")
    ref_na = importlib.import_module("dagnn")
    ref_bn = importlib.import_module("dagnn_bn")
    make_ipropagate(ref_na, "DAGNN", "iprop_na_h64_L2", kind="na", hs=64, L=2, w_seed=221, data_seed=31, K=6, n=8)
    make_ipropagate(ref_bn, "DAGNN_BN", "iprop_bn_h32_L3", kind="bn", hs=32, L=3, w_seed=222, data_seed=32, K=5, n=10)


def _dvae_default_hs():
    """The reference's own default width (`dvae/train.py:55`: --hs 501): not a multiple of anything the kernels tile by."""
    ref_util = importlib.import_module("util")
    ref_batch_mod = importlib.import_module("batch")
    ref_na = importlib.import_module("dagnn")
    ref_bn = importlib.import_module("dagnn_bn")
    make_na(ref_na, ref_util, ref_batch_mod, "na_h501_unidir", hs=501, L=2, bidir=False, w_seed=231, nrows=16)
    make_bn(ref_bn, ref_util, ref_batch_mod, "bn_h501_bidir", hs=501, L=2, bidir=True, w_seed=232, data_seed=9, nrows=12)
    make_na_grad(ref_na, ref_util, "grad_na_h501_unidir", hs=501, L=2, bidir=False, w_seed=233, nrows=12)
    make_bn_grad(ref_bn, ref_util, "grad_bn_h501_bidir", hs=501, L=2, bidir=True, w_seed=234, data_seed=10, nrows=10)


def _dvae_self_attn():
    """The D-VAE encoders with `agg='self_attn_h'` (`dvae/dagnn.py:49-54`: keys scored alone, no query half)."""
    ref_util = importlib.import_module("util")
    ref_batch_mod = importlib.import_module("batch")
    ref_na = importlib.import_module("dagnn")
    ref_bn = importlib.import_module("dagnn_bn")
    make_na(ref_na, ref_util, ref_batch_mod, "na_h64_self_attn_h", hs=64, L=2, bidir=True, w_seed=241, nrows=16, agg="self_attn_h")
    make_bn(ref_bn, ref_util, ref_batch_mod, "bn_h128_self_attn_h", hs=128, L=2, bidir=True, w_seed=242, data_seed=11, nrows=12,
            agg="self_attn_h")
    make_na_grad(ref_na, ref_util, "grad_na_h64_self_attn_h", hs=64, L=2, bidir=False, w_seed=243, nrows=16, agg="self_attn_h")
    make_bn_grad(ref_bn, ref_util, "grad_bn_h64_self_attn_h", hs=64, L=2, bidir=True, w_seed=244, data_seed=12, nrows=12,
                 agg="self_attn_h")


def _dvae_aggs():
    """The D-VAE encoders with the non-attention aggregators (`dvae/dagnn.py:60-70`, `dvae/dagnn_bn.py` same block):
    `gated_sum` (mapper / gate of the base class on [state ; vertex id] for NA, on the state for BN), `add`, `max`."""
    ref_util = importlib.import_module("util")
    ref_batch_mod = importlib.import_module("batch")
    ref_na = importlib.import_module("dagnn")
    ref_bn = importlib.import_module("dagnn_bn")
    k = 0
    for agg in ("gated_sum", "add", "max"):
        make_na(ref_na, ref_util, ref_batch_mod, "na_h64_%s" % agg, hs=64, L=2, bidir=(agg != "add"), w_seed=251 + k, nrows=16, agg=agg)
        make_na_grad(ref_na, ref_util, "grad_na_h64_%s" % agg, hs=64, L=2, bidir=(agg == "max"), w_seed=257 + k, nrows=12, agg=agg)
        if agg != "gated_sum":   # (the reference's own DAGNN_BN cannot run gated_sum: DVAE_BN_PYG re-creates the first layer's mapper /
            # gate with nvt inputs, models_pyg.py:539-560, and GatedSumConv feeds them hs-wide states - a shape error at the first layer)
            make_bn(ref_bn, ref_util, ref_batch_mod, "bn_h64_%s" % agg, hs=64, L=2, bidir=True, w_seed=254 + k, data_seed=13 + k,
                    nrows=12, agg=agg)
            make_bn_grad(ref_bn, ref_util, "grad_bn_h64_%s" % agg, hs=64, L=2, bidir=True, w_seed=260 + k, data_seed=16 + k,
                         nrows=10, agg=agg)
        k += 1


def _dvae_only():
    ref_util = importlib.import_module("util")
    ref_na = importlib.import_module("dagnn")
    ref_bn = importlib.import_module("dagnn_bn")
    make_na_grad(ref_na, ref_util, "grad_na_h64_unidir", hs=64, L=2, bidir=False, w_seed=211, nrows=24)
    make_bn_grad(ref_bn, ref_util, "grad_bn_h64_bidir", hs=64, L=2, bidir=True, w_seed=212, data_seed=7, nrows=20)


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not found at %s - fixtures can only be regenerated in the build container" % REF)
    _setup_paths()
    torch.manual_seed(0)
    ref_dagutils = importlib.import_module("src.utils_dag")
    ref_dagnn = _load_file("ref_ogbg_dagnn", os.path.join(REF, "ogbg-code", "model", "dagnn.py"))
    ref_utils = _load_file("ref_ogbg_utils", os.path.join(REF, "ogbg-code", "utils.py"))

    common = dict(V=48, S=5, n_attr=300)
    only = os.environ.get("GOLDEN_ONLY")  # "grad" / "augment": rewrite only those fixtures
    if only in (None, "augment"):
        make_augment("augment_edge2")
    if only == "augment":
        return
    if only == "dvae_grad":
        return _dvae_only()
    if only == "dvae_aggs":
        return _dvae_aggs()
    if only == "ipropagate":
        return _ipropagate_only()
    if only == "dvae_default_hs":
        return _dvae_default_hs()
    if only == "dvae_self_attn":
        return _dvae_self_attn()
    if only in (None, "grad", "grad_h300"):
        # the reference's own training width (scripts/ogb_tok.sh:17: --emb_dim=300): the 320-wide reverse sweep
        # (csrc/bwd_dataflow_w.hip) against the reference's loss.backward(), not only against the oracle's autograd
        make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_h300_bidir", data_seed=31, B=6, mean_n=40, H=300,
                        L=2, w_seed=131, y_seed=331, **common)
    if only == "grad_h300":
        return
    # training-step gradients: loss and parameter gradients of one step
    if True:
        make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_h32_bidir", data_seed=11, B=6, mean_n=30, H=32,
                        L=2, w_seed=101, y_seed=301, **common)
        make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_h256_bidir", data_seed=12, B=8, mean_n=60, H=256,
                        L=2, w_seed=102, y_seed=302, **common)
        make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_h128_deep", data_seed=17, B=3, mean_n=400, H=128,
                        L=2, w_seed=107, y_seed=303, **common)
        make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_h64_L3_wx", data_seed=21, B=5, mean_n=40, H=64,
                        L=3, w_seed=121, y_seed=304, out_wx=True, **common)
        # read-outs that go through torch ops on differentiable states: unidirectional (dagnn.py:195-202 branch),
        # mean pooling over all nodes of both directions
        make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_h64_unidir", data_seed=22, B=6, mean_n=35, H=64,
                        L=2, w_seed=122, y_seed=305, bidirectional=False, **common)
        make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_h64_mean_all", data_seed=23, B=6, mean_n=35, H=64,
                        L=2, w_seed=123, y_seed=306, out_pool_all=True, out_pool="mean", **common)
    if only == "grad":
        return
    if only in (None, "grad_var"):
        # training-step gradients of the constructor-string variants that train through HIP (SURVEY §8 f3:
        # csrc/variants_bwd.hip): GatedSumConv with / without mapper bias, MultAttnConv, AggConv add / max, the Linear cell
        for tag, seed, extra in (("gated_sum", 51, dict(agg="gated_sum")),
                                 ("gated_nobias", 52, dict(agg="gated_sum", mapper_bias=False)),
                                 ("mattn_h", 53, dict(agg="mattn_h")), ("add", 54, dict(agg="add")),
                                 ("mattn_h_L3", 55, dict(agg="mattn_h", num_layers=3)), ("max", 56, dict(agg="max")),
                                 ("recurr0_gated", 57, dict(agg="gated_sum", recurr=0)),
                                 ("recurr0_mattn", 58, dict(agg="mattn_h", recurr=0)),
                                 ("recurr0_attn_h", 59, dict(agg="attn_h", recurr=0)),
                                 ("recurr0_attn_x", 60, dict(agg="attn_x", recurr=0)),
                                 ("recurr0_self_attn_h", 61, dict(agg="self_attn_h", recurr=0)),
                                 ("aggx_attn_h", 62, dict(agg="attn_h", agg_x=True)),
                                 ("aggx_add", 63, dict(agg="add", agg_x=True)),
                                 ("aggx_gated", 64, dict(agg="gated_sum", agg_x=True)),
                                 ("aggx_mattn", 65, dict(agg="mattn_h", agg_x=True)),
                                 ("aggx_max_recurr0", 66, dict(agg="max", agg_x=True, recurr=0))):
            L_ = extra.pop("num_layers", 2)
            make_code2_grad(ref_dagnn, ref_utils, ref_dagutils, "grad_var_h64_" + tag, data_seed=seed, B=5, mean_n=30,
                            H=64, L=L_, w_seed=150 + seed, y_seed=350 + seed, **extra, **common)
    if only == "grad_var":
        return

    if only == "variants":
        return _variants_only(ref_dagnn, ref_utils, ref_dagutils, common)
    # tiny generic-H case, every hidden row stored
    make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h32_bidir", data_seed=11, B=6, mean_n=30, H=32, L=2,
               bidir=1, w_seed=101, row_stride=1, **common)
    # the headline shape (h=256, L=2, bidirectional) on a small batch
    make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h256_bidir", data_seed=12, B=8, mean_n=60, H=256, L=2,
               bidir=1, w_seed=102, row_stride=5, **common)
    # cfg-5 shaped (h=512, L=5), tiny
    make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h512_L5", data_seed=13, B=4, mean_n=25, H=512, L=5,
               bidir=1, w_seed=103, row_stride=4, **common)
    # emb 300 as in scripts/ogb_tok.sh (H not a power of two), 3 layers
    make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h300_L3", data_seed=14, B=5, mean_n=40,
               H=300, L=3, bidir=1, w_seed=104, row_stride=3, **common)
    # unidirectional + pooled over output nodes (dagnn.py:195-202 branch)
    make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h64_unidir", data_seed=15, B=7, mean_n=35, H=64, L=2,
               bidir=0, w_seed=105, row_stride=1, **common)
    # LP-style single classification head (num_class>0, dagnn.py:103-104,209-210)
    make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h64_numclass", data_seed=16, B=5, mean_n=30, H=64, L=2,
               bidir=1, w_seed=106, row_stride=1, V=48, S=5, n_attr=300, num_class=17)
    # the other additive-attention aggregators (src/constants.py:91-94; keys from x / no query)
    for agg, seed in (("attn_x", 18), ("self_attn_h", 19), ("self_attn_x", 20)):
        make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h64_" + agg, data_seed=seed, B=5, mean_n=30, H=64,
                   L=2, bidir=1, w_seed=100 + seed, row_stride=1, agg=agg, **common)
    # the constructor strings outside every BASELINE configuration: other aggregators, agg_x, recurr=0
    _variants_only(ref_dagnn, ref_utils, ref_dagutils, common)
    # deep chain stress: long graphs (depth ~ 200) to exercise many recurrent steps
    make_code2(ref_dagnn, ref_utils, ref_dagutils, "code2_h128_deep", data_seed=17, B=3, mean_n=400, H=128, L=2,
               bidir=1, w_seed=107, row_stride=7, **common)

    ref_util = importlib.import_module("util")
    ref_batch_mod = importlib.import_module("batch")
    ref_na = importlib.import_module("dagnn")
    ref_bn = importlib.import_module("dagnn_bn")
    if only in (None, "dvae_grad"):
        make_na_grad(ref_na, ref_util, "grad_na_h64_unidir", hs=64, L=2, bidir=False, w_seed=211, nrows=24)
        make_bn_grad(ref_bn, ref_util, "grad_bn_h64_bidir", hs=64, L=2, bidir=True, w_seed=212, data_seed=7, nrows=20)
    if only == "dvae_grad":
        return
    make_na(ref_na, ref_util, ref_batch_mod, "na_h128_unidir", hs=128, L=2, bidir=False, w_seed=201)
    make_na(ref_na, ref_util, ref_batch_mod, "na_h64_bidir", hs=64, L=2, bidir=True, w_seed=202, nrows=16)
    make_bn(ref_bn, ref_util, ref_batch_mod, "bn_h256_bidir", hs=256, L=2, bidir=True, w_seed=203, data_seed=5,
            nrows=32)
    make_bn(ref_bn, ref_util, ref_batch_mod, "bn_h64_unidir", hs=64, L=3, bidir=False, w_seed=204, data_seed=6,
            nrows=12)
    # pooling over all nodes (dvae/dagnn.py:163-172)
    make_na(ref_na, ref_util, ref_batch_mod, "na_h64_poolall_max", hs=64, L=2, bidir=False, w_seed=205, nrows=16,
            out_pool_all=True, out_pool="max")
    make_bn(ref_bn, ref_util, ref_batch_mod, "bn_h64_poolall_mean", hs=64, L=2, bidir=True, w_seed=206, data_seed=8,
            nrows=12, out_pool_all=True, out_pool="mean")
    _ipropagate_only()   # decoder-side single-vertex step (needs the igraph stand-in)
    _dvae_default_hs()
    _dvae_self_attn()
    _dvae_aggs()


if __name__ == "__main__":
    main()
