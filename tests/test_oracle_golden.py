"""Pins the oracle (oracle/dagnn_oracle.py) against outputs of the REAL reference
(tests/golden/*.npz, written by tests/golden/make_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from oracle import dagnn_oracle as O
from tests import helpers as Hh

TOL = 2e-5  # fp32 restatement vs fp32 reference: different summation order only


@pytest.mark.parametrize("name", Hh.CODE2 + Hh.VARIANTS)
@pytest.mark.parametrize("mode", ["csr", "faithful"])
def test_code2_oracle_matches_reference(name, mode):
    meta, arr = Hh.load(name)
    if mode == "faithful" and meta["N"] > 700:
        pytest.skip("faithful mirror is O(N*E); covered by the smaller fixtures")
    if mode == "csr" and name in Hh.VARIANTS:
        pytest.skip("the row-a12 variants have one (faithful) restatement")
    model = Hh.code2_model(meta)
    G = Hh.code2_batch(arr)
    kw = meta["ctor"]
    out = O.code2_forward(model.state_dict(), G, num_layers=kw["num_layers"], bidirectional=bool(kw["bidirectional"]),
                          out_wx=kw["out_wx"], out_pool_all=kw["out_pool_all"], out_pool=kw["out_pool"],
                          max_seq_len=meta["S"], num_class=kw.get("num_class", 0), mode=mode,
                          agg=kw.get("agg", "attn_h"), agg_x=kw.get("agg_x", False), recurr=kw.get("recurr", 1))
    out = out if isinstance(out, list) else [out]
    assert len(out) == arr["pred"].shape[0]
    for o, ref in zip(out, arr["pred"]):
        assert o.shape == ref.shape
        assert Hh.maxdiff(o, ref) < TOL
    rows = arr["rows"]
    assert Hh.maxdiff(G.x[rows], arr["x_emb"]) < 1e-6
    assert np.array_equal(G.node_depth.numpy(), arr["node_depth_after"])
    if isinstance(G.h, list):
        for d, hd in enumerate(G.h):
            for i, h in enumerate(hd):
                assert Hh.maxdiff(h[rows], arr["h_%d_%d" % (d, i)]) < TOL
    else:
        assert Hh.maxdiff(G.h, arr["h_cat"]) < TOL
        assert np.array_equal(G.batch.numpy(), arr["batch_after"])


@pytest.mark.parametrize("name", Hh.DVAE)
@pytest.mark.parametrize("mode", ["csr", "faithful"])
def test_dvae_oracle_matches_reference(name, mode):
    meta, arr = Hh.load(name)
    model, nn_ = Hh.dvae_model(meta)
    G = Hh.dvae_batch(arr)
    mu, logvar = O.dvae_encode(model.state_dict(), G, num_layers=meta["L"], bidirectional=meta["bidir"],
                               num_nodes=nn_, vids=meta["kind"] == "na", mode=mode,
                               out_pool_all=meta.get("out_pool_all", False), out_pool=meta.get("out_pool", "max"),
                               agg=meta.get("agg", "attn_h"))
    assert Hh.maxdiff(mu, arr["mu"]) < TOL
    assert Hh.maxdiff(logvar, arr["logvar"]) < TOL


def test_fp64_oracle_brackets_reference():
    """The fp64 oracle is the 'true' value: the fp32 reference must sit within fp32 round-off of it."""
    meta, arr = Hh.load("code2_h256_bidir")
    model = Hh.code2_model(meta)
    G = Hh.code2_batch(arr)
    kw = meta["ctor"]
    out = O.code2_forward(model.state_dict(), G, num_layers=kw["num_layers"], bidirectional=True, out_wx=False,
                          out_pool_all=False, out_pool="max", max_seq_len=meta["S"], dtype=torch.float64)
    assert max(Hh.maxdiff(o, r) for o, r in zip(out, arr["pred"])) < TOL


@pytest.mark.parametrize("name", Hh.GRAD)
def test_oracle_training_step_gradients_match_reference(name):
    """`code2_grads` (autograd through the restatement) against the reference's own `loss.backward()` on the
    same seeded step (ogbg-code/main_pyg.py:55-62)."""
    meta, arr = Hh.load(name)
    model = Hh.code2_model(meta)
    G = Hh.code2_batch(arr)
    kw = meta["ctor"]
    loss, grads = O.code2_grads(model.state_dict(), G, torch.from_numpy(arr["y"]), num_layers=kw["num_layers"],
                                bidirectional=bool(kw["bidirectional"]), out_wx=kw["out_wx"],
                                out_pool_all=kw["out_pool_all"], out_pool=kw["out_pool"], max_seq_len=meta["S"])
    assert abs(float(loss) - float(arr["loss"])) < 1e-5
    assert Hh.check_grads(meta, arr, grads, rtol=5e-5) < 5e-5


@pytest.mark.parametrize("name", Hh.GRAD_VAR)
def test_oracle_variant_training_step_gradients_match_reference(name):
    """The same for the constructor-string variants that train through HIP (`gated_sum` with / without mapper bias,
    `mattn_h`, `add`; dagnn.py:232-276,379-409): the oracle's autograd against the reference's `loss.backward()`."""
    meta, arr = Hh.load(name)
    model = Hh.code2_model(meta)
    G = Hh.code2_batch(arr)
    kw = meta["ctor"]
    loss, grads = O.code2_grads(model.state_dict(), G, torch.from_numpy(arr["y"]), num_layers=kw["num_layers"],
                                bidirectional=True, out_wx=kw["out_wx"], max_seq_len=meta["S"], agg=kw["agg"],
                                recurr=kw.get("recurr", 1), agg_x=kw.get("agg_x", False))
    assert abs(float(loss) - float(arr["loss"])) < 1e-5
    if kw["agg"] in ("add", "max"):
        # the reference builds ONE AggConv for every layer and direction (dagnn.py:74-75): `named_parameters()` lists it
        # once, under its first name, with the gradient of all its uses; the oracle differentiates a state_dict in
        # which every alias is a tensor of its own - add the aliases up
        import re
        for k in [k for k in grads if re.match(r"node_aggr_\d+\.\d+\.", k)]:
            first = re.sub(r"^node_aggr_\d+\.\d+\.", "node_aggr_0.0.", k)
            if k != first:
                grads[first] = grads[first] + grads[k]
    # (fp32 on both sides; the dot-product attention of a 3-layer stack leaves ~2e-4 of rounding in its smallest gradients)
    assert Hh.check_grads(meta, arr, grads, rtol=3e-4) < 3e-4


@pytest.mark.parametrize("name", Hh.DVAE_GRAD)
def test_oracle_dvae_encoder_gradients_match_reference(name):
    """`dvae_grads` against the reference encoder's own `.backward()` (dvae/dagnn.py:177-184 under autograd)."""
    import dagnn_amd
    meta, arr = Hh.load(name)
    model, nn_ = Hh.dvae_model(meta)
    G = dagnn_amd.GraphBatch.from_data_list(Hh.dvae_graphs(meta, arr))
    loss, grads = O.dvae_grads(model.state_dict(), G, torch.from_numpy(arr["r1"]), torch.from_numpy(arr["r2"]),
                               num_layers=meta["L"], bidirectional=meta["bidir"], num_nodes=nn_,
                               vids=meta["kind"] == "na", agg=meta.get("agg", "attn_h"))
    assert abs(float(loss) - float(arr["loss"])) < 1e-4 * max(1.0, abs(float(arr["loss"])))
    assert Hh.check_grads(meta, arr, grads, rtol=5e-5) < 5e-5
