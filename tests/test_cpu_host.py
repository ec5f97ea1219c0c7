"""CPU tier (-m "not gpu"): host logic, the C-ABI library's exports, data plumbing, sharding."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import dagnn_amd
from dagnn_amd import _lib, collate_sharded, dag_utils, shard_by_nodes, synth
from tests import helpers as Hh

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


# ------------------------------------------------------------------ C ABI surface
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dagnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dagnn_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), name
    assert set(declared) == set(_lib.SYMBOLS), set(declared) ^ set(_lib.SYMBOLS)
    assert lib.dagnn_version().decode().startswith("dagnn_hip")


def test_host_only_entry_points_without_gpu():
    lib = _lib.load()
    n = lib.dagnn_plan_bytes(16561, 25377, 128, 2)
    assert n > 0 and n % 4 == 0
    off = (ctypes.c_int64 * 26)()
    assert lib.dagnn_plan_layout(16561, 25377, 128, 2, off) == 0
    offs = list(off)
    assert offs[15] == n and all(a < b for a, b in zip(offs[:14], offs[1:15])) and all(o < n for o in offs[16:])
    assert lib.dagnn_plan_bytes(-1, 0, 0, 0) == 0
    # argument validation happens before any HIP call
    assert lib.dagnn_pack_whh(None, None, 8, None) == -22
    assert lib.dagnn_encode_ast(None, None, None, None, None, 20, None, 6, 5, 6, None) == -22  # H % 4 != 0
    assert lib.dagnn_gather_rows(None, 4, 4, 1, 8, 9, None, 4, 0, None) == -22
    assert lib.dagnn_readout_pool(None, None, 4, 4, 0, 0, None, 4, 0, None) == -22
    assert lib.dagnn_pack_batch(None, 1, None) == -22
    assert lib.dagnn_readout_max_batch(None, None, 1, None, 4, None) == -22
    rj = (_lib.ReadoutJob * 1)()
    rj[0].h, rj[0].ld_h, rj[0].width, rj[0].dir, rj[0].col_off = 64, 4, 8, 0, 0   # pitch < width
    plan0 = _lib.Plan(64, 0, 4, 3, 1, 0)
    assert lib.dagnn_readout_max_batch(ctypes.byref(plan0), rj, 1, 64, 8, None) == -22
    assert lib.dagnn_readout_max_batch(ctypes.byref(plan0), rj, 17, 64, 8, None) == -22 and lib.dagnn_readout_max_batch(ctypes.byref(plan0), rj, 0, 64, 8, None) == 0
    job = (_lib.PackJob * 1)()
    job[0].w, job[0].H, job[0].K = 64, 48, 64   # H % 32 != 0
    assert lib.dagnn_pack_batch(job, 1, None) == -22 and lib.dagnn_pack_batch(job, 0, None) == 0
    # variant entry points: the ctypes mirrors and the C structs agree on the layout (a misplaced field would move
    # the one bad value these cases plant)
    plan = _lib.Plan(None, 0, 4, 3, 1, 0)
    assert lib.dagnn_variant_run(None, None, None, None, None) == -22
    agg = _lib.VariantAggregator()
    agg.mode, agg.lands, agg.val_dim, agg.out_dim, agg.out, agg.ld_out = _lib.AGG_ADD, 1, 8, 8, 64, 8
    agg.vals, agg.ld_vals = 64, 8
    assert lib.dagnn_variant_aggregate(ctypes.byref(plan), ctypes.byref(agg), 0, 2, 2, None) == 0   # empty range
    assert lib.dagnn_variant_aggregate(ctypes.byref(plan), ctypes.byref(agg), 2, 0, 2, None) == -22  # direction
    assert lib.dagnn_variant_aggregate(ctypes.byref(plan), ctypes.byref(agg), 0, 0, 5, None) == -22  # beyond N
    agg.ld_out = 7
    assert lib.dagnn_variant_aggregate(ctypes.byref(plan), ctypes.byref(agg), 0, 2, 2, None) == -22  # pitch < width
    agg.ld_out, agg.mode = 8, _lib.AGG_GIVEN
    assert lib.dagnn_variant_aggregate(ctypes.byref(plan), ctypes.byref(agg), 0, 2, 2, None) == -22
    agg.mode, agg.val_dim = _lib.AGG_MAX, 2000
    assert lib.dagnn_variant_aggregate(ctypes.byref(plan), ctypes.byref(agg), 0, 2, 2, None) == -22  # > 1024 wide
    va = _lib.VariantArgs()
    va.num_stacked, va.dir_mask, va.H = 1, 1, 8
    ptrs = (ctypes.POINTER(ctypes.c_int32) * 2)()
    sched = (ctypes.c_int32 * 2)(0, 4)
    ptrs[0] = ctypes.cast(sched, ctypes.POINTER(ctypes.c_int32))
    nl = (ctypes.c_int32 * 2)(1, 0)
    c = va.cell[0][0]
    c.agg.mode, c.agg.lands, c.agg.val_dim, c.agg.out_dim, c.agg.out, c.agg.ld_out = _lib.AGG_ADD, 1, 8, 8, 64, 8
    c.agg.vals, c.agg.ld_vals = 64, 8
    c.recurrent, c.in_dim, c.input, c.ld_input, c.w_in_t, c.w_agg_t, c.h, c.ld_h = 0, 8, 64, 8, 64, 64, 64, 7
    assert lib.dagnn_variant_run(ctypes.byref(plan), ctypes.byref(va), ptrs, nl, None) == -22   # ld_h < H
    c.ld_h, c.num_maps = 8, 4
    assert lib.dagnn_variant_run(ctypes.byref(plan), ctypes.byref(va), ptrs, nl, None) == -22   # too many maps
    c.num_maps, c.recurrent = 0, 1
    assert lib.dagnn_variant_run(ctypes.byref(plan), ctypes.byref(va), ptrs, nl, None) == -22   # GRU without biases
    c.recurrent = 0
    assert lib.dagnn_variant_run(ctypes.byref(plan), ctypes.byref(va), ptrs, nl, None) == -22   # plan without data


def test_dataflow_argument_structs_carry_the_round6_fields():
    """`dagnn_dataflow_args.stat_rows` (the forward launch of a training pass writes the reverse sweep's static rows into the
    buffers passed as `gh_out`; `gi_out` must then be NULL) and `slices64`, `dagnn_bwd_dataflow_args.stat_rows_written`: the
    ctypes mirrors end where the C structs end, and the one bad value each case plants is refused before any HIP call."""
    lib = _lib.load()
    plan = _lib.Plan(64, 0, 4, 3, 1, 2, 0)
    a = _lib.DataflowArgs()
    a.num_stacked, a.dir_mask, a.H, a.ld_h, a.gld, a.pld, a.groups, a.epoch = 2, 1, 64, 68, 68, 192, 1, 1
    a.schedule, a.err = 64, 64
    for i in range(2):
        c = a.cell[0][i]
        c.w_hh = c.b_hh = c.w_key = c.h_out = c.granules = c.edge_gain = 64
    a.cell[0][0].gi0 = 64
    a.cell[0][1].w_ih = a.cell[0][1].b_ih = a.cell[0][1].proj_granules = 64
    a.stat_rows = 1
    a.cell[0][0].gh_out = a.cell[0][1].gh_out = 64
    a.cell[0][1].gi_out = 64                      # static rows AND kept pre-activations: refused
    assert lib.dagnn_dataflow_run(ctypes.byref(plan), ctypes.byref(a), None) == -22
    a.cell[0][1].gi_out, a.cell[0][1].gh_out = None, None   # a cell without its record buffer
    assert lib.dagnn_dataflow_run(ctypes.byref(plan), ctypes.byref(a), None) == -22
    a.cell[0][1].gh_out, a.slices64, a.H = 64, 1, 192       # the 64-unit shape exists for H = 256 / 320: other widths take the 32-unit one
    a.cell[0][1].gi_out = 64
    assert lib.dagnn_dataflow_run(ctypes.byref(plan), ctypes.byref(a), None) == -22
    a.H = 128                                                # ... and its own entry point refuses them
    assert lib.dagnn_dataflow_run_x(ctypes.byref(plan), ctypes.byref(a), None) == -22
    assert _lib.DataflowArgs.slices64.offset == ctypes.sizeof(_lib.DataflowArgs) - 8 or _lib.DataflowArgs.slices64.offset == ctypes.sizeof(_lib.DataflowArgs) - 4
    assert _lib.BwdDataflowArgs.stat_rows_written.offset >= _lib.BwdDataflowArgs.xcd_first.offset + 4


def test_engine_refuses_cpu_tensors():
    from dagnn_amd import engine
    with pytest.raises(_lib.DagnnHipError):
        engine.pack_whh(torch.randn(12, 4))
    meta, arr = Hh.load("code2_h32_bidir")
    model = Hh.code2_model(meta)
    with pytest.raises(_lib.DagnnHipError), torch.no_grad():
        model(Hh.code2_batch(arr))  # no CPU fallback for the hot path


# ------------------------------------------------------------------ layering / decoders vs the reference
@pytest.mark.parametrize("name", Hh.CODE2)
def test_layering_matches_reference_top_sort(name):
    meta, arr = Hh.load(name)
    batch, ei = arr["batch"], arr["edge_index"]
    for g in range(int(batch.max()) + 1):
        nodes = np.flatnonzero(batch == g)
        n0 = nodes[0]
        em = batch[ei[0]] == g
        sub = ei[:, em] - n0
        assert np.array_equal(dag_utils.longest_path_layers(sub, len(nodes)), arr["layer0"][nodes])
        assert np.array_equal(dag_utils.longest_path_layers(sub[::-1], len(nodes)), arr["layer1"][nodes])
        assert dag_utils.check_layers(sub, arr["layer0"][nodes])


def test_layering_rejects_cycles():
    with pytest.raises(ValueError):
        dag_utils.longest_path_layers(np.array([[0, 1, 2], [1, 2, 0]]), 3)


@pytest.mark.parametrize("name", Hh.DVAE)
def test_dvae_decoders_match_reference(name):
    meta, arr = Hh.load(name)
    rows = [json.loads(r) for r in arr["rows"]]
    dec = synth.decode_enas_row if meta["kind"] == "na" else synth.decode_bn_row
    b = synth.dvae_batch([dec(r) for r in rows])
    assert np.array_equal(b.x.numpy(), arr["x"])
    assert np.array_equal(b.edge_index.numpy(), arr["edge_index"])
    assert np.array_equal(b.bi_layer_index.numpy(), arr["bi_layer_index"])
    assert np.array_equal(b.batch.numpy(), arr["batch"])


def test_headline_generator_reproduces_survey_numbers():
    b = synth.code2_batch(0, 128)
    assert (b.x.shape[0], b.edge_index.shape[1]) == (16561, 25377)
    assert int(b._bi_layer_idx0.max()) + 1 == 374 and int(b._bi_layer_idx1.max()) + 1 == 374
    assert b.num_graphs == 128 and int(b.batch[-1]) == 127
    ea = b.edge_attr
    assert set(map(tuple, ea.unique(dim=0).tolist())) == {(0.0, 0.0), (1.0, 0.0)}  # utils2.py:44,67


# ------------------------------------------------------------------ collation / sharding
def test_collation_offsets_index_keys_only():
    gs = synth.code2_graphs(5, 3, 20)
    b = dagnn_amd.GraphBatch.from_data_list(gs)
    n = [g.num_nodes for g in gs]
    assert b.edge_index.shape[0] == 2 and int(b.edge_index.max()) < sum(n)
    # "_index" keys are shifted by the running node count, "_idx" keys are not (dagnn.py:129)
    assert torch.equal(b._bi_layer_index0, torch.arange(sum(n)))
    assert int(b._bi_layer_idx0.max()) < max(n)
    assert torch.equal(b.batch, torch.repeat_interleave(torch.arange(3), torch.tensor(n)))
    assert b.x.shape == (sum(n), 2) and b.edge_attr.shape[0] == b.edge_index.shape[1]


def test_shard_rule_matches_reference_collater():
    """`tg/dataloader.py:17-27` on the seed-0 headline batch: numbers measured from the reference
    (SURVEY.md §8(e))."""
    graphs = synth.code2_graphs(0, 128)
    n = [g.num_nodes for g in graphs]
    s2 = shard_by_nodes(n, 2)
    assert [(b - a, sum(n[a:b])) for a, b in zip(s2[:-1], s2[1:])] == [(60, 8250), (68, 8311)]
    s8 = shard_by_nodes(n, 8)
    got = [(b - a, sum(n[a:b])) for a, b in zip(s8[:-1], s8[1:])]
    assert got == [(12, 2095), (16, 2078), (16, 2012), (16, 2065), (15, 2099), (18, 2064), (17, 2052), (18, 2096)]
    assert shard_by_nodes(n, 1) == [0, 128]
    # more devices than graphs: empty devices are dropped
    assert len(shard_by_nodes([5, 5], 8)) - 1 == 2
    shards = collate_sharded(graphs, 8)
    assert [s.num_graphs for s in shards] == [g for g, _ in got]


# ------------------------------------------------------------------ module contract
@pytest.mark.parametrize("name", Hh.CODE2)
def test_state_dict_contract_code2(name):
    meta, _ = Hh.load(name)
    model = Hh.code2_model(meta)
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(sd.items()) == list(meta["state_dict"].items())  # names, shapes AND order


@pytest.mark.parametrize("name", Hh.DVAE)
def test_state_dict_contract_dvae(name):
    meta, _ = Hh.load(name)
    model, _ = Hh.dvae_model(meta)
    sd = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(sd.items()) == list(meta["state_dict"].items())
    # cells_d alias the base class's encoder GRUs (dvae/dagnn.py:73-75)
    assert model.cells_0[0].weight_hh is model.grue_forward[0].weight_hh


def test_ctor_errors_and_defaults():
    with pytest.raises(ValueError):
        dagnn_amd.DAGNN(10, 5, emb_dim=64, hidden_dim=32, out_dim=None, agg_x=True)  # dagnn.py:27-28
    m = dagnn_amd.DAGNN(10, 5, emb_dim=16, hidden_dim=16, out_dim=None, encoder=dagnn_amd.ASTNodeEncoder(16, 4, 4, 20))
    assert m.out_wx and m.output_all and m.out_hidden_dim == 16 * 2 + 16 * 2 * 2  # reference defaults (:19-21,44)
    m2 = dagnn_amd.DAGNN(10, 5, 16, 16, None, agg="gated_sum", encoder=None)
    assert "node_aggr_0.0.mapper.weight" in m2.state_dict()
    meta, arr = Hh.load("var_h64_gated_sum")
    with pytest.raises(_lib.DagnnHipError):   # the variants run on torch-ROCm ops: a CPU batch is refused, not computed
        Hh.code2_model(meta)(Hh.code2_batch(arr))


def test_checkpoint_roundtrip_with_module_prefix(tmp_path):
    """Reference checkpoints carry a `module.` prefix (saved from DataParallel, utils2.py:85-102)."""
    meta, _ = Hh.load("code2_h32_bidir")
    model = Hh.code2_model(meta)
    ckpt = {"epoch": 3, "model": {"module." + k: v for k, v in model.state_dict().items()}}
    path = tmp_path / "ckpt.pt"
    torch.save(ckpt, path)
    fresh = Hh.code2_model({**meta, "w_seed": 1})
    wrapped = torch.nn.DataParallel(fresh) if False else None  # DataParallel needs a GPU; strip the prefix instead
    sd = {k[len("module."):]: v for k, v in torch.load(path)["model"].items()}
    fresh.load_state_dict(sd, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), model.state_dict().values()))
    assert wrapped is None


def test_grad_mode_raises_instead_of_silently_detaching():
    """Configurations the HIP backward does not cover refuse a differentiable call; the covered one goes to
    the HIP path (and, on CPU tensors, fails loudly there - there is no fallback)."""
    meta, arr = Hh.load("code2_h64_unidir")
    model = Hh.code2_model(meta)
    with pytest.raises(_lib.DagnnHipError):   # every attn_h read-out is differentiable through the HIP path
        model(Hh.code2_batch(arr))
    meta, arr = Hh.load("code2_h64_attn_x")
    with pytest.raises(_lib.DagnnHipError):
        Hh.code2_model(meta)(Hh.code2_batch(arr))
    model = Hh.code2_model(meta)
    model.schedule = "pergraph"   # the backward sweep exists for the lock-step schedule only
    with pytest.raises(NotImplementedError):
        model(Hh.code2_batch(arr))
    meta, arr = Hh.load("code2_h32_bidir")
    with pytest.raises(_lib.DagnnHipError):
        Hh.code2_model(meta)(Hh.code2_batch(arr))
    model, _ = Hh.dvae_model(Hh.load("na_h64_bidir")[0])
    with pytest.raises(_lib.DagnnHipError):   # differentiable D-VAE encoders take the HIP path too
        model(Hh.dvae_batch(Hh.load("na_h64_bidir")[1]))


# ------------------------------------------------------------------ loader-side plan
def test_host_plan_layout_matches_library():
    from dagnn_amd import host_plan
    lib = _lib.load()
    names = ["node_ptr", "edge_ptr", "depth0", "depth1", "order0", "order1", "lstart0", "lstart1", "rowptr0",
             "rowptr1", "col0", "col1", "eattr0", "eattr1", "items", "total", "blptr0", "blptr1", "rowrec0",
             "rowrec1", "slot0", "slot1", "eidx0", "eidx1"]
    for N, E, B, R in ((16561, 25377, 128, 2), (7, 0, 3, 0), (512, 723, 64, 1), (0, 0, 0, 2)):
        off = (ctypes.c_int64 * 26)()
        assert lib.dagnn_plan_layout(N, E, B, R, off) == 0
        lay = host_plan.plan_layout(N, E, B, R)
        assert {k: lay[k] for k in names} == {k: int(v) // 4 for k, v in zip(names, off)}
        assert lay["total"] * 4 == lib.dagnn_plan_bytes(N, E, B, R)


def test_dataflow_layout_matches_library():
    from dagnn_amd import host_plan
    lib = _lib.load()
    names = ["grp_of", "gdepth", "gload", "loff", "gtab0", "gtab1", "lcnt0", "lcnt1", "glbase0", "glbase1", "grec0",
             "grec1", "total"]
    for N, B, G in ((16561, 128, 5), (7, 3, 2), (512, 64, 32), (1, 1, 1)):
        off = (ctypes.c_int64 * 13)()
        assert lib.dagnn_dataflow_layout(N, B, G, off) == 0
        lay = host_plan.dataflow_layout(N, B, G)
        assert {k: lay[k] for k in names} == {k: int(v) // 4 for k, v in zip(names, off)}
        assert lay["total"] * 4 == lib.dagnn_dataflow_bytes(N, B, G)
    # groups the device hosts: 2 per workgroup set, floor(CUs / (dirs * (2L - 1) * H / 32)) sets, capped by the graphs
    # (and by the 64 groups the assignment kernel handles); 0 = not applicable
    assert lib.dagnn_dataflow_groups(256, 2, 2, 256, 128) == 10
    assert lib.dagnn_dataflow_groups(256, 1, 2, 128, 64) == 42
    assert lib.dagnn_dataflow_groups(256, 2, 2, 256, 3) == 3
    assert lib.dagnn_dataflow_groups(256, 2, 5, 512, 256) == 0 and lib.dagnn_dataflow_groups(256, 2, 2, 300, 8) == 0
    # weight-stationary tile kernel (H = 512): launches = chunks of stacked layers the device hosts at 32 workgroups a cell
    assert lib.dagnn_tiles_launches(256, 2, 5, 512, 2) == 2 and lib.dagnn_tiles_launches(256, 2, 1, 512, 0) == 1
    assert lib.dagnn_tiles_launches(256, 1, 8, 512, 2) == 1 and lib.dagnn_tiles_launches(128, 2, 5, 512, 2) == 3   # (8 cells of one direction fit at once)
    assert lib.dagnn_tiles_launches(256, 2, 5, 256, 2) == 0 and lib.dagnn_tiles_launches(256, 2, 5, 512, 3) == 0
    assert lib.dagnn_tiles_launches(32, 2, 2, 512, 2) == 0
    assert lib.dagnn_dataflow_groups(16, 2, 2, 256, 8) == 0


@pytest.mark.parametrize("seed,B,mean_n,G", [(1, 17, 60, 5), (0, 128, 125, 5), (3, 4, 30, 8), (4, 40, 15, 3)])
def test_dataflow_schedule_host_properties(seed, B, mean_n, G):
    """The schedule the persistent kernel walks (host mirror of csrc/dataflow.hip): every node exactly once per
    direction, blocks of 4 records that never mix topological layers or groups, live records first, every
    predecessor in an earlier block of the same group (the kernel's deadlock-freedom argument), LPT assignment."""
    from dagnn_amd import host_plan, synth
    b = synth.code2_batch(seed, B, mean_n)
    N, E = b.x.shape[0], b.edge_index.shape[1]
    G = min(G, B)
    ws, _, _ = host_plan.build_plan_host(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, b.batch, B, b.edge_attr)
    out = host_plan.build_dataflow_schedule_host(ws, N, E, B, 2, G, 8, 1)
    S = host_plan.dataflow_layout(N, B, G)
    assert tuple(out[:3]) == (G, host_plan.DF_MAGIC, 4)
    grp = out[S["grp_of"]:S["grp_of"] + B]
    assert grp.min() >= 0 and grp.max() < G
    n_of = np.diff(b.ptr.numpy())
    depth = np.array([int(b._bi_layer_idx0[b.ptr[g]:b.ptr[g + 1]].max()) + 1 for g in range(B)])
    # the deepest graph opens group 0; loads are what the rule says
    assert grp[int(np.argmax(depth))] == 0
    gload = out[S["gload"]:S["gload"] + G]
    for k in range(G):
        members = np.flatnonzero(grp == k)
        if members.size:
            assert gload[k] == 8 * depth[members].max() + n_of[members].sum()
            assert out[S["gdepth"] + k] == depth[members].max()
    ei = b.edge_index.numpy()
    for d in (0, 1):
        layer = (b._bi_layer_idx0 if d == 0 else b._bi_layer_idx1).numpy()
        gt = out[S["gtab%d" % d]:S["gtab%d" % d] + 2 * G].reshape(G, 2)
        rec = out[S["grec%d" % d]:S["grec%d" % d] + 16 * (4 * N + 4)].reshape(-1, 16)
        live = rec[:, 0] >= 0
        assert sorted(rec[live, 0].tolist()) == list(range(N))
        assert gt[:, 0].tolist() == (np.cumsum(gt[:, 1] * 4) - gt[:, 1] * 4).tolist()
        assert not live[int(gt[-1, 0] + 4 * gt[-1, 1]):].any()
        block_of = np.full(N, -1)
        feed, other = (ei[1], ei[0]) if d == 0 else (ei[0], ei[1])
        for k in range(G):
            last_layer = -1
            for blk in range(gt[k, 1]):
                r = rec[gt[k, 0] + 4 * blk: gt[k, 0] + 4 * blk + 4]
                lv = r[:, 0] >= 0
                assert lv[0] and not (np.diff(lv.astype(int)) > 0).any()      # at least one record, live ones first
                nodes = r[lv, 0]
                assert (grp[b.batch.numpy()[nodes]] == k).all()
                assert len(set(layer[nodes].tolist())) == 1 and layer[nodes[0]] >= last_layer
                last_layer = layer[nodes[0]]
                block_of[nodes] = blk
                for row in r[lv]:
                    preds = other[feed == row[0]]
                    assert row[2] - row[1] == preds.size
                    assert (block_of[preds] >= 0).all() and (block_of[preds] < blk).all()
                    assert row[4:4 + min(4, preds.size)].tolist() == preds[:4].tolist()


def test_host_plan_against_brute_force():
    from dagnn_amd import host_plan, synth
    b = synth.code2_batch(3, 40, 40)   # wide enough for fat layers: shallow and deep graphs both exist
    B, N, E = 40, b.x.shape[0], b.edge_index.shape[1]
    ws, sched, splits = host_plan.build_plan_host(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, b.batch, B, b.edge_attr)
    lay = host_plan.plan_layout(N, E, B, 2)
    ei, gid = b.edge_index.numpy(), b.batch.numpy()
    for d in (0, 1):
        layer = (b._bi_layer_idx0 if d == 0 else b._bi_layer_idx1).numpy()
        T = int(layer.max()) + 1
        bl = ws[lay["blptr%d" % d]:lay["blptr%d" % d] + N + 2]
        assert bl[N + 1] == T and np.array_equal(bl[:T + 1], sched[d]) and bl[T] == N
        rec = ws[lay["rowrec%d" % d]:lay["rowrec%d" % d] + 16 * N].reshape(N, 16)
        col = ws[lay["col%d" % d]:lay["col%d" % d] + E]
        eidx = ws[lay["eidx%d" % d]:lay["eidx%d" % d] + E]
        eattr = ws[lay["eattr%d" % d]:lay["eattr%d" % d] + 2 * E].view(np.float32).reshape(E, 2)
        feed, other = ei[1 - d], ei[d]
        thr = int(ws[5 + d])
        width = np.bincount(layer, minlength=T)
        fat = width > host_plan.THIN_ROWS
        assert thr == (np.flatnonzero(fat).max() + 1 if fat.any() else 0)
        depth_g = np.array([layer[gid == g].max() + 1 for g in range(B)])
        sp = ws[lay["blsplit%d" % d]:lay["blsplit%d" % d] + T]
        assert np.array_equal(sp, splits[d])
        for t in range(T):
            nodes = rec[bl[t]:bl[t + 1], 0]
            members = np.flatnonzero(layer == t)
            deep = depth_g[gid[members]] > thr
            # inside a layer: the shallow graphs' rows, then the deep graphs' rows, each in (graph, id) order
            assert np.array_equal(nodes, np.concatenate([members[~deep], members[deep]]))
            assert sp[t] == bl[t] + (~deep).sum()
        for v, eb, ee, g in rec[:, :4]:
            e_ids = np.flatnonzero(feed == v)
            assert g == gid[v] and np.array_equal(eidx[eb:ee], e_ids) and np.array_equal(col[eb:ee], other[e_ids])
            assert np.array_equal(eattr[eb:ee], b.edge_attr.numpy()[e_ids])
        slot = ws[lay["slot%d" % d]:lay["slot%d" % d] + N]
        assert np.array_equal(rec[slot, 0], np.arange(N))
    bad = b.batch.clone()
    bad[2] = 5
    with pytest.raises(ValueError):
        host_plan.build_plan_host(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, bad, B, b.edge_attr)


def test_augment_edge2_matches_reference_fixture():
    from types import SimpleNamespace
    from dagnn_amd import augment_edge2
    meta, arr = Hh.load("augment_edge2")
    for k in range(meta["cases"]):
        d = SimpleNamespace(edge_index=torch.from_numpy(arr["in_edge_index_%d" % k]).long(),
                            node_is_attributed=torch.from_numpy(arr["in_attributed_%d" % k]).view(-1, 1))
        out = augment_edge2(d)
        assert np.array_equal(out.edge_index.numpy(), arr["out_edge_index_%d" % k])
        assert np.array_equal(out.edge_attr.numpy(), arr["out_edge_attr_%d" % k])
        assert out.edge_attr.dtype == torch.float32


# ------------------------------------------------------------------ decoder-side single-vertex step
@pytest.mark.parametrize("name", ["iprop_na_h64_L2", "iprop_bn_h32_L3"])
def test_ipropagate_to_oracle_matches_reference(name):
    """The CPU restatement of `_ipropagate_to` (oracle/iprop_oracle.py, the checker of the HIP step) against the
    reference's own (`dvae/dagnn.py:187-239`, `dvae/dagnn_bn.py:179-238`) on the fixture's stand-in igraph graphs:
    returned states, the states it writes into the vertices, a vertex without predecessors, graphs too short for `v`,
    and the `H`-given form.  (The product's `_ipropagate_to` is one HIP launch: tests/test_gpu_parity.py.)"""
    from oracle.iprop_oracle import ipropagate_to
    Hh.check_ipropagate(name, lambda model, G, v, H=None: ipropagate_to(model, G, v, model.grud, H=H), "cpu", 2e-6)


def test_ipropagate_to_has_no_cpu_path():
    """The product's single-vertex step is HIP only: a model on the CPU raises instead of falling back."""
    from dagnn_amd._lib import DagnnHipError
    meta, arr = Hh.load("iprop_bn_h32_L3")
    model, _ = Hh.dvae_model(dict(meta, bidir=False))
    G = Hh.iprop_graphs(meta, arr, "cpu")
    with pytest.raises(DagnnHipError):
        model._ipropagate_to(G, meta["vs"][0], model.grud)


def test_tile_kernel_tail_split_policy(monkeypatch):
    """Where a large h = 512 batch is handed from the per-layer launches to the tile kernel (engine.tiles_tail_split): per
    direction the first batch-level layer behind which no layer has more than TILES_TAIL_ROWS rows; no split without a
    tail of at least 32 layers, or when switched off."""
    from dagnn_amd import engine

    class Plan(object):
        def __init__(self, rows):
            self.rows = rows

        def read_schedule(self):
            return [np.concatenate([[0], np.cumsum(r)]).astype(np.int32) for r in self.rows]

    fwd = [300, 200, 90, 40, 33] + [5] * 60
    bwd = [50, 60, 70, 20, 10, 33, 8] + [2] * 40
    monkeypatch.setattr(engine, "TILES_TAIL_ROWS", 32)
    assert engine.tiles_tail_split(Plan([fwd, bwd]), [0, 1]) == [5, 6]
    assert engine.tiles_tail_split(Plan([fwd, bwd]), [0]) == [5, 0]
    assert engine.tiles_tail_split(Plan([[400] * 10 + [3] * 20, [400] * 10 + [3] * 20]), [0, 1]) is None   # tail of 20 layers
    assert engine.tiles_tail_split(Plan([[4] * 100, [7] * 90]), [0, 1]) == [0, 0]                          # nothing wide at all
    monkeypatch.setattr(engine, "TILES_TAIL_ROWS", 0)
    assert engine.tiles_tail_split(Plan([fwd, bwd]), [0, 1]) is None
    # few wide layers (D-VAE batches at hs = 501) stay on the launches: more than TILES_MAX_MEAN_ROWS rows per layer
    monkeypatch.setattr(engine, "TILES", 1)
    monkeypatch.setattr(engine, "TILES_MAX_MEAN_ROWS", 160)
    for rows, flat in (([128] * 10, False), ([512] * 10, True), ([200] * 10, True), ([150] * 12, False), ([44] * 374, False)):
        p = Plan([rows, rows[::-1]])
        p.N = sum(rows)
        assert engine.tiles_batch_too_flat(p, [0, 1]) == flat, rows[0]

    class NoRead(object):
        N = 1536

        def read_schedule(self):
            raise AssertionError("a small batch reads no schedule")
    assert engine.tiles_batch_too_flat(NoRead(), [0]) is False
    monkeypatch.setattr(engine, "TILES", 2)
    p = Plan([[512] * 10, [512] * 10])
    p.N = 5120
    assert engine.tiles_batch_too_flat(p, [0, 1]) is False


def test_state_width_policy(monkeypatch):
    """engine.state_width: the next multiple of 64, except that 256 < H < 512 goes to 512 (the tile kernel's width) for
    stacked models, and for single-layer ones from 385 up; never with more than two edge features, nor when the tile
    kernel or the padding is switched off."""
    from dagnn_amd import engine
    monkeypatch.setattr(engine, "TILES", 1)
    monkeypatch.setattr(engine, "TILES_PAD", 1)
    assert [engine.state_width(h, 2) for h in (32, 200, 256, 257, 300, 384, 448, 501, 512, 600)] == \
        [64, 256, 256, 512, 512, 512, 512, 512, 512, 640]
    assert [engine.state_width(h, 1) for h in (257, 320, 384, 385, 448, 512)] == [320, 320, 384, 512, 512, 512]
    assert engine.state_width(300, 3, 3) == 320
    monkeypatch.setattr(engine, "TILES_PAD", 2)
    assert engine.state_width(300, 1) == 512
    monkeypatch.setattr(engine, "TILES_PAD", 0)
    assert engine.state_width(300, 3) == 320
    monkeypatch.setattr(engine, "TILES_PAD", 1)
    monkeypatch.setattr(engine, "TILES", 0)
    assert engine.state_width(300, 3) == 320
