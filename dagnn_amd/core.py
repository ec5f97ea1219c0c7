"""Shared forward machinery of the three DAGNN modules (ogbg-code DAGNN, D-VAE NA and BN encoders).

`run_stack` drives the HIP path for `L` stacked GRU layers in every requested direction:

    for i in range(L):                      # stacked layers (loop 3 of dagnn.py:171)
        gi[d] = u_i @ W_ih[d][i]^T + b_ih   # ONE batched fp32-MFMA GEMM per layer, all nodes
        h[d][i] = recurrence(gi[d], ...)    # persistent per-(graph, direction) workgroups walk
                                            # all topological layers (loops 1+2 of dagnn.py:145-157)

which is the reference's loop nest `for d: for layer: for i:` re-ordered legally: layer i of the
stack only needs layer i-1 at the SAME node (`dagnn.py:177,181`) and layer i at the predecessors,
so finishing layer i-1 for all nodes first changes no value.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import engine
from ._lib import DagnnHipError


def round_up4(n: int) -> int:
    return (n + 3) // 4 * 4


def round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


SCHEDULES = ("lockstep", "pergraph")


class DerivedCache(object):
    """Caches tensors derived from parameters (packed / padded weights, folded edge gains) and
    rebuilds them when any source parameter changed (in-place updates bump `_version`).

    The version counter does not see every update: fused optimizers (`torch.optim.Adam(fused=True)`) and writes
    through `.data` change a parameter without bumping it.  The modules therefore ask for `fresh=True` while they are
    in training mode (derived tensors are rebuilt on every forward - parameters change every step there anyway) and
    drop their caches whenever `train()` / `eval()` is called; the cache serves evaluation, where parameters only
    change through `load_state_dict` / `copy_` (which do bump the counter).  `invalidate()` is the manual override."""

    def __init__(self):
        self._key = None
        self._val = None
        self._built = None   # (event behind the kernels that built the value, stream they ran on, streams that have met it)

    def invalidate(self) -> None:
        self._key = None
        self._val = None
        self._built = None

    def get(self, params: Sequence[torch.Tensor], fn: Callable[[], object], fresh: bool = False):
        key = None if fresh else tuple([(p.data_ptr(), p._version, p.device.index) for p in params])
        if fresh or key != self._key:
            with torch.no_grad():
                self._val = fn()
            self._key = key
            self._built = built_marker(params[0] if params else None)
        else:
            meet_built(self._built)
        return self._val


class ParamGuard(object):
    """Catches what `DerivedCache` cannot see.  The caches key on (data_ptr, _version); a fused optimizer step or a write
    through `.data` moves neither, and an evaluation loop that interleaves such updates without a `train()` / `eval()` call
    would read derived weights of the OLD parameters - silently.  Every evaluation pass therefore fingerprints the module's
    parameters on the device (`dagnn_param_fingerprint`: 1024 words spread over each tensor, ONE small launch): when the
    cache key moved the fingerprints are recorded, otherwise they are compared and a difference sets a bit of the arena's
    error word, which travels to the host with the pass's other error words (no synchronisation) and raises at the next
    `poll()` / `check()` like any device-side failure.  A sampled check: an update that touches every element (any optimizer
    step, `copy_` of new weights) is caught with certainty, a write to a few hand-picked elements need not be.
    `DAGNN_AMD_PARAM_GUARD=0` switches it off."""

    def __init__(self):
        self._key = None
        self._fp = None
        self._args = None

    def reset(self) -> None:
        self._key = None

    def check(self, params: Sequence[torch.Tensor], err: Optional[torch.Tensor]) -> None:
        if not engine.PARAM_GUARD or err is None or not params:
            return
        import ctypes as C
        key = tuple([(p.data_ptr(), p._version) for p in params])
        record = key != self._key
        if record or self._args is None:
            chunks = []
            cap = 96
            for o in range(0, len(params), cap):
                part = [p for p in params[o:o + cap]]
                ptrs = (C.c_void_p * len(part))(*[p.data_ptr() for p in part])
                numel = (C.c_int64 * len(part))(*[p.numel() if p.element_size() == 4 else 0 for p in part])
                chunks.append((ptrs, numel, len(part), o))
            self._args = chunks
            if self._fp is None or self._fp.numel() < len(params) or self._fp.device != err.device:
                self._fp = torch.empty(max(len(params), 1), dtype=torch.int64, device=err.device)
            self._key = key
        lib = engine._lib.load()
        st = engine._stream(err)
        for ptrs, numel, n, o in self._args:
            engine.check(lib.dagnn_param_fingerprint(ptrs, numel, n, self._fp.data_ptr() + 8 * o, 0 if record else 1,
                                                     err.data_ptr(), engine.ERR_PARAMS_MOVED, st), "dagnn_param_fingerprint")


def guard_params(mod, err: Optional[torch.Tensor]) -> None:
    """`ParamGuard.check` of module `mod`'s fp32 GPU parameters (evaluation passes of all three modules call this)."""
    if mod.training or not engine.PARAM_GUARD or err is None:
        return
    g = mod.__dict__.get("_param_guard")
    if g is None:
        g = mod.__dict__["_param_guard"] = ParamGuard()
        mod.__dict__["_param_list"] = [p for p in mod.parameters() if p.is_cuda and p.dtype == torch.float32]
    g.check(mod.__dict__["_param_list"], err)


def drop_guard(mod) -> None:
    mod.__dict__.pop("_param_list", None)
    mod.__dict__.pop("_param_guard", None)


def built_marker(t):
    """(event, stream, seen) behind derived tensors just built on the current stream of `t`'s device - a pass on ANOTHER
    stream (`bench.py --streams k`, micro-batches in flight) must not read them before the kernels that made them are done."""
    if t is None or not t.is_cuda:
        return None
    st = torch.cuda.current_stream(t.device)
    ev = torch.cuda.Event()
    ev.record(st)
    return (ev, st.cuda_stream, {st.cuda_stream}, t.device)


def meet_built(built) -> None:
    if built is None:
        return
    ev, _, seen, dev = built
    cur = torch.cuda.current_stream(dev)
    if cur.cuda_stream not in seen:
        cur.wait_event(ev)
        seen.add(cur.cuda_stream)


def _pad_gate_rows(w: torch.Tensor, H: int, Hp: int) -> torch.Tensor:
    """[3H, ...] -> [3Hp, ...] keeping the (r, z, n) blocks aligned; pad rows are zero."""
    if Hp == H:
        return w.contiguous()
    out = w.new_zeros((3 * Hp,) + tuple(w.shape[1:]))
    for g in range(3):
        out[g * Hp:g * Hp + H] = w[g * H:(g + 1) * H]
    return out


def _pad_cols(w: torch.Tensor, cols: int) -> torch.Tensor:
    if w.shape[-1] == cols:
        return w.contiguous()
    out = w.new_zeros(tuple(w.shape[:-1]) + (cols,))
    out[..., :w.shape[-1]] = w
    return out


class CellParams(object):
    """Device-side, kernel-ready parameters of one (direction, stacked layer) cell."""

    __slots__ = ("w_ih", "b_ih", "w_hh_t", "b_hh", "w_key", "edge_gain", "vid_bias", "w_hh_pk", "w_ih_pk",
                 "b_ih_dev", "Hp", "key_raw", "w_hh_raw", "w_hh_df", "w_ih_df", "w_hh_bt", "w_ih_bt", "df_ok", "gain_src", "fold", "built",
                 "agg", "agg_w", "agg_b")   # (agg / agg_w / agg_b: plain aggregators on the dataflow kernel, variants.run_plain_dataflow)


def pack_dataflow(cells, transposed_too: bool = False) -> None:
    """The dataflow kernel's weight layout of every cell (Hp <= 320), all matrices in one launch; `transposed_too` (training
    passes): the reverse sweep's gate-wise transposed layouts in the same launch; the edge gains of cells derived with
    `pack=False` ride along."""
    cells = list(cells)
    todo = []
    for c in cells:
        if c.w_hh_df is None:
            todo.append((c, "w_hh_df", c.w_hh_raw, False))
            if c.b_ih_dev is not None:
                todo.append((c, "w_ih_df", c.w_ih, False))
        if transposed_too and c.w_hh_bt is None:
            todo.append((c, "w_hh_bt", c.w_hh_raw, True))
            if c.b_ih_dev is not None:
                todo.append((c, "w_ih_bt", c.w_ih, True))
    gains = [c for c in cells if c.gain_src is not None]
    if not todo:
        fill_gains(gains)
        if cells:
            meet_built(cells[0].built)   # (packed by a pass on another stream, perhaps still in flight)
        return
    packed_all = engine.pack_dataflow_batch([(w, tr) for _, _, w, tr in todo], todo[0][0].Hp,
                                            gains=[(c.gain_src[0], c.gain_src[1], c.edge_gain) for c in gains])
    for c in gains:
        c.gain_src = None
    for (c, name, _, _), packed in zip(todo, packed_all):
        setattr(c, name, packed)
    marker = built_marker(packed_all[0])
    for c in cells:
        c.built = marker


def fill_gains(cells) -> None:
    """Edge gains `W_e^T w_key` of cells derived with `pack=False` that no batched pack launch took along."""
    for c in cells:
        if c.gain_src is not None:
            c.edge_gain.copy_(c.gain_src[0].t() @ c.gain_src[1])
            c.gain_src = None


def pack_lockstep(cells, force: bool = False) -> None:
    """The lock-step kernels' weight layouts of every cell derived with `pack=False`: one `dagnn_pack_batch` launch
    for all matrices instead of three launches per matrix.  Cells the dataflow kernel serves (Hp <= 256) get its
    layout instead; `force` (the fallback inside `run_stack_lockstep`) packs the per-layer launches' layouts too."""
    cells = list(cells)
    if cells and cells[0].df_ok and not force:
        pack_dataflow(cells)
        return
    fill_gains(cells)
    if cells and engine.TILES and cells[0].Hp == 512 and not force:
        return   # the tile kernel (csrc/tiles.hip) reads the torch layouts; the fallback packs with `force` if it is taken
    cells = [c for c in cells if c.w_hh_pk is None]
    todo = []
    for c in cells:
        todo.append((c, "w_hh_pk", c.w_hh_raw))
        if c.b_ih_dev is not None:
            todo.append((c, "w_ih_pk", c.w_ih))
    if not todo:
        return
    for (c, name, _), packed in zip(todo, engine.pack_batch([w for _, _, w in todo], todo[0][0].Hp)):
        setattr(c, name, packed)


def derive_cell(w_ih, w_hh, b_ih, b_hh, attn_w, H: int, dq: int, in_is_hidden: bool,
                edge_w: Optional[torch.Tensor], vid_nodes: int, schedule: str = "pergraph",
                key_dim: Optional[int] = None, pack: bool = True, stacked: int = 1) -> CellParams:
    """Fold / pack one cell's parameters for the kernels.

    attn_w is `attn_lin.weight` [1, dq + H (+ vid_nodes)]: the first dq entries multiply the query
    (they cancel in the segment softmax, as does the bias), the next H the key h_j, and for the NA
    variant the last `vid_nodes` the one-hot vertex id of the key (`dvae/dagnn.py:130-134`).
    `edge_w` is `edge_encoder.weight` [H, R]; its bias is constant inside a segment and cancels.
    `stacked` is the number of stacked cells of the model: it only picks the padded width (`engine.state_width`).
    """
    lock = schedule == "lockstep"
    wide_ok = edge_w is not None and not vid_nodes and (key_dim is None or key_dim == H)   # what dagnn_dataflow_run_wide takes
    Hp = engine.state_width(H, stacked, 0 if edge_w is None else int(edge_w.shape[1]), wide_ok=wide_ok) if lock else round_up4(H)
    c = CellParams()
    c.Hp = Hp
    c.fold = None   # (model._folded_tables: the encoder's tables folded through this cell's W_ih, stacked layer 0 only)
    c.built = None  # (`built_marker` behind the last lazy pack launch: passes on other streams meet it before they read the layouts)
    wi = _pad_gate_rows(w_ih.detach().float(), H, Hp)
    if in_is_hidden:
        wi = _pad_cols(wi, Hp)
    c.w_ih = wi
    c.b_ih = _pad_gate_rows(b_ih.detach().float(), H, Hp)
    whh = _pad_cols(_pad_gate_rows(w_hh.detach().float(), H, Hp), Hp)
    c.w_hh_raw = whh  # torch layout, padded: the backward sweep reads it as is
    c.w_hh_t = None if lock else engine.pack_whh(whh)
    c.w_hh_pk = c.w_ih_pk = None
    c.w_hh_df = c.w_ih_df = None
    c.w_hh_bt = c.w_ih_bt = None   # reverse sweep: packed gate-wise transposes (engine.bwd_dataflow_sweep)
    # the dataflow kernel's layout instead (packed below); a 320-wide cell only where `dagnn_dataflow_run_wide` applies (two
    # edge features, hidden-state keys, no vertex-id biases) - other 320-wide shapes run on the per-layer launches and
    # would otherwise be packed twice on every training step
    use_df = bool(engine.DATAFLOW and engine.dataflow_width(Hp) and
                  (Hp <= 256 or (wide_ok and edge_w is not None and int(edge_w.shape[1]) == 2)))
    c.df_ok = use_df
    if lock and pack and not use_df:   # pack=False: the caller batches the packing of all its cells (pack_lockstep)
        c.w_hh_pk = {js: engine.pack_slices(whh, Hp, js) for js in (16, 32)}
        c.w_hh_pk["mfma"] = engine.pack_mfma(whh, Hp)
        if in_is_hidden:
            c.w_ih_pk = {js: engine.pack_slices(wi, Hp, js) for js in (16, 32)}
            c.w_ih_pk["mfma"] = engine.pack_mfma(wi, Hp)
    c.b_ih_dev = c.b_ih if (lock and in_is_hidden) else None
    if lock and pack and use_df:
        pack_dataflow([c])
    c.b_hh = _pad_gate_rows(b_hh.detach().float(), H, Hp)
    kd = H if key_dim is None else key_dim  # keys are hidden states (H) or, for the `*_x` aggregators, inputs
    key = attn_w.detach().float()[0, dq:dq + kd]
    c.key_raw = key.contiguous()
    c.w_key = _pad_cols(key, Hp) if kd == H else None
    c.gain_src = None
    if edge_w is None:
        c.edge_gain = None
    elif pack or not lock:
        c.edge_gain = (edge_w.detach().float().t() @ key).contiguous()
    else:   # the caller batches: the gain is computed by `pack_lockstep` (inside the dataflow pack launch where that runs)
        c.edge_gain = torch.empty(edge_w.shape[1], dtype=torch.float32, device=edge_w.device)
        c.gain_src = (edge_w.detach().float().contiguous(), c.key_raw)
    c.vid_bias = attn_w.detach().float()[0, dq + kd:dq + kd + vid_nodes].contiguous() if vid_nodes else None
    return c


_OFF_DATAFLOW_SEEN = set()


def _warn_off_dataflow(dev, ndirs: int, L: int, Hp: int) -> None:
    """Say ONCE per model shape that it runs on the per-layer launches instead of the persistent dataflow kernel (which
    is 1.5-2x faster where it applies): the applicability limits are otherwise a silent cliff."""
    key = (str(dev), ndirs, L, Hp)
    if key in _OFF_DATAFLOW_SEEN:
        return
    _OFF_DATAFLOW_SEEN.add(key)
    import warnings
    cells = ndirs * (2 * L - 1)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    if Hp == 320 and not engine.DF_WIDE:
        why = "DAGNN_AMD_DF_WIDE=0 keeps hidden sizes 257..320 off the dataflow kernel's 320-wide shape"
    elif Hp == 320:
        why = "hidden size 320 runs on the dataflow kernel only with exactly two edge features, hidden-state keys and no vertex-id key biases"
    elif Hp > 256:
        why = "hidden size %d > 256 (a 32-unit slice of the [3H, H] matrices no longer fits the register file)" % Hp
    elif cells > 24:
        why = "%d kernel cells (directions x (2 x stacked layers - 1)) > 24" % cells
    elif cells * (Hp // 32) > cus:
        why = "%d kernel cells x %d slices need more than the %d CUs of this device" % (cells, Hp // 32, cus)
    else:
        why = "the kernel does not apply to this shape"
    warnings.warn("dagnn_amd: this model (hidden %d, %d stacked layers, %d direction(s)) runs on the per-layer launch path, "
                  "not the persistent dataflow kernel: %s" % (Hp, L, ndirs, why), RuntimeWarning, stacklevel=3)


def run_stack_lockstep(plan: engine.PlanHandle, x: torch.Tensor, cells: Dict[Tuple[int, int], CellParams],
                       dirs: Sequence[int], L: int, H: int, vid_nodes: int = 0,
                       arena: Optional[engine.GranuleArena] = None, static_score=None,
                       keep: Optional[dict] = None, gi0: Optional[Sequence[torch.Tensor]] = None) -> List[List[torch.Tensor]]:
    """Lock-step schedule (default): one batched input GEMM for stacked layer 0 of every direction,
    then T + L - 1 frontier launches covering all cells (csrc/frontier.hip).  `keep` (training) receives
    what the backward pass needs: the raw state buffers (`h_buf`) and `gi0`.  `gi0` (one [N, 3Hp] per direction): the
    input-side pre-activations of stacked layer 0 when the caller already has them (model._folded_gi0)."""
    Hp = cells[(dirs[0], 0)].Hp
    N, dev = x.shape[0], x.device
    if gi0 is None:
        gi0 = engine.gemm_nt_bias([x] * len(dirs), [cells[(d, 0)].w_ih for d in dirs], [cells[(d, 0)].b_ih for d in dirs])
    gi = [None, None]
    for q, d in enumerate(dirs):
        gi[d] = gi0[q]
    ld = engine.frontier_ld(Hp)  # state rows carry their H/16 partial attention scores behind the states
    h = [[torch.empty(N, ld, dtype=torch.float32, device=dev) if d in dirs else None for _ in range(L)]
         for d in range(2)]
    plan.wait_ready()   # a plan built on the side stream (model._plan_of) meets the caller's stream here
    groups = engine.dataflow_groups(dev, len(dirs), L, Hp, plan.B, training=keep is not None) if (arena is not None and N > 0) else 0
    if N * 3 * Hp >= (1 << 31):   # the dataflow kernels address their granule buffers with 32-bit row offsets
        groups = 0
    if Hp > 256 and (plan.R != 2 or static_score is not None or vid_nodes or any(c.edge_gain is None or c.w_key is None for c in cells.values())):
        groups = 0   # the H = 320 shape has the lean loaders only (dagnn_dataflow_run_wide)
    if arena is not None:
        arena.poll()   # a failure an earlier pass reported (no synchronisation); either path below is watched
    tiles = 0
    if groups == 0 and arena is not None and N > 0 and static_score is None and \
            N * ld * 4 < (1 << 31) - 16 and all(cells[(d, 0)].w_key is not None for d in dirs):
        tiles = engine.tiles_launches(dev, len(dirs), L, Hp, plan.R, N)   # wide states (H = 512): csrc/tiles.hip
        if tiles > 0 and engine.tiles_batch_too_flat(plan, dirs):
            tiles = 0   # few wide layers: the launches' 32-row tiles win (no thin tail either: the split below says None)
    if groups == 0 and tiles == 0 and arena is not None and N > 0 and engine.DATAFLOW and not (engine.TILES == 1 and L >= 2 and Hp == 512):
        _warn_off_dataflow(dev, len(dirs), L, Hp)
    split = None
    if tiles == 0 and groups == 0 and engine.TILES == 1 and L >= 2 and arena is not None and N > 0 and static_score is None and \
            N * ld * 4 < (1 << 31) - 16 and all(cells[(d, 0)].w_key is not None for d in dirs) and \
            engine._lib.load().dagnn_tiles_launches(engine._num_cus(dev), len(dirs), L, Hp, plan.R) > 0:
        split = engine.tiles_tail_split(plan, dirs)   # a batch too large for the tile kernel alone: it takes the thin tail
    if tiles > 0:
        engine.tiles_run(plan, dirs, L, Hp, cells, gi, h, arena, vid_mod=vid_nodes)
    elif split is not None:
        pack_lockstep(cells.values(), force=True)
        engine.frontier_run(plan, dirs, L, Hp, cells, gi, h, vid_mod=vid_nodes, arena=None, static_score=static_score, stop_layer=split,
                            chains=arena)
        engine.tiles_run(plan, dirs, L, Hp, cells, gi, h, arena, first_layer=split, vid_mod=vid_nodes)
    elif groups > 0:
        pack_dataflow(cells.values(), transposed_too=keep is not None and bool(engine.BWD_DATAFLOW))
        preact = {} if (keep is not None and engine.BWD_DATAFLOW) else None
        if preact is not None and engine.stat_rows_ok(dev, N, Hp, len(dirs), L, plan.B, groups):
            preact["stat_rows"] = True
        engine.dataflow_run(plan, dirs, L, Hp, cells, gi, h, groups, vid_mod=vid_nodes, arena=arena,
                            static_score=static_score, score_parts=keep is not None, preact=preact, training=keep is not None)
        if keep is not None:
            keep["preact"] = preact
    else:
        pack_lockstep(cells.values(), force=True)
        engine.frontier_run(plan, dirs, L, Hp, cells, gi, h, vid_mod=vid_nodes, arena=arena, static_score=static_score, chains=arena)
    if keep is not None:
        keep["h_buf"], keep["gi0"], keep["Hp"], keep["groups"] = h, gi, Hp, groups
    return [[h[d][i][:, :H] if h[d][i] is not None else None for i in range(L)] for d in range(2)]


def run_stack(plan: engine.PlanHandle, x: torch.Tensor, cells: Dict[Tuple[int, int], CellParams],
              dirs: Sequence[int], L: int, H: int, vid_nodes: int = 0,
              schedule: str = "pergraph", arena: Optional[engine.GranuleArena] = None,
              static_score=None, gi0: Optional[Sequence[torch.Tensor]] = None) -> List[List[torch.Tensor]]:
    """Hidden states h[d][i] ([N, H] each) of all stacked layers and directions.  `static_score[(d, i)]`
    ([N]) replaces the hidden-state attention scores for the aggregators whose keys are the inputs."""
    if schedule == "lockstep":
        return run_stack_lockstep(plan, x, cells, dirs, L, H, vid_nodes, arena, static_score, gi0=gi0)
    Hp = round_up4(H)
    N = x.shape[0]
    dev = x.device
    h: List[List[Optional[torch.Tensor]]] = [[None] * L for _ in range(2)]
    score = [torch.empty(N, dtype=torch.float32, device=dev) for _ in range(2)]
    gi = [torch.empty(N, 3 * Hp, dtype=torch.float32, device=dev) if d in dirs else None for d in range(2)]
    for i in range(L):
        A = [x if i == 0 else h[d][i - 1] for d in dirs]
        engine.gemm_nt_bias(A, [cells[(d, i)].w_ih for d in dirs], [cells[(d, i)].b_ih for d in dirs],
                            out=[gi[d] for d in dirs])
        pick = lambda name: [getattr(cells[(d, i)], name) if d in dirs else None for d in range(2)]  # noqa: E731
        has_gain = all(cells[(d, i)].edge_gain is not None for d in dirs)
        sc = score if static_score is None else [static_score[(d, i)] if d in dirs else None for d in range(2)]
        out = engine.recurrence_layer(plan, dirs, Hp, gi, pick("w_hh_t"), pick("b_hh"), pick("w_key"),
                                      edge_gain=pick("edge_gain") if has_gain else None,
                                      vid_bias=pick("vid_bias") if vid_nodes else None, vid_mod=vid_nodes,
                                      score=sc, static_score=static_score is not None)
        for d in dirs:
            h[d][i] = out[d]
    if Hp != H:
        return [[h[d][i][:, :H] if h[d][i] is not None else None for i in range(L)] for d in range(2)]
    return h  # type: ignore[return-value]


def check_arenas(mod) -> None:
    """Blocking device-side error check of every pass `mod` has launched (all devices / streams it ran on): raises
    `DagnnHipError` if a bounded wait of a persistent kernel expired or a batch violated the plan contract.  The
    healthy path never synchronises - `forward` only looks at finished read-backs of EARLIER passes - so a caller that
    consumes outputs without another forward behind them (the last batch of an evaluation loop,
    `ogbg-code/main_pyg.py:91-124`) calls this once where it synchronises anyway."""
    for a in list(mod._arenas.values()):
        a.check()


def default_schedule() -> str:
    import os
    s = os.environ.get("DAGNN_AMD_SCHEDULE", "lockstep")
    if s not in SCHEDULES:
        raise ValueError("DAGNN_AMD_SCHEDULE must be one of %s" % (SCHEDULES,))
    return s


def num_graphs_of(G) -> int:
    ng = getattr(G, "num_graphs", None)
    if isinstance(ng, int):
        return ng
    return int(G.batch[-1]) + 1 if G.batch.numel() else 0  # one D2H sync, like dagnn.py:137
