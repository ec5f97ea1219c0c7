"""Host-side driver of the HIP path: turns torch tensors into C-ABI calls (include/dagnn_hip.h).

PyTorch is plumbing here - device memory, the current stream, and the dense head GEMMs; every
step of the DAGNN hot path itself runs in libdagnn_hip.so.  Nothing in this module has a CPU
implementation: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import (BackwardArgs, BwdDataflowArgs, DagnnHipError, DataflowArgs, FrontierArgs, GemmGroup, LayerArgs, Plan,
                   TilesArgs, check)


class KernelTimer(object):
    """Optional HIP-event timing of individual launches (bench.py): events are recorded on the
    stream the kernel is launched on (torch's current stream), so they bracket exactly that kernel."""

    def __init__(self, only=None):
        self.spans = {}
        self.only = set(only) if only else None   # every event pair costs the stream ~8 us: time only what is asked for

    def span(self, name, device):
        if self.only is not None and name not in self.only:
            return _NoSpan()
        return _Span(self, name, device)

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1)) for k, v in self.spans.items()}


class _Span(object):
    def __init__(self, timer, name, device):
        self.timer, self.name, self.device = timer, name, device

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record(torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        self.b.record(torch.cuda.current_stream(self.device))
        self.timer.spans.setdefault(self.name, []).append((self.a, self.b))


class _NoSpan(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


TIMER: Optional[KernelTimer] = None  # set by bench.py around its timed region

def _env_int(name: str, default: int) -> int:
    return int(os.environ.get(name, default))


# Launch-geometry knobs (read once at import; the defaults are the measured optima on MI355X).
# Tests flip some of them to force the less common code paths.
RB4_MAX_WGS = _env_int("DAGNN_AMD_RB4_MAX_WGS", 0)          # 4-row vs 8-row blocks of the streamed kernel; 0 = library default
DUAL_CHAINS = _env_int("DAGNN_AMD_DUAL_CHAINS", 1)         # the two directions' per-layer launches on two streams (H > 256)
MFMA_MIN_ROWS = _env_int("DAGNN_AMD_MFMA_MIN_ROWS", 300)    # launches with at least this many rows use MFMA tiles; 0 = never
TAIL_SLICE = _env_int("DAGNN_AMD_TAIL_SLICE", 32)           # hidden units per workgroup of the persistent kernel (16 | 32)
TAIL_REPLICAS = _env_int("DAGNN_AMD_TAIL_REPLICAS", 4)      # workgroups per (cell, slice); 0 = one launch per layer throughout
TAIL_MAX_BLOCKS = _env_int("DAGNN_AMD_TAIL_MAX_BLOCKS", 2)  # without split mode: layers of <= 4 * replicas * this rows go to it
SPLIT_DEEP = _env_int("DAGNN_AMD_SPLIT_DEEP", 1)            # 1: deep graphs on a side stream from layer 0 (forward and backward)
BWD_THIN_WGS = _env_int("DAGNN_AMD_BWD_THIN_WGS", 0)        # backward: slice kernel vs rows + MFMA kernels; 0 = library default
BWD_TAIL_REPLICAS = _env_int("DAGNN_AMD_BWD_TAIL_REPLICAS", 2)    # backward persistent kernel; 0 = one launch per layer
BWD_TAIL_MAX_BLOCKS = _env_int("DAGNN_AMD_BWD_TAIL_MAX_BLOCKS", 4)
# 1: plan / schedule kernels on a side stream next to the encoder + input GEMM.  Off by default - measured on MI355X: no
# gain (2.405 vs 2.413 ms per forward); the GEMM's workgroups hold the CUs' LDS, so the 33-KB-LDS plan kernels only get
# their turn as it drains (rocprofv3 timeline: plan_graph_kernel 101 us next to the GEMM, 60 us alone).
# Tried again with the plan kernels' LDS cut to 17 / 12 KB so that they fit next to the GEMM's four workgroups per CU:
# they then run concurrently but 2-4x slower (plan_ptr 61 us instead of 15, plan_graph 103 instead of 49), bench.py
# 2.302-2.312 against 2.3085 ms - nothing - and two processes sharing one GPU (the two-rank bench test) failed.
PLAN_OVERLAP = _env_int("DAGNN_AMD_PLAN_OVERLAP", 0)              # 1: training passes launch plan / schedule on the arena's side stream next to encoder + input GEMM (model._plan_of); measured: 13 separate launches 1.64 -> 1.59 ms per forward, the fused pipeline (csrc/prepare.hip) gains nothing from it (fork + join cost what it hides: training step 5.42 against 5.32 ms)
SIDE_PRIORITY = _env_int("DAGNN_AMD_SIDE_PRIORITY", 0)     # stream priority of an arena's side stream (-1: high)
VARIANT_DATAFLOW = _env_int("DAGNN_AMD_VARIANT_DATAFLOW", 1)  # 1: evaluation passes of agg = add / max (GRU cells, no agg_x, H <= 256) run on the persistent dataflow kernel (a plain fold in its loader) instead of the per-layer variant launches
PARAM_GUARD = _env_int("DAGNN_AMD_PARAM_GUARD", 1)            # 1: evaluation passes fingerprint the parameters behind the derived-weight caches (core.ParamGuard: one small launch per pass); a write the version counters missed is reported like a device-side failure
ERR_PARAMS_MOVED = 0x10000                                       # bit of the arena's error word the guard sets
FOLD_INPUT = _env_int("DAGNN_AMD_FOLD_INPUT", 1)              # 1: evaluation passes over an ASTNodeEncoder fold the embedding tables through W_ih of stacked layer 0 once per
                                                            # weight version (gi0 = three folded rows summed per node instead of the [N, emb] x [emb, 3H] GEMM; model._folded_tables)
PREPARE_FUSED = _env_int("DAGNN_AMD_PREPARE", 1)            # 1: evaluation passes build plan + schedule + encoder rows + side effect 1 as one pipeline of 7 launches (csrc/prepare.hip)
PLAN_SMALL = _env_int("DAGNN_AMD_PLAN_SMALL", 1)            # 1: batches of <= 2048 nodes / 4096 edges / 512 graphs build plan and schedule with one workgroup each (csrc/small.hip)
PLAN_GENERAL_BUILD = 1                                      # dagnn_plan.flags: keep the plan on the general kernels
DATAFLOW = _env_int("DAGNN_AMD_DATAFLOW", 1)                # 1: the persistent graph-affine dataflow kernel where it applies (H <= 256)
DF_COST_LAYER = _env_int("DAGNN_AMD_DF_COST_LAYER", 4)      # schedule cost of one dependent layer, in rows (hop latency / row cost;
                                                            # round 3: 4 -> 6 after the block got cheaper; round 4: back to 4 after the hop did (lean loader), scripts/df_cost_sweep.py)
DF_COST_ROW = _env_int("DAGNN_AMD_DF_COST_ROW", 1)
DF_GROUPS = _env_int("DAGNN_AMD_DF_GROUPS", 0)              # 0 = as many groups as the device hosts
DF_XCD = _env_int("DAGNN_AMD_DF_XCD", 1)                    # 1: XCD-aware workgroup ids + hand-offs through the shared L2 where the run-time check allows
TILES = _env_int("DAGNN_AMD_TILES", 1)                      # 1: the weight-stationary tile kernel (H = 512: csrc/tiles.hip) where it is the faster path
                                                            # (>= 2 stacked layers: alone up to TILES_MAX_NODES nodes, behind the per-layer launches of the wide first
                                                            # layers on larger batches); 2: alone wherever it is supported; 0: never
TILES_PAD = _env_int("DAGNN_AMD_TILES_PAD", 1)              # 1: hidden sizes in (256, 512) of stacked models are zero-padded to 512 so that the tile kernel takes them
TILES_TAIL_ROWS = _env_int("DAGNN_AMD_TILES_TAIL_ROWS", 32)
TILES_MAX_MEAN_ROWS = _env_int("DAGNN_AMD_TILES_MAX_MEAN_ROWS", 160)   # policy (TILES=1): batches above 1536 nodes with more rows per batch-level layer than this stay on the launches    # larger batches: per-layer launches for the wide first layers, the tile kernel from the
                                                            # first layer on behind which no layer has more rows than this (0: no such split)
TILES_MAX_NODES = _env_int("DAGNN_AMD_TILES_MAX_NODES", 10000)  # measured on MI355X at L = 5 (scripts/tiles_sweep.py, tiles_hybrid.py): the kernel alone takes
                                                            # 0.75-0.82x the per-layer launches' time up to 8 k nodes, 0.95x at 15 k, 1.04-1.10x at 30 k
                                                            # (cfg 5); split at the thin tail 0.86x at 15 k and 0.88x at 30 k: larger batches are split
BWD_DATAFLOW = _env_int("DAGNN_AMD_BWD_DATAFLOW", 1)        # 1: the reverse sweep as one persistent dataflow launch (H <= 256)
DEBUG_WG = _env_int("DAGNN_AMD_DEBUG_WG", 0)                # workgroup whose blocks scripts/df_stamps.py stamps
SPIN_LIMIT = _env_int("DAGNN_AMD_SPIN_LIMIT", 0)            # polls before a device-side wait gives up; 0 = library default
DEBUG_TIMING: Optional[torch.Tensor] = None  # int64[8] device tensor: phase ticks of the deepest work item
_NOSPAN = _NoSpan()


def _span(name, tensor):
    return TIMER.span(name, tensor.device) if TIMER is not None else _NOSPAN


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


import threading

_LAUNCH_TLS = threading.local()   # per thread: stack of `launch_on` streams (the reference's DataParallel runs one thread per
                                  # device, tg/data_parallel.py:59-62: one thread's override must not redirect another's launches)


def _launch_stack() -> list:
    st = getattr(_LAUNCH_TLS, "stack", None)
    if st is None:
        st = _LAUNCH_TLS.stack = []
    return st


def _stream_obj(device):
    """The stream library launches go to for `device`: the innermost `launch_on` stream of this thread, else torch's current
    stream.  `_stream` (raw handle) and `persistent_launch` (events) both resolve it here."""
    st = _launch_stack()
    if st and st[-1] is not None:
        return st[-1]
    return torch.cuda.current_stream(device)


class launch_on(object):
    """Library launches inside the block go to `stream`; torch's CURRENT stream - hence the caching allocator's pool and
    every torch op - stays the caller's.  Memory allocated inside is the caller stream's: no `record_stream`, no
    foreign-pool blocks that cannot be reused (a forward pass that allocated under a side stream cost seven `hipMalloc`
    calls per batch).  The caller orders the two streams itself (fork before, join after)."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        _launch_stack().append(self.stream)
        return self

    def __exit__(self, *exc):
        _launch_stack().pop()
        return False


def _stream(t: torch.Tensor) -> int:
    """Raw hipStream_t of torch's current stream on the tensor's device (the Stream object costs ~1.5 us per call and
    the small-batch forward asks seven times)."""
    st = getattr(_LAUNCH_TLS, "stack", None)
    if st and st[-1] is not None:
        return st[-1].cuda_stream
    if _RAW_STREAM is not None:
        return _RAW_STREAM(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


_CUS = {}


class persistent_launch(object):
    """Device-wide rule for the all-resident persistent kernels (`dagnn_dataflow_run[_wide]`, `dagnn_bwd_dataflow_run[_wide]`,
    `dagnn_tiles_run`, the persistent tails of `dagnn_frontier_run` / `dagnn_backward_run`, `dagnn_encode_forward`): each of
    them sizes its grid to the whole device (one workgroup per CU, every workgroup resident, bounded spins on the others'
    granules), so two of them in flight on different streams would each hold part of the CUs and wait for workgroups
    that cannot be dispatched - the micro-batch experiment of round 4 DEADLOCKED exactly like that until the bounded
    waits expired.  Rule: inside one process, persistent launches on one device never overlap - a launch on stream B
    first waits (on the DEVICE: `hipStreamWaitEvent`, the host never blocks) for the previous persistent launch of
    another stream A to drain.  Everything else of a pass (plan, encoder, GEMMs, read-out, heads) still overlaps freely.
    Across processes nothing can enforce it: ranks must not share a GPU (multi-GPU section)."""
    _last = {}   # device index -> (event, stream handle) of the most recent persistent launch

    def __init__(self, tensor: torch.Tensor):
        self.dev = tensor.device

    def __enter__(self):
        if self.dev.type != "cuda":
            return self
        cur = _stream_obj(self.dev)   # (the stream the kernel really goes to: a `launch_on` override counts)
        rec = persistent_launch._last.get(self.dev.index)
        if rec is not None and rec[1] != cur.cuda_stream:
            cur.wait_event(rec[0])
        return self

    def __exit__(self, *exc):
        if self.dev.type != "cuda":
            return False
        cur = _stream_obj(self.dev)
        rec = persistent_launch._last.get(self.dev.index)
        ev = rec[0] if (rec is not None and rec[1] == cur.cuda_stream) else torch.cuda.Event()
        ev.record(cur)   # (a waiter that was queued on an earlier record of this event keeps that earlier record)
        persistent_launch._last[self.dev.index] = (ev, cur.cuda_stream)
        return False


def _num_cus(device) -> int:
    key = device.index if isinstance(device, torch.device) else device
    n = _CUS.get(key)
    if n is None:
        n = _CUS[key] = torch.cuda.get_device_properties(device).multi_processor_count
    return n


def _dev(t: torch.Tensor, what: str, dtype=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DagnnHipError(
            "%s must be a tensor on a ROCm GPU: the DAGNN hot path is hand-written HIP for gfx950 and has no "
            "CPU fallback (got %s)" % (what, getattr(t, "device", type(t))))
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t if t.is_contiguous() else t.contiguous()


def _rows(t: torch.Tensor, what: str) -> torch.Tensor:
    """fp32 GPU matrix whose rows are contiguous; a row pitch larger than the width is kept (views of the
    lock-step state buffers, [:, :H] of [N, H + H/16]) instead of being copied."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        return _dev(t, what, torch.float32)
    if t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1] \
            and t.data_ptr() % 16 == 0 and t.stride(0) % 4 == 0:
        return t
    return _dev(t, what, torch.float32)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


_STATUS_POOL = {}


def _status_words(device) -> torch.Tensor:
    """Four zeroed int32 words (a plan's contract status).  They come from a pool zeroed 1024 plans at a time - a fill kernel
    per plan sat on the critical path of every forward (4.4 us + its launch gap) - and a slot is never handed out twice: the
    plan's view keeps its pool alive, an exhausted pool is simply dropped."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, _RAW_STREAM(idx) if _RAW_STREAM is not None else torch.cuda.current_stream(dev).cuda_stream)
    pool = _STATUS_POOL.get(key)   # one pool per (device, stream): zeroed on, and owned by, the stream whose passes use it
    if pool is None or pool[1] >= pool[0].shape[0]:
        pool = _STATUS_POOL[key] = [torch.zeros(1024, 4, dtype=torch.int32, device=dev), 0]
    pool[1] += 1
    return pool[0][pool[1] - 1]


class PlanHandle(object):
    """Device workspace holding the layer-sorted per-graph CSR of one batch (both directions)."""

    def __init__(self, N: int, E: int, B: int, R: int, device):
        lib = _lib.load()
        self.N, self.E, self.B, self.R = int(N), int(E), int(B), int(R)
        nbytes = lib.dagnn_plan_bytes(self.N, self.E, self.B, self.R)
        self.ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=device)
        self.status = _status_words(device)
        self.desc = Plan(self.ws.data_ptr(), nbytes, self.N, self.E, self.B, self.R, 0 if PLAN_SMALL else PLAN_GENERAL_BUILD)
        self.ready = None   # event to wait for when the plan was built on another stream

    def launch_build(self) -> None:
        edge_index, layer_fwd, layer_bwd, batch, edge_attr = self._keep
        with _span("plan_build", edge_index):
            check(_lib.load().dagnn_plan_build(C.byref(self.desc), edge_index.data_ptr(), layer_fwd.data_ptr(),
                                               layer_bwd.data_ptr(), batch.data_ptr(), _ptr(edge_attr),
                                               self.status.data_ptr(), _stream(edge_index)), "dagnn_plan_build")

    def launch_prepare(self, groups: int = 0, enc=None, stack=None) -> None:
        """The fused pipeline (`dagnn_prepare`, csrc/prepare.hip): this plan, its dataflow schedule for `groups` groups
        (0: none) and the row work that rides along - `enc` = (x [N,2], depth [N], max_depth, [(type, attr, depth tables, out),
        ...]): encoder rows of up to three table sets; `stack` = (four [N] int64 tensors, out [4, N]): side effect 1."""
        lib = _lib.load()
        edge_index, layer_fwd, layer_bwd, batch, edge_attr = self._keep
        rows = _lib.PrepareRows()
        keep = []
        if enc is not None:
            x, depth, max_depth, tables = enc
            x = _dev(x, "x", torch.int64)
            if not (depth.is_cuda and depth.dtype == torch.int64 and depth.is_contiguous()):
                raise DagnnHipError("node_depth must be a contiguous int64 GPU tensor (it is clamped in place)")
            rows.x, rows.depth, rows.max_depth, rows.num_tables = x.data_ptr(), depth.data_ptr(), int(max_depth), len(tables)
            for k, (tw, aw, dw, out) in enumerate(tables):
                tw, aw, dw = _dev(tw, "type table", torch.float32), _dev(aw, "attribute table", torch.float32), _dev(dw, "depth table", torch.float32)
                keep += [tw, aw, dw]
                t = rows.table[k]
                t.type_emb, t.attr_emb, t.depth_emb, t.out = tw.data_ptr(), aw.data_ptr(), dw.data_ptr(), out.data_ptr()
                t.width, t.ld_out = tw.shape[1], out.stride(0)
            keep.append(x)
        if stack is not None:
            srcs, out = stack
            srcs = [_dev(t, "layer index", torch.int64) for t in srcs]
            keep += srcs
            for j in range(4):
                rows.stack_src[j] = srcs[j].data_ptr()
            rows.stack_out = out.data_ptr()
        ws, nbytes, key = None, 0, None
        if groups > 0:
            key = (int(groups), DF_COST_LAYER, DF_COST_ROW)
            ws = self.dataflow_schedule(groups, launch=False)
            if key in self._df:   # (it came with the plan from the loader)
                ws, groups, key = None, 0, None
            else:
                nbytes = lib.dagnn_dataflow_bytes(self.N, self.B, key[0])
        with _span("prepare", edge_index):
            check(lib.dagnn_prepare(C.byref(self.desc), edge_index.data_ptr(), layer_fwd.data_ptr(), layer_bwd.data_ptr(),
                                    batch.data_ptr(), _ptr(edge_attr), self.status.data_ptr(), _ptr(ws), nbytes, int(groups),
                                    DF_COST_LAYER, DF_COST_ROW, C.byref(rows) if (enc is not None or stack is not None) else None,
                                    _stream(edge_index)), "dagnn_prepare")
        if key is not None and key in self.__dict__.get("_df_pending", {}):
            self._df[key] = self._df_pending.pop(key)

    def wait_ready(self) -> None:
        """Order the caller's stream behind the plan's construction (no-op for a plan built on this stream)."""
        ev = getattr(self, "ready", None)
        if ev is not None:
            torch.cuda.current_stream(self.ws.device).wait_event(ev)
            self.ready = None

    @classmethod
    def from_words(cls, ws: torch.Tensor, meta: dict, dataflow_words: Optional[torch.Tensor] = None) -> "PlanHandle":
        """Wrap a plan built on the host (`dagnn_amd.host_plan.build_plan_host`, e.g. in a loader worker) and
        already moved to the GPU: no plan kernels and - the schedule being known on the host - no device->host
        read in `forward`."""
        lib = _lib.load()
        if not (isinstance(ws, torch.Tensor) and ws.is_cuda and ws.dtype == torch.int32 and ws.is_contiguous()):
            raise DagnnHipError("a host-built plan must be a contiguous int32 tensor on the GPU (move the batch "
                                "with .to(device) first)")
        self = cls.__new__(cls)
        self.N, self.E, self.B, self.R = (int(meta[k]) for k in ("N", "E", "B", "R"))
        nbytes = lib.dagnn_plan_bytes(self.N, self.E, self.B, self.R)
        if ws.numel() * 4 < nbytes:
            raise DagnnHipError("host-built plan has %d bytes, the layout needs %d" % (ws.numel() * 4, nbytes))
        self.ws = ws
        self.status = _status_words(ws.device)
        self.desc = Plan(ws.data_ptr(), nbytes, self.N, self.E, self.B, self.R, 0 if PLAN_SMALL else PLAN_GENERAL_BUILD)
        self._schedule = [a for a in meta["schedule"]]
        self._splits = [a for a in meta["splits"]]
        if dataflow_words is not None and meta.get("dataflow_key") is not None and dataflow_words.is_cuda:
            self._df_host = {"key": tuple(meta["dataflow_key"]), "words": dataflow_words}
        return self

    def layout(self) -> dict:
        off = (C.c_int64 * 26)()
        check(_lib.load().dagnn_plan_layout(self.N, self.E, self.B, self.R, off), "dagnn_plan_layout")
        names = ["node_ptr", "edge_ptr", "depth0", "depth1", "order0", "order1", "lstart0", "lstart1", "rowptr0",
                 "rowptr1", "col0", "col1", "eattr0", "eattr1", "items", "total", "blptr0", "blptr1", "rowrec0",
                 "rowrec1", "slot0", "slot1", "eidx0", "eidx1", "blsplit0", "blsplit1"]
        return {k: int(v) // 4 for k, v in zip(names, off)}

    def read_schedule(self):
        """Batch-level layer offsets of both directions as host int32 arrays (ONE device->host read,
        the counterpart of the reference's `.item()` at dagnn.py:137).  Returns [ptr_fwd, ptr_bwd],
        ptr_d has T_d + 1 entries."""
        if getattr(self, "_schedule", None) is None:
            lay = self.layout()
            a, b, c, e = lay["blptr0"], lay["blptr1"], lay["blsplit0"], lay["blsplit1"]
            host = self.ws[a:e + self.N + 2].cpu().numpy()   # blptr0 .. blsplit1 in one copy
            out, spl = [], []
            for off, soff in ((0, c - a), (b - a, e - a)):
                T = int(host[off + self.N + 1])
                out.append(host[off:off + T + 1].copy())
                spl.append(host[soff:soff + T].copy())
            self._schedule, self._splits = out, spl
        return self._schedule

    def read_splits(self):
        """Per direction, per batch-level layer: first slot of the deep graphs' rows (see include/dagnn_hip.h);
        comes with the same device->host read as the schedule."""
        self.read_schedule()
        return self._splits

    def dataflow_schedule(self, groups: int, launch: bool = True) -> torch.Tensor:
        """The plan's rows dealt to `groups` independent groups (`dagnn_dataflow_schedule`), built once per plan.
        `launch=False` allocates the workspace only; the next call with the same key issues the kernels."""
        cache = self.__dict__.setdefault("_df", {})
        pending = self.__dict__.setdefault("_df_pending", {})
        key = (int(groups), DF_COST_LAYER, DF_COST_ROW)
        if key not in cache:
            meta = getattr(self, "_df_host", None)
            if meta is not None and meta.get("key") == key:   # built by the loader (host_plan.attach_plan)
                cache[key] = meta["words"]
            else:
                lib = _lib.load()
                nbytes = lib.dagnn_dataflow_bytes(self.N, self.B, key[0])
                ws = pending.pop(key, None)
                if ws is None:
                    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=self.ws.device)
                if not launch:
                    pending[key] = ws
                    return ws
                with _span("dataflow_schedule", self.ws):
                    check(lib.dagnn_dataflow_schedule(C.byref(self.desc), ws.data_ptr(), nbytes, key[0], key[1], key[2],
                                                      self.status.data_ptr(), _stream(self.ws)), "dagnn_dataflow_schedule")
                cache[key] = ws
        return cache[key]

    def dataflow_layout(self, groups: int) -> dict:
        off = (C.c_int64 * 13)()
        check(_lib.load().dagnn_dataflow_layout(self.N, self.B, int(groups), off), "dagnn_dataflow_layout")
        names = ["grp_of", "gdepth", "gload", "loff", "gtab0", "gtab1", "lcnt0", "lcnt1", "glbase0", "glbase1", "grec0",
                 "grec1", "total"]
        return {k: int(v) // 4 for k, v in zip(names, off)}

    def check_status(self) -> None:
        """Debug helper (synchronises): raises if the batch violated the layout contract."""
        s = int(self.status[0])
        if s:
            msgs = [m for b, m in ((1, "edges not grouped by graph"), (2, "edge crosses graphs / out of range"),
                                   (4, "batch vector not sorted"), (8, "layer id >= nodes of its graph")) if s & b]
            raise DagnnHipError("plan contract violated: " + ", ".join(msgs))


def build_plan(edge_index: torch.Tensor, layer_fwd: torch.Tensor, layer_bwd: torch.Tensor, batch: torch.Tensor,
               num_graphs: int, edge_attr: Optional[torch.Tensor] = None, launch: bool = True) -> PlanHandle:
    """`launch=False`: allocate and initialise only; `plan.launch_build()` issues the kernels (e.g. under `launch_on`)."""
    edge_index = _dev(edge_index, "edge_index", torch.int64)
    layer_fwd = _dev(layer_fwd, "layer ids", torch.int64)
    layer_bwd = _dev(layer_bwd, "layer ids", torch.int64)
    batch = _dev(batch, "batch", torch.int64)
    N, E = layer_fwd.numel(), edge_index.shape[1]
    R = 0
    if edge_attr is not None:
        edge_attr = _dev(edge_attr, "edge_attr", torch.float32).view(E, -1)
        R = edge_attr.shape[1]
    plan = PlanHandle(N, E, num_graphs, R, edge_index.device)
    plan._keep = (edge_index, layer_fwd, layer_bwd, batch, edge_attr)
    if launch:
        plan.launch_build()
    return plan


def encode_ast(x: torch.Tensor, depth: torch.Tensor, type_w: torch.Tensor, attr_w: torch.Tensor,
               depth_w: torch.Tensor, max_depth: int) -> torch.Tensor:
    """out = type_emb[x0] + attr_emb[x1] + depth_emb[min(depth, max_depth)]; clamps `depth` in place."""
    x = _dev(x, "x", torch.int64)
    if not (depth.is_cuda and depth.dtype == torch.int64 and depth.is_contiguous()):
        raise DagnnHipError("node_depth must be a contiguous int64 GPU tensor (it is clamped in place)")
    N, H = x.shape[0], type_w.shape[1]
    out = torch.empty(N, H, dtype=torch.float32, device=x.device)
    check(_lib.load().dagnn_encode_ast(x.data_ptr(), depth.data_ptr(), _dev(type_w, "type table").data_ptr(),
                                       _dev(attr_w, "attribute table").data_ptr(),
                                       _dev(depth_w, "depth table").data_ptr(), int(max_depth), out.data_ptr(), H,
                                       N, H, _stream(x)), "dagnn_encode_ast")
    return out


def gemm_nt_bias(A: Sequence[torch.Tensor], W: Sequence[torch.Tensor], bias: Sequence[Optional[torch.Tensor]],
                 out: Optional[Sequence[torch.Tensor]] = None) -> List[torch.Tensor]:
    """C_g = A_g @ W_g^T + bias_g for up to 4 groups sharing M, K and Nc (one launch)."""
    n = len(A)
    A = [_rows(a, "A") for a in A]
    if len({a.stride(0) for a in A}) > 1:
        A = [a.contiguous() for a in A]
    lda = A[0].stride(0)
    W = [_dev(w, "W", torch.float32) for w in W]
    M, K = A[0].shape
    Nc = W[0].shape[0]
    for a, w in zip(A, W):
        if tuple(a.shape) != (M, K) or tuple(w.shape) != (Nc, K):
            raise DagnnHipError("grouped GEMM needs identical shapes per group")
    if out is None:
        out = [torch.empty(M, Nc, dtype=torch.float32, device=A[0].device) for _ in range(n)]
    groups = (GemmGroup * n)()
    keep = []
    for g in range(n):
        b = None if bias[g] is None else _dev(bias[g], "bias", torch.float32)
        keep.append(b)
        groups[g] = GemmGroup(A[g].data_ptr(), W[g].data_ptr(), _ptr(b), out[g].data_ptr())
    with _span("gemm_nt_bias", A[0]):
        check(_lib.load().dagnn_gemm_nt_bias(groups, n, M, Nc, K, lda, K, Nc, _stream(A[0])), "dagnn_gemm_nt_bias")
    return list(out)


def pack_whh(w_hh: torch.Tensor) -> torch.Tensor:
    """[3H, H] (torch GRUCell layout) -> k-major [H, 3H]."""
    w_hh = _dev(w_hh, "weight_hh", torch.float32)
    H = w_hh.shape[1]
    out = torch.empty(H, 3 * H, dtype=torch.float32, device=w_hh.device)
    check(_lib.load().dagnn_pack_whh(w_hh.data_ptr(), out.data_ptr(), H, _stream(w_hh)), "dagnn_pack_whh")
    return out


def pack_slices(w: torch.Tensor, H: int, slice_units: int) -> torch.Tensor:
    """[3H, K] (torch layout) -> slice/lane order consumed by the lock-step kernel."""
    w = _dev(w, "weight", torch.float32)
    K = w.shape[1]
    out = torch.empty(3 * H * K, dtype=torch.float32, device=w.device)
    check(_lib.load().dagnn_pack_slices(w.data_ptr(), out.data_ptr(), H, K, slice_units, _stream(w)),
          "dagnn_pack_slices")
    return out


def pack_mfma(w: torch.Tensor, H: int) -> torch.Tensor:
    """[3H, K] (torch layout) -> MFMA B-fragment order of the 32-row tile kernel."""
    w = _dev(w, "weight", torch.float32)
    out = torch.empty(3 * H * w.shape[1], dtype=torch.float32, device=w.device)
    check(_lib.load().dagnn_pack_mfma(w.data_ptr(), out.data_ptr(), H, w.shape[1], _stream(w)), "dagnn_pack_mfma")
    return out


def pack_batch(mats: Sequence[torch.Tensor], H: int):
    """All three lock-step layouts ({16: .., 32: .., "mfma": ..}) of every [3H, K] matrix in `mats`, one launch per 16
    matrices (`dagnn_pack_batch`)."""
    mats = [_dev(w, "weight", torch.float32) for w in mats]
    outs = []
    for base in range(0, len(mats), _lib.MAX_PACK_JOBS):
        chunk = mats[base:base + _lib.MAX_PACK_JOBS]
        jobs = (_lib.PackJob * len(chunk))()
        for j, w in zip(jobs, chunk):
            o = {k: torch.empty(3 * H * w.shape[1], dtype=torch.float32, device=w.device) for k in (16, 32, "mfma")}
            outs.append(o)
            j.w, j.out_slices16, j.out_slices32, j.out_mfma = w.data_ptr(), o[16].data_ptr(), o[32].data_ptr(), \
                o["mfma"].data_ptr()
            j.H, j.K = H, w.shape[1]
        check(_lib.load().dagnn_pack_batch(jobs, len(chunk), _stream(chunk[0])), "dagnn_pack_batch")
    return outs


def pack_dataflow(w: torch.Tensor, H: int) -> torch.Tensor:
    """[3H, H] (torch layout) -> slice / lane order of the dataflow kernel."""
    w = _dev(w, "weight", torch.float32)
    if tuple(w.shape) != (3 * H, H):
        raise DagnnHipError("pack_dataflow needs a [3H, H] matrix, got %s" % (tuple(w.shape),))
    out = torch.empty(3 * H * H, dtype=torch.float32, device=w.device)
    check(_lib.load().dagnn_pack_dataflow(w.data_ptr(), out.data_ptr(), H, _stream(w)), "dagnn_pack_dataflow")
    return out


def pack_dataflow_batch(mats, H: int, gains=()):
    """`pack_dataflow` / `pack_dataflow_transposed` of several [3H, H] matrices in ONE launch per 16: `mats` = list of (w,
    transposed); returns the packed tensors in order.  `gains`: (edge_w [kd, R], key [kd], out [R]) triples - the cells' edge
    gains `W_e^T w_key` ride in the same launch (their outputs are written in place)."""
    lib = _lib.load()
    outs, jobs, keep = [], [], []
    for w, tr in mats:
        w = _dev(w, "weight", torch.float32)
        if tuple(w.shape) != (3 * H, H):
            raise DagnnHipError("pack_dataflow_batch needs [3H, H] matrices, got %s" % (tuple(w.shape),))
        o = torch.empty(3 * H * H, dtype=torch.float32, device=w.device)
        keep.append(w)
        outs.append(o)
        jobs.append(_lib.DfPackJob(w.data_ptr(), o.data_ptr(), None, 1 if tr else 0, 0, 0))
    for ew, key, out in gains:
        ew, key = _dev(ew, "edge_encoder.weight", torch.float32), _dev(key, "key weights", torch.float32)
        keep += [ew, key]
        jobs.append(_lib.DfPackJob(ew.data_ptr(), out.data_ptr(), key.data_ptr(), 2, ew.shape[0], ew.shape[1]))
    for k0 in range(0, len(jobs), _lib.MAX_PACK_JOBS):
        part = jobs[k0:k0 + _lib.MAX_PACK_JOBS]
        arr = (_lib.DfPackJob * len(part))(*part)
        check(lib.dagnn_pack_dataflow_batch(arr, len(part), H, _stream(keep[0])), "dagnn_pack_dataflow_batch")
    return outs


RESERVED_CUS = _env_int("DAGNN_AMD_RESERVED_CUS", -1)   # CUs the persistent kernels of a TRAINING pass leave free; -1 = automatic (below)
RCCL_MAX_WORKGROUPS = 64   # upper bound of RCCL's channel count (one workgroup per channel) when NCCL_MAX_NCHANNELS does not pin it
_COLLECTIVE = {"registered": False, "group": None}


def register_collective(group=None) -> None:
    """The process group the gradient exchange of this process runs on (`train.GradBucket` / `OverlappedGradReducer` /
    `DataParallel` call this): `reserved_cus` then looks at THAT group's size instead of the default group's - a model trained
    under a one-rank sub-group overlaps no collective and keeps every CU."""
    _COLLECTIVE["registered"] = True
    _COLLECTIVE["group"] = group


def reserved_cus_info(training: bool):
    """(CUs a training pass leaves free, why) - see `reserved_cus`."""
    if RESERVED_CUS >= 0:
        return (RESERVED_CUS if training else 0), "DAGNN_AMD_RESERVED_CUS=%d" % RESERVED_CUS
    if not training:
        return 0, "inference passes overlap no collective"
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0, "no communicator"
    try:
        world = dist.get_world_size(_COLLECTIVE["group"]) if _COLLECTIVE["registered"] else dist.get_world_size()
    except (RuntimeError, ValueError):   # (this rank is not a member of the registered group)
        world = 1
    if world <= 1:
        return 0, "the gradient exchange's group has one rank"
    pinned = _env_int("NCCL_MAX_NCHANNELS", 0)
    ch = pinned if pinned > 0 else RCCL_MAX_WORKGROUPS
    r = min((ch + 7) // 8 * 8, RCCL_MAX_WORKGROUPS)   # whole CUs per XCD: the launches are spread evenly over the 8 XCDs
    return r, ("NCCL_MAX_NCHANNELS=%d pins RCCL's workgroups" % pinned if pinned > 0 else
               "RCCL may run up to %d channels (one workgroup each); NCCL_MAX_NCHANNELS pins fewer" % RCCL_MAX_WORKGROUPS)


def reserved_cus(training: bool) -> int:
    """The dataflow kernels need EVERY workgroup resident (one per CU, the whole register file of the CU each: 240 of 256
    CUs at the headline shape).  In a data-parallel training step the heads' gradient bucket is all-reduced WHILE the
    reverse sweep runs (`train.OverlappedGradReducer`): the collective's kernels (RCCL: one workgroup per channel) were
    launched first and hold CUs the sweep counts on - the resident part of the sweep then spins on granules of workgroups
    that cannot be dispatched before the collective drains: a serialisation at best, an expired bounded wait at worst.  So
    a training pass whose gradient exchange runs on a group of more than one rank (`register_collective`; the default group
    when nothing registered) sizes its persistent launches - forward and reverse share one schedule - for `num_cus - r`,
    spread evenly over the XCDs, where r = RCCL's channel count rounded up to whole CUs per XCD: `NCCL_MAX_NCHANNELS` when
    the launcher pins it (16 leaves 240 CUs = all five workgroup sets of the headline shape), else RCCL's upper bound of 64
    (24 of 32 CUs per XCD: four sets).  `DAGNN_AMD_RESERVED_CUS=n` overrides the rule in both directions (0: never
    reserve; n: always, also in a single process - what the GPU tests of the rule use).  Inference passes overlap no
    collective and keep every CU.  Reserving changes which group a graph is dealt to, never a result (GPU test)."""
    return reserved_cus_info(training)[0]


def effective_cus(device, training: bool = False) -> int:
    n = _num_cus(device)
    r = reserved_cus(training)
    return max(n - r, 8) if r > 0 else n


def dataflow_groups(device, num_dirs: int, num_stacked: int, H: int, B: int, training: bool = False) -> int:
    """Groups the dataflow kernel runs on this device for this model shape; 0 = not applicable.  `training`: a pass
    whose reverse sweep may overlap a gradient collective (`reserved_cus`)."""
    if not DATAFLOW or not dataflow_width(int(H)):
        return 0
    cus = effective_cus(device, training)
    g = _lib.load().dagnn_dataflow_groups(cus, int(num_dirs), int(num_stacked), int(H), int(B))
    return min(g, DF_GROUPS) if DF_GROUPS > 0 else g


def dataflow_run(plan: PlanHandle, dirs: Sequence[int], L: int, H: int, cells, gi0, h, groups: int, vid_mod: int = 0,
                 arena: Optional["GranuleArena"] = None, static_score=None, score_parts: bool = False,
                 preact: Optional[dict] = None, training: bool = False) -> None:
    """The whole recurrence as one persistent launch (csrc/dataflow.hip).  Same operands as `frontier_run`;
    `score_parts` adds the partial attention scores behind the state rows (what the backward pass reads); `preact`
    (a dict, training passes) receives the pre-activations the kernel computed anyway - `("gh", d, i)` [N,3H] for every
    cell and `("gi", d, i)` for the stacked layers above the first - so the reverse sweep need not recompute them."""
    if arena is None:
        raise DagnnHipError("the dataflow kernel needs a GranuleArena (persistent, zero-initialised granule buffers)")
    lib = _lib.load()
    args = dataflow_args(plan, dirs, L, H, cells, gi0, h, groups, vid_mod, arena, static_score, preact, training)
    with persistent_launch(plan.ws), _span("dataflow_run", plan.ws):
        check(lib.dagnn_dataflow_run(C.byref(plan.desc), C.byref(args), _stream(plan.ws)), "dagnn_dataflow_run")
    if score_parts and static_score is None:
        pairs = [(h[d][i], cells[(d, i)].w_key) for d in dirs for i in range(L)]   # (same row pitch, width and N: one launch per 16)
        for o in range(0, len(pairs), 16):
            part = pairs[o:o + 16]
            hp = (C.c_void_p * len(part))(*[t.data_ptr() for t, _ in part])
            wp = (C.c_void_p * len(part))(*[w.data_ptr() for _, w in part])
            check(lib.dagnn_score_parts_batch(hp, wp, len(part), part[0][0].shape[1], H, plan.N, _stream(plan.ws)), "dagnn_score_parts_batch")
    arena.watch(plan, folded=True)


def lib_static_floats(N: int, H: int) -> int:
    return int(_lib.load().dagnn_bwd_dataflow_static_bytes_h(int(N), int(H))) // 4


STAT_FWD = _env_int("DAGNN_AMD_STAT_FWD", 1)   # 1: a training pass's forward launch writes the reverse sweep's static rows itself


def stat_rows_ok(device, N: int, H: int, num_dirs: int, L: int, B: int, groups: int) -> bool:
    """Whether the forward dataflow launch of a training pass should write the reverse sweep's static rows (`stat_rows`):
    the reverse pass will be `bwd_dataflow_sweep` on the same schedule, and a node's record is addressable with 32 bits."""
    return bool(STAT_FWD) and groups > 0 and N > 0 and bwd_dataflow_groups(device, num_dirs, L, H, B) == groups and \
        bwd_dataflow_fits(device, N, num_dirs * L) and lib_static_floats(N, H) * 4 < (1 << 32)


def dataflow_args(plan: PlanHandle, dirs: Sequence[int], L: int, H: int, cells, gi0, h, groups: int, vid_mod: int = 0,
                  arena: Optional["GranuleArena"] = None, static_score=None, preact: Optional[dict] = None,
                  training: bool = False, args=None):
    """The argument struct of `dagnn_dataflow_run` for this pass (a fresh epoch of `arena`'s granule buffers, the plan's
    schedule for `groups`); `args`: fill this struct instead of a new one (the `df` member of `EncodeArgs`)."""
    if args is None:
        args = DataflowArgs()
    gld = H + H // 16
    keys = [(d, i) for d in dirs for i in range(L)] + [("p", d, i) for d in dirs for i in range(1, L)]
    # (projection granules: 16 bytes {tag, r, z, n} per unit = 2 H words of 8 bytes per node)
    gran, epoch, err = arena.get(keys, plan.N, gld, plan.ws.device, widths={k: 2 * H for k in keys if k[0] == "p"})
    mask = 0
    for d in dirs:
        mask |= 1 << d
        for i in range(L):
            c, fc = cells[(d, i)], args.cell[d][i]
            fc.w_hh = c.w_hh_df.data_ptr()
            fc.w_ih = c.w_ih_df.data_ptr() if c.w_ih_df is not None else None
            fc.b_hh, fc.b_ih = c.b_hh.data_ptr(), _ptr(c.b_ih_dev)
            if static_score is not None:
                fc.static_score = _dev(static_score[(d, i)], "static score", torch.float32).data_ptr()
            else:
                fc.w_key = c.w_key.data_ptr()
            fc.edge_gain = _ptr(c.edge_gain) if plan.R > 0 else None
            fc.vid_bias = _ptr(c.vid_bias) if vid_mod > 0 else None
            agg = getattr(c, "agg", 0)
            if agg:   # a plain aggregator (add / max / none): no keys, no gains - the edge encoder itself
                fc.agg, fc.agg_edge_w, fc.agg_edge_b = int(agg), _ptr(getattr(c, "agg_w", None)), _ptr(getattr(c, "agg_b", None))
                fc.w_key = fc.static_score = fc.edge_gain = None
            fc.gi0 = gi0[d].data_ptr() if i == 0 else None
            fc.h_out = h[d][i].data_ptr()
            fc.granules = gran[(d, i)].data_ptr()
            fc.proj_granules = gran[("p", d, i)].data_ptr() if i > 0 else None
            if preact is not None and preact.get("stat_rows"):
                # the reverse sweep's static record of every node, rows 1..7 written by this launch (bwd_dataflow_sweep adds
                # row 0); widths other than 256 / 320 leave the columns >= H to the fill
                nfl = lib_static_floats(plan.N, H)
                preact[("stat", d, i)] = (torch.empty if H in (256, 320) else torch.zeros)(nfl, dtype=torch.float32, device=plan.ws.device)
                fc.gh_out = preact[("stat", d, i)].data_ptr()
                args.stat_rows = 1
            elif preact is not None:
                preact[("gh", d, i)] = torch.empty(plan.N, 3 * H, dtype=torch.float32, device=plan.ws.device)
                fc.gh_out = preact[("gh", d, i)].data_ptr()
                if i > 0:
                    preact[("gi", d, i)] = torch.empty(plan.N, 3 * H, dtype=torch.float32, device=plan.ws.device)
                    fc.gi_out = preact[("gi", d, i)].data_ptr()
    args.num_stacked, args.dir_mask, args.H, args.ld_h, args.gld = L, mask, H, h[dirs[0]][0].shape[1], gld
    args.pld = H   # (row pitch of the projection granules, in 16-byte granules)
    args.vid_mod, args.groups, args.epoch = int(vid_mod), int(groups), epoch
    sched = plan.dataflow_schedule(groups)
    args.schedule, args.err = sched.data_ptr(), err.data_ptr()
    args.debug_timing = DEBUG_TIMING.data_ptr() if DEBUG_TIMING is not None else None
    args.spin_limit = SPIN_LIMIT
    args.debug_wg = DEBUG_WG
    if DF_XCD:
        args.num_cus = effective_cus(plan.ws.device, training)   # (the placement spreads the workgroups over num_cus / 8 per XCD)
        args.xcc_table = arena.xcc_table(plan.ws.device).data_ptr()
        args.xcd_first = arena.xcd_first
    args.plan_status = plan.status.data_ptr()
    # 64-unit slices (csrc/dataflow_x.hip): the cell variants its pair loader exists for - two edge features, keys from the states
    lean = plan.R == 2 and static_score is None and not vid_mod and \
        all(getattr(cells[(d, i)], "edge_gain", None) is not None and not getattr(cells[(d, i)], "agg", 0) for d in dirs for i in range(L))
    args.slices64 = 1 if (DF_SLICES64 and H in (256, 320) and lean) else 0
    return args


DF_SLICES64 = _env_int("DAGNN_AMD_DF_SLICES64", 0)   # 1: 64 hidden units, 8 compute waves and one stream per workgroup (csrc/dataflow_x.hip)
DF_WIDE = _env_int("DAGNN_AMD_DF_WIDE", 1)   # 1: hidden sizes 257..320 run 320 wide on the dataflow kernel's 8-wave shape (csrc/dataflow_w.hip)


def dataflow_width(Hp: int) -> bool:
    """Padded widths the forward dataflow kernel exists for: multiples of 64 up to 256, and 320 (`dagnn_dataflow_run_wide`:
    two edge features, hidden-state keys, no vertex-id key biases - `state_width` only picks 320 for such models)."""
    return Hp <= 256 or (Hp == 320 and bool(DF_WIDE))


def state_width(H: int, num_stacked: int = 1, num_edge_feats: int = 0, wide_ok: bool = False) -> int:
    """Padded width of a state row on the lock-step path: the next multiple of 64 - except that 256 < H < 512 is padded
    to 512 for a stacked model (and 384 < H for a single-layer one).  512 is the one width the tile kernel (csrc/tiles.hip)
    is built for and the one above 256 whose per-layer kernels, forward and reverse, run on MFMA tiles; padded units stay
    exactly 0 (zero weight rows and biases give r = z = 1/2, n = 0, h' = z * a = 0).  Measured (scripts/tiles_pad.py;
    DESIGN 4f): L >= 2 forward 0.46-0.98x and training step 0.46-0.99x of the unpadded width, one case of 1.06x
    (H = 300, L = 5, B = 32 forward); L = 1 wins from 448 up only, hence the second rule.  `DAGNN_AMD_TILES_PAD=0`
    keeps the next multiple of 64, `=2` pads every 256 < H < 512."""
    Hp = (int(H) + 63) // 64 * 64
    if Hp == 320 and wide_ok and DATAFLOW and DF_WIDE and num_edge_feats == 2:
        # hidden sizes 257..320 (the reference trains at 300, scripts/ogb_tok.sh:17): the dataflow kernel at H = 320 -
        # measured (round 4, B = 160, L = 2): forward 8.7 ms padded to 512 on the tile kernel / launches
        return 320
    if TILES and TILES_PAD and 256 < Hp < 512 and num_edge_feats <= 2 and (num_stacked >= 2 or Hp > 384 or TILES_PAD >= 2):
        Hp = 512
    return Hp


def tiles_launches(device, num_dirs: int, num_stacked: int, H: int, num_edge_feats: int, num_nodes: int = 0) -> int:
    """Launches the weight-stationary tile kernel makes for this model shape on this device; 0 = not applicable, or (with
    the default `DAGNN_AMD_TILES=1`) not the faster path for this depth / batch size."""
    if not TILES:
        return 0
    if TILES == 1 and (num_stacked < 2 or num_nodes > TILES_MAX_NODES):
        return 0
    return _lib.load().dagnn_tiles_launches(_num_cus(device), int(num_dirs), int(num_stacked), int(H), int(num_edge_feats))


def tiles_batch_too_flat(plan: PlanHandle, dirs: Sequence[int]) -> bool:
    """Policy (`DAGNN_AMD_TILES=1`): True for a batch of few, wide layers, which the per-layer launches' 32-row MFMA tiles
    take faster than the tile kernel's 16-row tiles gathered by all 32 slices - D-VAE BN batches at the reference's default
    width, 10 layers deep (scripts/dvae_wide.py, hs = 501, L = 2, both directions, tile kernel : launches): 32 rows per layer
    0.23 : 0.36 ms, 128: 0.46 : 0.50, 192: 0.62 : 0.61, 256: 0.77 : 0.66, 512: 1.42 : 0.99; code2 batches have ~40 rows per
    layer.  Batches of up to 1536 nodes are never too flat (and cost no read of the schedule: the tile kernel walks the
    plan's layers on the device)."""
    if TILES != 1 or TILES_MAX_MEAN_ROWS <= 0 or plan.N <= 1536:
        return False
    sched = plan.read_schedule()
    T = max([len(sched[d]) - 1 for d in dirs] + [1])
    return plan.N > TILES_MAX_MEAN_ROWS * T


def tiles_tail_split(plan: PlanHandle, dirs: Sequence[int]):
    """Where the tile kernel takes over from the per-layer launches on a large batch: per direction the first batch-level
    layer behind which no layer has more than `TILES_TAIL_ROWS` rows - the long thin tail, where a launch per layer costs
    30-50 us and a layer of the tile kernel 12.  None: no tail worth a second kernel (or the split is switched off)."""
    if TILES_TAIL_ROWS <= 0:
        return None
    import numpy as np
    sched = plan.read_schedule()
    first, tail = [0, 0], 0
    for d in dirs:
        rows = np.diff(sched[d].astype(np.int64))
        wide = np.nonzero(rows > TILES_TAIL_ROWS)[0]
        first[d] = int(wide[-1]) + 1 if len(wide) else 0
        tail = max(tail, len(rows) - first[d])
    return first if tail >= 32 else None


def tiles_run(plan: PlanHandle, dirs: Sequence[int], L: int, H: int, cells, gi0, h, arena: "GranuleArena",
              first_layer: Optional[Sequence[int]] = None, vid_mod: int = 0) -> None:
    """The whole recurrence at H = 512 as one persistent launch per chunk of stacked layers (csrc/tiles.hip): the
    weights stay in registers, the rows pass in tiles of 16.  Same operands as `frontier_run` (raw torch-layout
    matrices: nothing is packed); writes the states and the partial attention scores behind them; no device->host
    read."""
    if arena is None:
        raise DagnnHipError("the tile kernel needs a GranuleArena (persistent, zero-initialised progress counters)")
    args = TilesArgs()
    bufs, epoch, err = arena.get(["tiles"], 4096, 1, plan.ws.device)
    mask = 0
    for d in dirs:
        mask |= 1 << d
        for i in range(L):
            c, fc = cells[(d, i)], args.cell[d][i]
            fc.w_hh = c.w_hh_raw.data_ptr()
            fc.b_hh = c.b_hh.data_ptr()
            if i > 0:
                fc.w_ih, fc.b_ih = c.w_ih.data_ptr(), c.b_ih.data_ptr()
            fc.w_key = c.w_key.data_ptr()
            fc.edge_gain = _ptr(c.edge_gain) if plan.R > 0 else None
            fc.gi0 = gi0[d].data_ptr() if i == 0 else None
            fc.h_out = h[d][i].data_ptr()
            fc.vid_bias = _ptr(c.vid_bias) if vid_mod > 0 else None
    args.num_stacked, args.dir_mask, args.H, args.ld_h = L, mask, H, h[dirs[0]][0].shape[1]
    args.num_cus, args.vid_mod = _num_cus(plan.ws.device), int(vid_mod)
    args.epoch, args.counters, args.err = epoch, bufs["tiles"].data_ptr(), err.data_ptr()
    args.spin_limit = SPIN_LIMIT
    args.plan_status = plan.status.data_ptr()
    if first_layer is not None:   # the layers before these are complete already (`frontier_run(stop_layer=...)`)
        for d in dirs:
            args.first_layer[d] = int(first_layer[d])
    args.debug_timing = DEBUG_TIMING.data_ptr() if DEBUG_TIMING is not None else None
    with persistent_launch(plan.ws), _span("tiles_run", plan.ws):
        check(_lib.load().dagnn_tiles_run(C.byref(plan.desc), C.byref(args), _stream(plan.ws)), "dagnn_tiles_run")
    arena.watch(plan, folded=True)


def frontier_ld(H: int) -> int:
    """Row pitch of the lock-step state buffers: H states + H/16 partial scores, 16-byte multiple."""
    return H + (H // 16 + 3) // 4 * 4


class GranuleArena(object):
    """Persistent, zero-initialised granule buffers (tagged 8-byte copies of the state rows) of one
    module on one device, plus the strictly increasing epoch that tags a forward pass.  They must
    outlive single calls: a tag is only meaningful against memory that never held a larger one."""

    def __init__(self):
        self.bufs = {}
        self.cap = 0
        self.gld = 0
        self.epoch = 0
        self.err = None
        self.side = None   # second stream: the persistent kernel runs next to the per-layer launches (split mode)
        self.xcd_first = 0   # XCD the dataflow launches of this arena pack their workgroups from (arenas of further streams: 2, 4, ..)

    def get(self, keys, N: int, gld: int, device, widths=None):
        """`widths`: row pitch (granules) of the buffers that differ from `gld`."""
        widths = widths or {}
        if N > self.cap or gld != self.gld or self.err is None or self.err.device != device or \
                set(keys) != set(self.bufs) or self.epoch >= 0x7FFFFFF0:
            self.cap, self.gld, self.epoch = max(N, int(self.cap * 1.5)), gld, 0
            self.bufs = {k: torch.zeros(self.cap * widths.get(k, gld), dtype=torch.int64, device=device) for k in keys}
            self.err = torch.zeros(1, dtype=torch.int32, device=device)
        self.epoch += 1
        return self.bufs, self.epoch, self.err

    def xcc_table(self, device):
        """Tagged table the workgroups of a dataflow launch publish their XCD in (zero-initialised once, tagged with the
        arena's epochs like the granule buffers; re-created with them)."""
        t = getattr(self, "_xcc", None)
        if t is None or t.device != device or getattr(self, "_xcc_gen", None) is not self.err:
            t = self._xcc = torch.zeros(1024, dtype=torch.int64, device=device)
            self._xcc_gen = self.err   # (the arena restarts its epochs whenever it re-creates `err`)
        return t

    def side_stream(self, device):
        """Second stream for the split mode (created once): the persistent kernel runs on it next to the per-layer
        launches of the caller's stream."""
        if self.side is None or self.side.device != device:
            self.side = torch.cuda.Stream(device, priority=SIDE_PRIORITY)
        return self.side

    def fork_join_events(self, device):
        """The two events a split-mode call forks and joins with (raw hipEvent_t handles).  They belong to the arena:
        the library itself creates and destroys nothing (include/dagnn_hip.h, conventions)."""
        evs = getattr(self, "_fj", None)
        if evs is None or evs[2] != device:
            a, b = torch.cuda.Event(), torch.cuda.Event()
            st = torch.cuda.current_stream(device)
            a.record(st)   # (torch creates the underlying event at the first record)
            b.record(st)
            evs = self._fj = (a, b, device)
        return evs[0].cuda_event, evs[1].cuda_event

    WATCH_SLOTS = 64   # read-backs in flight before the oldest one is waited for

    def watch(self, plan=None, folded: bool = False) -> None:
        """Queue an asynchronous read-back of the device-side error words (the kernels' bounded-wait flag and the
        plan's contract status) behind the work just launched; `poll()` looks at the finished ones without
        synchronising.  Every watch owns a slot of a pinned ring and an event: an async evaluation loop that issues
        many passes before anything synchronises loses none of their reports (one shared buffer would let a later clean
        pass overwrite a violation).  `folded`: the launch was one of the dataflow kernels, which carry the plan's status
        in bits 8-15 of the error word themselves - one copy instead of two."""
        import collections
        dev = self.err.device
        if getattr(self, "_host", None) is None:
            self._host = torch.zeros(self.WATCH_SLOTS, 2, dtype=torch.int32).pin_memory()
            self._host_err = [self._host[k, 0:1] for k in range(self.WATCH_SLOTS)]
            self._host_status = [self._host[k, 1:2] for k in range(self.WATCH_SLOTS)]
            self._events = [None] * self.WATCH_SLOTS
            self._pending = collections.deque()
            self._seq = 0
        if len(self._pending) >= self.WATCH_SLOTS:
            self._drain(block_first=True)   # (raises if that oldest pass failed)
        slot = self._seq % self.WATCH_SLOTS
        self._seq += 1
        self._host_err[slot].copy_(self.err, non_blocking=True)
        if plan is not None and not folded:
            self._host_status[slot].copy_(plan.status[0:1], non_blocking=True)
        else:
            self._host_status[slot].zero_()
        ev = self._events[slot]   # (the slot's previous watch has been drained: its event is free again)
        if ev is None:
            ev = self._events[slot] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._pending.append((ev, slot))

    def _drain(self, block: bool = False, block_first: bool = False) -> None:
        pend = getattr(self, "_pending", None)
        failed = None
        while pend:
            ev, slot = pend[0]
            if block or block_first:
                ev.synchronize()
                block_first = False
            elif not ev.query():
                break
            pend.popleft()
            e, s = int(self._host[slot, 0]), int(self._host[slot, 1])
            if e & 0xff00:   # a dataflow kernel found the plan's status word set and reported it in bits 8-15
                s, e = s | ((e >> 8) & 0xff), e & ~0xff00
            if (e or s) and failed is None:
                failed = (e, s)
        if failed is None:
            return
        e, s = failed
        if self.err is not None:
            self.err.zero_()   # the flag is sticky on the device (a lost pass makes later ones give up early): consumed here
        msgs = []
        if e & ERR_PARAMS_MOVED:
            msgs.append("a parameter changed without its version counter moving (a fused optimizer step, a write through "
                        ".data) while the module was in evaluation mode: the pass read derived weights cached from the OLD "
                        "values - call module.train() / .eval() (or .invalidate_caches()) after such an update")
            e &= ~ERR_PARAMS_MOVED
            if not e and not s:
                raise DagnnHipError("results of an earlier DAGNN pass are invalid: " + msgs[0])
        if e & 3:
            msgs.append("a bounded device-side wait expired (code %d): the persistent kernel's workgroups were not "
                        "co-resident, or a producer failed" % e)
        if e & 8:
            msgs.append("the dataflow schedule handed to the kernel was built for another group count (or not built)")
        if (e & 4) and not s:
            msgs.append("the persistent kernel found the plan's status word set (the batch violates the plan contract)")
        if s:
            msgs.append("the batch violates the plan contract (status %d: 1 edges not grouped by graph, 2 edge "
                        "crosses graphs / out of range, 4 batch vector not sorted, 8 layer id out of range)" % s)
        if not msgs:
            msgs.append("device-side error word %d" % e)
        raise DagnnHipError("results of an earlier DAGNN pass are invalid: " + "; ".join(msgs))

    def poll(self, block: bool = False) -> None:
        """Raise `DagnnHipError` if a finished earlier pass reported a device-side failure.  Without `block` this only
        looks at read-backs that have already completed (no synchronisation); with it, at every pass launched so far."""
        self._drain(block=block)

    def check(self) -> None:
        """Synchronising check of every pass launched so far (bounded waits and plan contract)."""
        self._drain(block=True)
        if self.err is not None and int(self.err[0]):
            e = int(self.err[0])
            self.err.zero_()
            if e == ERR_PARAMS_MOVED:
                raise DagnnHipError("a parameter changed without its version counter moving while the module was in evaluation "
                                    "mode: the last pass read stale derived weights (call module.train() / .eval() after such an update)")
            raise DagnnHipError("persistent kernel: a bounded wait expired (results are invalid)")


def frontier_run(plan: PlanHandle, dirs: Sequence[int], L: int, H: int, cells, gi0, h, vid_mod: int = 0,
                 arena: Optional[GranuleArena] = None, static_score=None, stop_layer: Optional[Sequence[int]] = None,
                 chains: Optional[GranuleArena] = None) -> None:
    """Lock-step recurrence over all batch-level layers.  `cells[(d, i)]` are kernel-ready parameter
    holders (core.CellParams); gi0[d] [N,3H]; h[d][i] [N, frontier_ld(H)] outputs.  `chains`: an arena whose side stream
    and events the call may use to run the two directions' launches as two independent chains (no persistent tail in
    the call: `arena` is None or the width has no tail kernel)."""
    args = FrontierArgs()
    mask = 0
    use_tail = arena is not None and TAIL_REPLICAS > 0 and H <= 256
    gran, epoch, err = arena.get([(d, i) for d in dirs for i in range(L)], plan.N, H + H // 16, plan.ws.device) \
        if use_tail else ({}, 0, None)
    for d in dirs:
        mask |= 1 << d
        for i in range(L):
            c, fc = cells[(d, i)], args.cell[d][i]
            fc.granules = gran[(d, i)].data_ptr() if use_tail else None
            fc.w_hh_pk16, fc.w_hh_pk32 = c.w_hh_pk[16].data_ptr(), c.w_hh_pk[32].data_ptr()
            fc.w_hh_mfma = c.w_hh_pk["mfma"].data_ptr()
            if c.w_ih_pk is not None:
                fc.w_ih_pk16, fc.w_ih_pk32 = c.w_ih_pk[16].data_ptr(), c.w_ih_pk[32].data_ptr()
                fc.w_ih_mfma = c.w_ih_pk["mfma"].data_ptr()
            fc.b_hh, fc.b_ih = c.b_hh.data_ptr(), _ptr(c.b_ih_dev)
            if static_score is not None:
                fc.static_score = _dev(static_score[(d, i)], "static score", torch.float32).data_ptr()
            else:
                fc.w_key = c.w_key.data_ptr()
            fc.edge_gain = _ptr(c.edge_gain) if plan.R > 0 else None
            fc.vid_bias = _ptr(c.vid_bias) if vid_mod > 0 else None
            fc.gi0 = gi0[d].data_ptr() if i == 0 else None
            fc.h_out = h[d][i].data_ptr()
    args.num_stacked, args.dir_mask, args.H, args.ld_h, args.vid_mod = L, mask, H, h[dirs[0]][0].shape[1], int(vid_mod)
    args.debug_timing = DEBUG_TIMING.data_ptr() if DEBUG_TIMING is not None else None
    args.num_cus = _num_cus(plan.ws.device)
    args.rb4_max_wgs = RB4_MAX_WGS
    # everything above is independent of the schedule: the one device->host read of the forward pass comes
    # last, so the host-side argument marshalling overlaps the plan / GEMM kernels still in flight
    sched = plan.read_schedule()
    if stop_layer is not None:   # only the batch-level layers before these (the rest: `tiles_run(first_layer=...)`)
        sched = [s_[:min(len(s_), int(stop_layer[d]) + 1)] for d, s_ in enumerate(sched)]
    args.mfma_min_rows = MFMA_MIN_ROWS
    if MFMA_MIN_ROWS > 0:  # fat launches (csrc/fat.hip): scratch rows for the aggregates of rows with more than 4 predecessors
        import numpy as np
        nst = max(len(sched[0]), len(sched[1])) - 1 + L
        widest = 0
        for d in dirs:   # per direction: the two directions' launches may run as two chains, each with its own rows
            width = np.zeros(nst, dtype=np.int64)
            w = np.diff(sched[d].astype(np.int64))
            for i in range(L):  # layer t of stacked layer i runs in launch t + i
                width[i:i + len(w)] += w
            widest += int(width.max())
        plan.agg_scratch = torch.empty(max(widest, 1) * H, dtype=torch.float32, device=plan.ws.device)
        args.agg_scratch, args.agg_scratch_rows = plan.agg_scratch.data_ptr(), widest
    args.tail_replicas, args.tail_max_blocks = (TAIL_REPLICAS if use_tail else 0), TAIL_MAX_BLOCKS
    args.tail_slice_units = TAIL_SLICE
    args.epoch, args.tail_err = epoch, _ptr(err)
    ptrs = (C.POINTER(C.c_int32) * 2)()
    nl = (C.c_int32 * 2)()
    for d in (0, 1):
        ptrs[d] = sched[d].ctypes.data_as(C.POINTER(C.c_int32))
        nl[d] = len(sched[d]) - 1
    if chains is not None and not use_tail and len(dirs) == 2 and DUAL_CHAINS and DEBUG_TIMING is None:
        args.side_stream = chains.side_stream(plan.ws.device).cuda_stream
        args.fork_event, args.join_event = chains.fork_join_events(plan.ws.device)
    if use_tail and SPLIT_DEEP and DEBUG_TIMING is None:
        splits = plan.read_splits()
        args.side_stream = arena.side_stream(plan.ws.device).cuda_stream
        args.fork_event, args.join_event = arena.fork_join_events(plan.ws.device)
        for d in dirs:
            args.layer_split[d] = splits[d].ctypes.data_as(C.POINTER(C.c_int32))
    with persistent_launch(plan.ws), _span("frontier_run", plan.ws):
        check(_lib.load().dagnn_frontier_run(C.byref(plan.desc), C.byref(args), ptrs, nl, _stream(plan.ws)),
              "dagnn_frontier_run")
    if arena is not None and arena.err is not None:
        arena.watch(plan)   # the persistent tail's bounded waits and the plan's status word, like the dataflow path


def recurrence_layer(plan: PlanHandle, dirs: Sequence[int], H: int, gi, w_hh_t, b_hh, w_key, edge_gain=None,
                     vid_bias=None, vid_mod: int = 0, out=None, score=None,
                     static_score: bool = False) -> List[Optional[torch.Tensor]]:
    """One stacked GRU layer over all topological layers.  Per-direction lists indexed by d."""
    dev = plan.ws.device
    h = [None, None]
    args = LayerArgs()
    keep = []
    mask = 0
    for d in dirs:
        mask |= 1 << d
        h[d] = out[d] if out is not None else torch.empty(plan.N, H, dtype=torch.float32, device=dev)
        sc = score[d] if score is not None else torch.empty(plan.N, dtype=torch.float32, device=dev)
        ts = [_dev(gi[d], "gi", torch.float32), _dev(w_hh_t[d], "w_hh_t", torch.float32),
              _dev(b_hh[d], "b_hh", torch.float32),
              sc if static_score else _dev(w_key[d], "w_key", torch.float32)]
        eg = _dev(edge_gain[d], "edge_gain", torch.float32) if (edge_gain is not None and plan.R > 0) else None
        vb = _dev(vid_bias[d], "vid_bias", torch.float32) if (vid_bias is not None and vid_mod > 0) else None
        keep += ts + [eg, vb, sc]
        args.gi[d], args.w_hh_t[d], args.b_hh[d], args.w_key[d] = (t.data_ptr() for t in ts)
        args.edge_gain[d], args.vid_bias[d] = _ptr(eg), _ptr(vb)
        args.h[d], args.score[d] = h[d].data_ptr(), sc.data_ptr()
    args.vid_mod, args.ld_h, args.static_score = int(vid_mod), H, int(static_score)
    args.debug_timing = DEBUG_TIMING.data_ptr() if DEBUG_TIMING is not None else None
    with _span("recurrence_layer", plan.ws):
        check(_lib.load().dagnn_recurrence_layer(C.byref(plan.desc), C.byref(args), mask, H, _stream(plan.ws)),
              "dagnn_recurrence_layer")
    return h


def readout_max(plan: PlanHandle, h: torch.Tensor, direction: int, out: torch.Tensor, col_off: int) -> None:
    h = _rows(h, "h")
    check(_lib.load().dagnn_readout_max(C.byref(plan.desc), h.data_ptr(), h.stride(0), h.shape[1], direction,
                                        out.data_ptr(), out.shape[1], col_off, _stream(h)), "dagnn_readout_max")


def readout_max_batch(plan: PlanHandle, jobs, out: torch.Tensor) -> None:
    """`readout_max` for several (h, direction, col_off) in one launch."""
    arr = (_lib.ReadoutJob * len(jobs))()
    keep = []
    for k, (h, direction, col_off) in enumerate(jobs):
        h = _rows(h, "h")
        keep.append(h)
        arr[k].h, arr[k].ld_h, arr[k].width, arr[k].dir, arr[k].col_off = h.data_ptr(), h.stride(0), h.shape[1], int(direction), int(col_off)
    check(_lib.load().dagnn_readout_max_batch(C.byref(plan.desc), arr, len(jobs), out.data_ptr(), out.shape[1], _stream(out)),
          "dagnn_readout_max_batch")


def readout_pool(plan: PlanHandle, h: torch.Tensor, scope: int, how: str, out: torch.Tensor, col_off: int) -> None:
    """out[:, col_off : col_off + width] = max / add / mean pool of h over scope 0 / 1 (output nodes of that direction)
    or 2 (all nodes of the graph)."""
    h = _rows(h, "hidden states")
    mode = {"max": _lib.POOL_MAX, "add": _lib.POOL_ADD, "sum": _lib.POOL_ADD, "mean": _lib.POOL_MEAN}[how]
    check(_lib.load().dagnn_readout_pool(C.byref(plan.desc), h.data_ptr(), h.stride(0), h.shape[1], int(scope), mode,
                                         out.data_ptr(), out.stride(0), int(col_off), _stream(h)), "dagnn_readout_pool")


def readout_max_backward(plan: PlanHandle, h: torch.Tensor, direction: int, grad_out: torch.Tensor, col_off: int,
                         grad_h: torch.Tensor) -> None:
    """grad_h[v, :] += grad_out[g, col_off : col_off + width] at the arg-max output node of every graph/column."""
    h = _rows(h, "h")
    grad_out = _dev(grad_out, "grad_out", torch.float32)
    check(_lib.load().dagnn_readout_max_backward(C.byref(plan.desc), h.data_ptr(), h.stride(0), h.shape[1], direction,
                                                 grad_out.data_ptr(), grad_out.shape[1], col_off, grad_h.data_ptr(),
                                                 grad_h.stride(0), _stream(h)), "dagnn_readout_max_backward")


def readout_max_backward_batch(plan: PlanHandle, jobs, grad_out: torch.Tensor) -> None:
    """`readout_max_backward` for several (h, direction, col_off, grad_h) jobs in ONE launch (distinct `grad_h` each)."""
    if not jobs:
        return
    grad_out = _dev(grad_out, "grad_out", torch.float32)
    lib = _lib.load()
    for k0 in range(0, len(jobs), _lib.MAX_READOUT_JOBS):
        part = jobs[k0:k0 + _lib.MAX_READOUT_JOBS]
        arr = (_lib.ReadoutBwdJob * len(part))()
        for q, (h, d, col, gh) in enumerate(part):
            h = _rows(h, "h")
            arr[q] = _lib.ReadoutBwdJob(h.data_ptr(), gh.data_ptr(), h.stride(0), gh.stride(0), h.shape[1], int(d), int(col))
        check(lib.dagnn_readout_max_backward_batch(C.byref(plan.desc), arr, len(part), grad_out.data_ptr(), grad_out.shape[1],
                                                   _stream(grad_out)), "dagnn_readout_max_backward_batch")


def attn_grads(jobs):
    """Gradients of attn_lin.weight / edge_encoder.{weight, bias} of every cell from the epilogue's column sums, ONE launch
    (`dagnn_attn_grads_run`).  `jobs`: list of dicts {key_sum [kd], feat_sum [R] | None, sigma_sum [1] | None, edge_w [kd, R] |
    None, edge_b, attn_w [1, attn_len], dq}; returns a list of (g_attn [1, attn_len], g_edge_w | None, g_edge_b | None) - views of
    one buffer."""
    lib = _lib.load()
    dev = jobs[0]["attn_w"].device
    sizes = []
    for j in jobs:
        kd = j["key_sum"].numel()
        R = j["edge_w"].shape[1] if j["edge_w"] is not None else 0
        sizes.append((j["attn_w"].shape[1], kd * R, kd if R else 0))
    buf = torch.empty(sum(a + b + c for a, b, c in sizes), dtype=torch.float32, device=dev)
    out, off = [], 0
    arr = (_lib.AttnGradJob * len(jobs))()
    keep = []
    for q, (j, (a, b, c)) in enumerate(zip(jobs, sizes)):
        g_attn = buf[off:off + a].view(1, a); off += a
        g_ew = buf[off:off + b].view(-1, j["edge_w"].shape[1]) if b else None; off += b
        g_eb = buf[off:off + c] if c else None; off += c
        t = {k: (None if j[k] is None else j[k].detach().contiguous()) for k in ("key_sum", "feat_sum", "sigma_sum", "edge_w", "edge_b", "attn_w")}
        keep.append(t)
        kd = t["key_sum"].numel()
        arr[q] = _lib.AttnGradJob(t["key_sum"].data_ptr(), _ptr(t["feat_sum"]), _ptr(t["sigma_sum"]), _ptr(t["edge_w"]), _ptr(t["edge_b"]),
                                  t["attn_w"].data_ptr(), g_attn.data_ptr(), _ptr(g_ew), _ptr(g_eb), int(j["dq"]), kd, a,
                                  0 if t["edge_w"] is None else t["edge_w"].shape[1])
        out.append((g_attn, g_ew, g_eb))
    for k0 in range(0, len(jobs), _lib.ATTN_GRAD_MAX_JOBS):
        n = min(_lib.ATTN_GRAD_MAX_JOBS, len(jobs) - k0)
        sub = (_lib.AttnGradJob * n)(*[arr[k0 + i] for i in range(n)])
        check(lib.dagnn_attn_grads_run(sub, n, _stream(buf)), "dagnn_attn_grads_run")
    return out


def backward_sweep(plan: PlanHandle, dirs: Sequence[int], L: int, H: int, cells, h, gi0, g_ext,
                   arena: Optional[GranuleArena] = None, vid_mod: int = 0, static_score=None):
    """Reverse pass of the lock-step recurrence (csrc/backward.hip).  `h[d][i]` [N, frontier_ld(H)] are the
    forward state buffers, `gi0[d]` [N,3H] the input-side pre-activations of stacked layer 0, `g_ext[d][i]`
    [N,H] the gradients reaching the states from outside (modified: stacked layers below the top receive the
    upper layer's input gradient).  Returns per cell (d, i) a dict with a, dgi, dgh, sigma, edge_feat_grad."""
    dev = plan.ws.device
    N, E, R = plan.N, plan.E, plan.R
    args = BackwardArgs()
    out = {}
    mask = 0
    f32 = dict(dtype=torch.float32, device=dev)
    use_tail = arena is not None and BWD_TAIL_REPLICAS > 0 and H <= 256
    gkeys = [("da", d, i) for d in dirs for i in range(L)] + [("du", d, i) for d in dirs for i in range(L - 1)]
    gran, epoch, err = arena.get(gkeys, N, H, dev) if use_tail else ({}, 0, None)
    keep = []
    keep_alive = keep
    if arena is not None:
        arena.poll()
    for d in dirs:
        mask |= 1 << d
        for i in range(L):
            c, bc = cells[(d, i)], args.cell[d][i]
            if use_tail:
                bc.da_granules = gran[("da", d, i)].data_ptr()
                if i + 1 < L:   # the sweep adds the upper layer's du into g_ext: keep what it was before
                    static = g_ext[d][i].clone()
                    keep.append(static)
                    bc.du_granules, bc.g_ext_static = gran[("du", d, i)].data_ptr(), static.data_ptr()
            o = dict(a=torch.empty(N, H, **f32), alpha=torch.empty(max(E, 1), **f32), da=torch.empty(N, H, **f32),
                     dgi=torch.empty(N, 3 * H, **f32), dgh=torch.empty(N, 3 * H, **f32), sigma=torch.empty(N, **f32),
                     edge_feat_grad=torch.empty(N, R, **f32) if R > 0 else None)
            out[(d, i)] = o
            bc.w_hh, bc.w_ih = c.w_hh_raw.data_ptr(), (c.w_ih.data_ptr() if i > 0 else None)
            if static_score is not None:   # keys from the inputs: scores are given, the states' gradient gets no key term
                zero_key = torch.zeros(H, **f32)
                keep_alive.append(zero_key)
                bc.w_key, bc.static_score = zero_key.data_ptr(), _dev(static_score[(d, i)], "static score", torch.float32).data_ptr()
            else:
                bc.w_key = c.w_key.data_ptr()
            bc.edge_gain = _ptr(c.edge_gain) if R > 0 else None
            bc.vid_bias = _ptr(c.vid_bias) if vid_mod > 0 else None
            bc.h, bc.a, bc.alpha = h[d][i].data_ptr(), o["a"].data_ptr(), o["alpha"].data_ptr()
            bc.g_ext, bc.da, bc.dgi, bc.dgh = g_ext[d][i].data_ptr(), o["da"].data_ptr(), o["dgi"].data_ptr(), o["dgh"].data_ptr()
            bc.sigma, bc.edge_feat_grad = o["sigma"].data_ptr(), _ptr(o["edge_feat_grad"])
    args.num_stacked, args.dir_mask, args.H, args.ld_h = L, mask, H, h[dirs[0]][0].shape[1]
    args.vid_mod = int(vid_mod)
    args.num_cus = _num_cus(dev)
    args.thin_wgs = BWD_THIN_WGS
    args.tail_replicas, args.tail_max_blocks = (BWD_TAIL_REPLICAS if use_tail else 0), BWD_TAIL_MAX_BLOCKS
    args.epoch, args.tail_err = epoch, _ptr(err)
    lib = _lib.load()
    with _span("backward_prepare", plan.ws):
        check(lib.dagnn_backward_prepare(C.byref(plan.desc), C.byref(args), _stream(plan.ws)), "dagnn_backward_prepare")
    # pre-activations of every cell, recomputed by batched MFMA GEMMs (all rows at once: every h is known)
    keys = [(d, i) for d in dirs for i in range(L)]
    gh = {}
    for k0 in range(0, len(keys), 4):
        grp = keys[k0:k0 + 4]
        res = gemm_nt_bias([out[k]["a"] for k in grp], [cells[k].w_hh_raw for k in grp], [cells[k].b_hh for k in grp])
        gh.update(dict(zip(grp, res)))
    gi = {(d, 0): gi0[d] for d in dirs}
    up = [(d, i) for d in dirs for i in range(1, L)]
    for k0 in range(0, len(up), 4):
        grp = up[k0:k0 + 4]
        res = gemm_nt_bias([h[d][i - 1][:, :H] for d, i in grp], [cells[k].w_ih for k in grp],
                           [cells[k].b_ih for k in grp])
        gi.update(dict(zip(grp, res)))
    for k in keys:
        args.cell[k[0]][k[1]].gi, args.cell[k[0]][k[1]].gh = gi[k].data_ptr(), gh[k].data_ptr()
        out[k]["gi"], out[k]["gh"] = gi[k], gh[k]
    sched = plan.read_schedule()
    ptrs = (C.POINTER(C.c_int32) * 2)()
    nl = (C.c_int32 * 2)()
    for d in (0, 1):
        ptrs[d] = sched[d].ctypes.data_as(C.POINTER(C.c_int32))
        nl[d] = len(sched[d]) - 1
    if use_tail and SPLIT_DEEP:
        splits = plan.read_splits()
        args.side_stream = arena.side_stream(dev).cuda_stream
        args.fork_event, args.join_event = arena.fork_join_events(dev)
        for d in dirs:
            args.layer_split[d] = splits[d].ctypes.data_as(C.POINTER(C.c_int32))
    with persistent_launch(plan.ws), _span("backward_run", plan.ws):
        check(lib.dagnn_backward_run(C.byref(plan.desc), C.byref(args), ptrs, nl, _stream(plan.ws)),
              "dagnn_backward_run")
    if use_tail:
        arena.watch()
    return out


def pack_dataflow_transposed(w: torch.Tensor, H: int) -> torch.Tensor:
    """[3H, H] (torch layout) -> the dataflow kernel's slice / lane order of the GATE-WISE TRANSPOSED matrix
    (out[g H + j][u] = W[g H + u][j]): the A operands of the reverse products W^T dg."""
    w = _dev(w, "weight", torch.float32)
    if tuple(w.shape) != (3 * H, H):
        raise DagnnHipError("pack_dataflow_transposed needs a [3H, H] matrix, got %s" % (tuple(w.shape),))
    out = torch.empty(3 * H * H, dtype=torch.float32, device=w.device)
    check(_lib.load().dagnn_pack_dataflow_transposed(w.data_ptr(), out.data_ptr(), H, _stream(w)),
          "dagnn_pack_dataflow_transposed")
    return out


def bwd_dataflow_groups(device, num_dirs: int, num_stacked: int, H: int, B: int) -> int:
    """Groups the reverse dataflow launch runs with (same cell count and workgroup shape as the forward kernel, so the
    forward pass's schedule workspace serves both); 0 = not applicable."""
    if not BWD_DATAFLOW or not dataflow_width(int(H)):
        return 0
    return dataflow_groups(device, num_dirs, num_stacked, H, B, training=True)


# cap on what the reverse dataflow sweep may keep: unset (-1) = a third of the device's memory (96 GB on an MI355X); a value is
# taken as it is - 0 switches the reverse dataflow sweep off (every training pass then takes the reverse lock-step launches)
BWD_DF_MAX_BYTES = _env_int("DAGNN_AMD_BWD_DF_MAX_BYTES", -1)


_TOTAL_MEMORY = {}


def _total_memory(device) -> int:
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _TOTAL_MEMORY:
        _TOTAL_MEMORY[key] = int(torch.cuda.get_device_properties(key).total_memory)
    return _TOTAL_MEMORY[key]


def bwd_dataflow_fits(device, N: int, cells: int) -> bool:
    """The persistent reverse sweep keeps ONE 8 KB (H = 320: 10 KB) static record per (cell, node)
    (`dagnn_bwd_dataflow_static_bytes`) whatever H is, plus - about as much again, twice at H = 256 - the 256-byte
    successor records, the hand-off granules (`da`, `dgi`, `du`) and the forward pass's pre-activations: ~1.5 GB for the
    headline batch, tens of GB for a very large batch at H = 64.  The choice between the sweep and the reverse
    lock-step launches (`backward_sweep`, which need none of it) is a pure function of (device model, N, cells): the byte cap
    is a third of the device's TOTAL memory (96 GB on an MI355X; `DAGNN_AMD_BWD_DF_MAX_BYTES` overrides it, 0 = never) - not
    of the memory that happens to be free - so every rank and every step takes the same reverse path and gradients stay
    bitwise reproducible run to run, and a smaller device falls back to the launches instead of running out of memory."""
    need = int(_lib.load().dagnn_bwd_dataflow_static_bytes(int(N))) * int(cells) * 3   # records x 1.25 (H = 320) + granules + preact
    cap = BWD_DF_MAX_BYTES
    if cap < 0:
        cap = _total_memory(device) // 3
    return need <= cap


def bwd_dataflow_sweep(plan: PlanHandle, dirs: Sequence[int], L: int, H: int, cells, h, gi0, g_ext, groups: int,
                       arena: "GranuleArena", vid_mod: int = 0, static_score=None, preact: Optional[dict] = None):
    """Reverse pass of the recurrence as ONE persistent dataflow launch (csrc/bwd_dataflow.hip).  Same operands and
    the same result dictionary as `backward_sweep`; `groups` must be the group count of the forward pass's schedule."""
    dev = plan.ws.device
    N, E, R = plan.N, plan.E, plan.R
    lib = _lib.load()
    f32 = dict(dtype=torch.float32, device=dev)
    arena.poll()
    out, keep = {}, []
    pargs = BackwardArgs()
    mask = 0
    for d in dirs:
        mask |= 1 << d
        for i in range(L):
            c, bc = cells[(d, i)], pargs.cell[d][i]
            o = dict(a=torch.empty(N, H, **f32), alpha=torch.empty(max(E, 1), **f32),
                     dgi=torch.empty(N, 3 * H, **f32), dgh=torch.empty(N, 3 * H, **f32), sigma=torch.empty(N, **f32),
                     edge_feat_grad=torch.empty(N, R, **f32) if R > 0 else None)
            out[(d, i)] = o
            if static_score is not None:
                zero_key = torch.zeros(H, **f32)
                keep.append(zero_key)
                bc.w_key, bc.static_score = zero_key.data_ptr(), _dev(static_score[(d, i)], "static score", torch.float32).data_ptr()
                o["_wkey"] = zero_key
            else:
                bc.w_key = c.w_key.data_ptr()
                o["_wkey"] = c.w_key
            bc.edge_gain = _ptr(c.edge_gain) if R > 0 else None
            bc.vid_bias = _ptr(c.vid_bias) if vid_mod > 0 else None
            bc.h, bc.a, bc.alpha = h[d][i].data_ptr(), o["a"].data_ptr(), o["alpha"].data_ptr()
    pargs.num_stacked, pargs.dir_mask, pargs.H, pargs.ld_h = L, mask, H, h[dirs[0]][0].shape[1]
    pargs.vid_mod = int(vid_mod)
    with _span("backward_prepare", plan.ws):
        check(lib.dagnn_backward_prepare(C.byref(plan.desc), C.byref(pargs), _stream(plan.ws)), "dagnn_backward_prepare")
        keys = [(d, i) for d in dirs for i in range(L)]
        gh = {}
        gi = {(d, 0): gi0[d] for d in dirs}
        up = [(d, i) for d in dirs for i in range(1, L)]
        stat_fwd = bool(preact) and all(("stat",) + k in preact for k in keys)   # the forward launch wrote rows 1..7 (`stat_rows`)
        if stat_fwd:
            gi = {}
        elif preact and all(("gh",) + k in preact for k in keys):   # the forward kernel kept the pre-activations of this pass
            gh = {k: preact[("gh",) + k] for k in keys}
            gi.update({k: preact[("gi",) + k] for k in up})
        else:
            for k0 in range(0, len(keys), 4):
                grp = keys[k0:k0 + 4]
                res = gemm_nt_bias([out[k]["a"] for k in grp], [cells[k].w_hh_raw for k in grp], [cells[k].b_hh for k in grp])
                gh.update(dict(zip(grp, res)))
            for k0 in range(0, len(up), 4):
                grp = up[k0:k0 + 4]
                res = gemm_nt_bias([h[d][i - 1][:, :H] for d, i in grp], [cells[k].w_ih for k in grp],
                                   [cells[k].b_ih for k in grp])
                gi.update(dict(zip(grp, res)))
        gkeys = [("da", d, i) for d in dirs for i in range(L)] + [("q", d, i) for d in dirs for i in range(L)] + \
                [("dgi", d, i) for d in dirs for i in range(1, L)] + [("du", d, i) for d in dirs for i in range(L - 1)]
        widths = {k: (1 if k[0] == "q" else 3 * H) for k in gkeys if k[0] in ("q", "dgi")}
        gran, epoch, err = arena.get(gkeys, N, H, dev, widths=widths)
        args = BwdDataflowArgs()
        stat_bytes = lib.dagnn_bwd_dataflow_static_bytes_h(N, H)
        for k in keys:
            d, i = k
            c, bc, o = cells[k], args.cell[d][i], out[k]
            o["gi"], o["gh"] = gi.get(k), gh.get(k)
            if getattr(c, "w_hh_bt", None) is None:
                c.w_hh_bt = pack_dataflow_transposed(c.w_hh_raw, H)
                c.w_ih_bt = pack_dataflow_transposed(c.w_ih, H) if i > 0 else None
            stat = preact[("stat",) + k] if stat_fwd else torch.empty(stat_bytes // 4, **f32)
            keep.append(stat)
            bc.w_hh_t, bc.w_ih_t = c.w_hh_bt.data_ptr(), _ptr(c.w_ih_bt)
            bc.w_key, bc.alpha = o["_wkey"].data_ptr(), o["alpha"].data_ptr()
            bc.gi, bc.gh, bc.a, bc.b_hh = _ptr(gi.get(k)), _ptr(gh.get(k)), o["a"].data_ptr(), c.b_hh.data_ptr()
            bc.h, bc.g_ext, bc.stat = h[d][i].data_ptr(), g_ext[d][i].data_ptr(), stat.data_ptr()
            bc.da_granules, bc.q_granules = gran[("da", d, i)].data_ptr(), gran[("q", d, i)].data_ptr()
            bc.dgi_granules = gran[("dgi", d, i)].data_ptr() if i > 0 else None
            bc.du_granules = gran[("du", d, i)].data_ptr() if i + 1 < L else None
            bc.dgi, bc.dgh, bc.sigma = o["dgi"].data_ptr(), o["dgh"].data_ptr(), o["sigma"].data_ptr()
            bc.edge_feat_grad = _ptr(o["edge_feat_grad"])
        args.num_stacked, args.dir_mask, args.H = L, mask, H
        args.ld_h, args.ld_g, args.gld, args.groups = h[dirs[0]][0].shape[1], g_ext[dirs[0]][0].shape[1], H, int(groups)
        args.epoch, args.spin_limit = epoch, SPIN_LIMIT
        args.stat_rows_written = 1 if stat_fwd else 0
        sched = plan.dataflow_schedule(groups)
        recs = torch.empty(lib.dagnn_bwd_dataflow_record_bytes(N) // 4, dtype=torch.int32, device=dev)
        keep.append(recs)
        args.schedule, args.records, args.err = sched.data_ptr(), recs.data_ptr(), err.data_ptr()
        args.plan_status = plan.status.data_ptr()
        if DF_XCD:
            args.num_cus = effective_cus(dev, True)
            args.xcc_table = arena.xcc_table(dev).data_ptr()
            args.xcd_first = arena.xcd_first
        check(lib.dagnn_bwd_dataflow_prepare(C.byref(plan.desc), C.byref(args), _stream(plan.ws)), "dagnn_bwd_dataflow_prepare")
    with persistent_launch(plan.ws), _span("backward_run", plan.ws):
        check(lib.dagnn_bwd_dataflow_run(C.byref(plan.desc), C.byref(args), _stream(plan.ws)), "dagnn_bwd_dataflow_run")
    arena.watch(plan, folded=True)
    for o in out.values():
        o.pop("_wkey", None)
    out["_keep"] = keep   # buffers the launch reads: alive until the caller drops the result
    return out


_WGRAD_WS = {}   # per (kind, device, stream): the partial-tile workspaces of `wgrad` / `colsums` (grown on demand, reused by
                 # every step of that stream; backward passes on different streams of one device never share a buffer)
WGRAD_MAX_JOBS = 32   # jobs one `dagnn_wgrad_run` / `dagnn_colsum_run` call takes (WG_MAX_JOBS / CS_MAX_JOBS in csrc/wgrad.hip)


def _wgrad_ws(kind: str, ref: torch.Tensor, nbytes: int) -> torch.Tensor:
    key = (kind, ref.device, _stream(ref))
    ws = _WGRAD_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        # (a grown buffer replaces one that launches already queued on this stream may still read: stream-ordered
        # allocation keeps the old block alive until they have run)
        ws = _WGRAD_WS[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=ref.device)
    return ws


def wgrad(jobs, N: int, Hp: int, H: int):
    """Weight / bias gradients of GRU cells in one batch (`dagnn_wgrad_run`).  `jobs`: list of (dg [N, >= 3 Hp], inp
    [N, in_dim] (row pitch kept), want_bias); returns a list of (d_weight [3H, in_dim], d_bias [3H] or None).  More than
    WGRAD_MAX_JOBS jobs (bidirectional models of more than 8 stacked layers) go in several launches."""
    if len(jobs) > WGRAD_MAX_JOBS:
        out = []
        for k in range(0, len(jobs), WGRAD_MAX_JOBS):
            out += wgrad(jobs[k:k + WGRAD_MAX_JOBS], N, Hp, H)
        return out
    lib = _lib.load()
    dev = jobs[0][0].device
    arr = (_lib.WgradJob * len(jobs))()
    outs, keep = [], []
    kmax = 0
    for q, (dg, inp, want_bias) in enumerate(jobs):
        dg, inp = _rows(dg, "dg"), _rows(inp, "wgrad input")
        K0 = K2 = inp.shape[1]
        if K2 % 2:   # the kernel reads input columns in pairs (an odd width: the reference's default hs = 501, dvae/train.py:55)
            if inp.stride(0) > K2 and inp.stride(0) % 2 == 0:   # a view of wider rows: take the next word of each row along (its
                inp = inp.as_strided((inp.shape[0], K2 + 1), inp.stride())   # column of d_weight is dropped below)
            else:
                inp = torch.nn.functional.pad(inp, (0, 1))
            K2 += 1
        keep += [dg, inp]
        kmax = max(kmax, K2)
        dW = torch.empty(3 * H, K2, dtype=torch.float32, device=dev)
        db = torch.empty(3 * H, dtype=torch.float32, device=dev) if want_bias else None
        outs.append((dW if K2 == K0 else dW[:, :K0], db))
        arr[q] = _lib.WgradJob(dg.data_ptr(), inp.data_ptr(), dW.data_ptr(), _ptr(db), dg.stride(0), inp.stride(0), K2)
    cus = _num_cus(dev)
    splits = max(lib.dagnn_wgrad_splits(cus, len(jobs), Hp, kmax, max(N, 1)), 1)
    nbytes = lib.dagnn_wgrad_workspace_bytes(len(jobs), Hp, kmax, splits)
    ws = _wgrad_ws("wg", jobs[0][0], nbytes)
    check(lib.dagnn_wgrad_run(arr, len(jobs), N, Hp, H, splits, ws.data_ptr(), ws.numel() * 4, _stream(jobs[0][0])),
          "dagnn_wgrad_run")
    return [(dW if dW.is_contiguous() else dW.contiguous(), db) for dW, db in outs]


def colsums(jobs, N: int):
    """Weighted column sums in one batch (`dagnn_colsum_run`).  `jobs`: list of (x [N, cols] (row pitch kept; a 1-d
    tensor counts as [N, 1]), weight [N] or None); returns the list of [cols] sums.  More than WGRAD_MAX_JOBS jobs (a
    bidirectional model with edge features has three per cell: 12 cells = 36) go in several launches."""
    if len(jobs) > WGRAD_MAX_JOBS:
        out = []
        for k in range(0, len(jobs), WGRAD_MAX_JOBS):
            out += colsums(jobs[k:k + WGRAD_MAX_JOBS], N)
        return out
    lib = _lib.load()
    dev = jobs[0][0].device
    arr = (_lib.ColsumJob * len(jobs))()
    outs, keep = [], []
    kmax = 0
    for q, (x, w) in enumerate(jobs):
        x = x.view(-1, 1) if x.dim() == 1 else x
        if not (x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1):
            x = _dev(x, "colsum input", torch.float32)
        w = None if w is None else _dev(w, "colsum weight", torch.float32)
        keep += [x, w]
        o = torch.empty(x.shape[1], dtype=torch.float32, device=dev)
        outs.append(o)
        kmax = max(kmax, x.shape[1])
        arr[q] = _lib.ColsumJob(x.data_ptr(), _ptr(w), o.data_ptr(), x.stride(0) if x.shape[0] > 1 else x.shape[1], x.shape[1])
    nbytes = lib.dagnn_colsum_workspace_bytes(len(jobs), kmax)
    ws = _wgrad_ws("cs", jobs[0][0], nbytes)
    check(lib.dagnn_colsum_run(arr, len(jobs), N, ws.data_ptr(), ws.numel() * 4, _stream(jobs[0][0])), "dagnn_colsum_run")
    return outs


def gather_rows(h: torch.Tensor, num_graphs: int, stride: int, node_off: int, out: torch.Tensor,
                col_off: int) -> None:
    h = _dev(h, "h", torch.float32)
    check(_lib.load().dagnn_gather_rows(h.data_ptr(), h.shape[1], h.shape[1], num_graphs, stride, node_off,
                                        out.data_ptr(), out.shape[1], col_off, _stream(h)), "dagnn_gather_rows")


def iprop_step(values: Optional[torch.Tensor], pred_vid: Optional[torch.Tensor], w_key: torch.Tensor,
               vid_bias: Optional[torch.Tensor], H_given: Optional[torch.Tensor], x: torch.Tensor, cells) -> torch.Tensor:
    """`_ipropagate_to` for all graphs of a decoder step in one launch (`dagnn_iprop_step`).  values [B,P,hs] /
    pred_vid [B,P] int32 (or None with `H_given` [B,hs]); x [B,in0]; cells: the propagator's GRUCells.  Returns the
    new states [L,B,hs]."""
    x = _dev(x, "x", torch.float32)
    B, in0 = x.shape
    hs = cells[0].weight_hh.shape[1]
    L = len(cells)
    P = 0
    if H_given is not None:
        H_given = _dev(H_given, "H", torch.float32)
    elif values is not None:
        values = _dev(values, "predecessor states", torch.float32)
        pred_vid = _dev(pred_vid, "predecessor ids", torch.int32)
        P = values.shape[1]
    layers = (_lib.IpropLayer * L)()
    keep = []
    for l, c in enumerate(cells):
        ts = [_dev(t.detach(), "GRU parameter", torch.float32) for t in (c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh)]
        keep.append(ts)
        layers[l] = _lib.IpropLayer(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), ts[0].shape[1])
    states = torch.empty(L, B, hs, dtype=torch.float32, device=x.device)
    if H_given is None and P == 0:   # no graph has a predecessor: the aggregate is zero (`_get_zero_hidden`)
        H_given = torch.zeros(B, hs, dtype=torch.float32, device=x.device)
    wk = _dev(w_key.detach(), "w_key", torch.float32)
    vb = None if vid_bias is None else _dev(vid_bias.detach(), "vid_bias", torch.float32)
    check(_lib.load().dagnn_iprop_step(_ptr(values), _ptr(pred_vid), B, P, hs, wk.data_ptr(), _ptr(vb), _ptr(H_given),
                                       x.data_ptr(), in0, layers, L, states.data_ptr(), _stream(x)), "dagnn_iprop_step")
    return states


def gather_rows_batch(jobs, num_graphs: int, stride: int, out: torch.Tensor) -> None:
    """`gather_rows` for several (h, node_off, col_off) in one launch."""
    arr = (_lib.GatherJob * len(jobs))()
    keep = []
    for k, (h, node_off, col_off) in enumerate(jobs):
        h = _rows(h, "h")
        keep.append(h)
        arr[k] = _lib.GatherJob(h.data_ptr(), h.stride(0), h.shape[1], int(node_off), int(col_off))
    check(_lib.load().dagnn_gather_rows_batch(arr, len(jobs), num_graphs, stride, out.data_ptr(), out.shape[1], _stream(out)),
          "dagnn_gather_rows_batch")


def topo_layers(edge_index: torch.Tensor, batch: torch.Tensor, num_graphs: int):
    """(layer_fwd, layer_bwd, status): longest-path layer ids of both orientations for a collated batch, on the
    device (`src/utils_dag.py:8-52` without the per-graph numpy pass).  `status` is a device int32[1]: bit 16 =
    some graph has a cycle (check it with `int(status)` when the input is not trusted - that synchronises)."""
    edge_index = _dev(edge_index, "edge_index", torch.int64)
    batch = _dev(batch, "batch", torch.int64)
    N, E = batch.numel(), edge_index.shape[1]
    lf = torch.empty(N, dtype=torch.int64, device=batch.device)
    lb = torch.empty(N, dtype=torch.int64, device=batch.device)
    status = torch.zeros(1, dtype=torch.int32, device=batch.device)
    check(_lib.load().dagnn_topo_layers(edge_index.data_ptr(), batch.data_ptr(), N, E, int(num_graphs), lf.data_ptr(),
                                        lb.data_ptr(), status.data_ptr(), _stream(batch)), "dagnn_topo_layers")
    return lf, lb, status
