"""ctypes binding of libdagnn_hip.so (the C ABI declared in include/dagnn_hip.h).

There is NO fallback: if the library cannot be loaded the product path raises.  Tensors are passed
as raw device pointers; torch only owns the memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

from . import build as _build

MAX_DIRS = 2
MAX_GROUPS = 4
MAX_READOUT_JOBS = 24      # DAGNN_MAX_READOUT_JOBS
ATTN_GRAD_MAX_JOBS = 16    # DAGNN_ATTN_GRAD_MAX_JOBS

DAGNN_OK = 0
_ERRORS = {-22: "DAGNN_EINVAL (bad argument)", -28: "DAGNN_ENOSPC (workspace too small)"}


class DagnnHipError(RuntimeError):
    pass


class ReadoutJob(C.Structure):
    _fields_ = [("h", C.c_void_p), ("ld_h", C.c_int), ("width", C.c_int), ("dir", C.c_int), ("col_off", C.c_int)]


class OptTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("numel", C.c_int64)]


class DfPackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("aux", C.c_void_p), ("transposed", C.c_int32), ("rows", C.c_int32),
                ("cols", C.c_int32)]


class ReadoutBwdJob(C.Structure):
    _fields_ = [("h", C.c_void_p), ("grad_h", C.c_void_p), ("ld_h", C.c_int32), ("ld_g", C.c_int32), ("width", C.c_int32),
                ("dir", C.c_int32), ("col_off", C.c_int32)]


class AttnGradJob(C.Structure):
    _fields_ = [("key_sum", C.c_void_p), ("feat_sum", C.c_void_p), ("sigma_sum", C.c_void_p), ("edge_w", C.c_void_p),
                ("edge_b", C.c_void_p), ("attn_w", C.c_void_p), ("g_attn", C.c_void_p), ("g_edge_w", C.c_void_p),
                ("g_edge_b", C.c_void_p), ("dq", C.c_int32), ("kd", C.c_int32), ("attn_len", C.c_int32), ("R", C.c_int32)]


class Plan(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bytes", C.c_size_t), ("N", C.c_int64), ("E", C.c_int64),
                ("B", C.c_int64), ("num_edge_feats", C.c_int), ("flags", C.c_int)]


class GemmGroup(C.Structure):
    _fields_ = [("A", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("C", C.c_void_p)]


class LayerArgs(C.Structure):
    _fields_ = [("gi", C.c_void_p * MAX_DIRS), ("w_hh_t", C.c_void_p * MAX_DIRS), ("b_hh", C.c_void_p * MAX_DIRS),
                ("w_key", C.c_void_p * MAX_DIRS), ("edge_gain", C.c_void_p * MAX_DIRS),
                ("vid_bias", C.c_void_p * MAX_DIRS), ("h", C.c_void_p * MAX_DIRS), ("score", C.c_void_p * MAX_DIRS),
                ("vid_mod", C.c_int), ("ld_h", C.c_int), ("static_score", C.c_int), ("debug_timing", C.c_void_p)]


MAX_STACKED = 8


class PackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out_slices16", C.c_void_p), ("out_slices32", C.c_void_p), ("out_mfma", C.c_void_p),
                ("H", C.c_int32), ("K", C.c_int32)]


MAX_PACK_JOBS = 16


class FrontierCell(C.Structure):
    _fields_ = [("w_hh_pk16", C.c_void_p), ("w_hh_pk32", C.c_void_p), ("w_ih_pk16", C.c_void_p),
                ("w_ih_pk32", C.c_void_p), ("w_hh_mfma", C.c_void_p), ("w_ih_mfma", C.c_void_p), ("b_hh", C.c_void_p), ("b_ih", C.c_void_p), ("w_key", C.c_void_p),
                ("static_score", C.c_void_p), ("edge_gain", C.c_void_p), ("vid_bias", C.c_void_p), ("gi0", C.c_void_p), ("h_out", C.c_void_p),
                ("granules", C.c_void_p)]


class FrontierArgs(C.Structure):
    _fields_ = [("cell", (FrontierCell * MAX_STACKED) * MAX_DIRS), ("num_stacked", C.c_int), ("dir_mask", C.c_int),
                ("H", C.c_int), ("ld_h", C.c_int), ("vid_mod", C.c_int), ("num_cus", C.c_int), ("rb4_max_wgs", C.c_int), ("mfma_min_rows", C.c_int),
                ("agg_scratch", C.c_void_p), ("agg_scratch_rows", C.c_int),
                ("tail_replicas", C.c_int), ("tail_slice_units", C.c_int), ("tail_max_blocks", C.c_int), ("epoch", C.c_uint),
                ("tail_err", C.c_void_p), ("debug_timing", C.c_void_p), ("side_stream", C.c_void_p),
                ("layer_split", C.POINTER(C.c_int32) * MAX_DIRS), ("fork_event", C.c_void_p), ("join_event", C.c_void_p)]


class DataflowCell(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w_hh", "w_ih", "b_hh", "b_ih", "w_key", "static_score", "edge_gain",
                                          "vid_bias", "gi0", "h_out", "granules", "proj_granules", "gh_out", "gi_out",
                                          "agg_edge_w", "agg_edge_b")] + [("agg", C.c_int)]


class DataflowArgs(C.Structure):
    _fields_ = [("cell", (DataflowCell * MAX_STACKED) * MAX_DIRS), ("num_stacked", C.c_int), ("dir_mask", C.c_int),
                ("H", C.c_int), ("ld_h", C.c_int), ("gld", C.c_int), ("pld", C.c_int), ("vid_mod", C.c_int), ("groups", C.c_int),
                ("epoch", C.c_uint), ("schedule", C.c_void_p), ("err", C.c_void_p), ("debug_timing", C.c_void_p),
                ("spin_limit", C.c_uint), ("debug_wg", C.c_int), ("num_cus", C.c_int), ("xcc_table", C.c_void_p),
                ("plan_status", C.c_void_p), ("xcd_first", C.c_int), ("stat_rows", C.c_int), ("slices64", C.c_int)]


class TilesCell(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w_hh", "w_ih", "b_hh", "b_ih", "w_key", "edge_gain", "gi0", "h_out", "vid_bias")]


class TilesArgs(C.Structure):
    _fields_ = [("cell", (TilesCell * MAX_STACKED) * MAX_DIRS), ("num_stacked", C.c_int), ("dir_mask", C.c_int),
                ("H", C.c_int), ("ld_h", C.c_int), ("num_cus", C.c_int), ("epoch", C.c_uint), ("counters", C.c_void_p),
                ("err", C.c_void_p), ("spin_limit", C.c_uint), ("plan_status", C.c_void_p), ("first_layer", C.c_int * MAX_DIRS),
                ("debug_timing", C.c_void_p), ("vid_mod", C.c_int)]


class BackwardCell(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w_hh", "w_ih", "w_key", "edge_gain", "vid_bias", "static_score", "h", "a", "alpha", "gi", "gh", "g_ext",
                                          "da", "dgi", "dgh", "sigma", "edge_feat_grad", "da_granules", "du_granules",
                                          "g_ext_static")]


class BackwardArgs(C.Structure):
    _fields_ = [("cell", (BackwardCell * MAX_STACKED) * MAX_DIRS), ("num_stacked", C.c_int), ("dir_mask", C.c_int),
                ("H", C.c_int), ("ld_h", C.c_int), ("vid_mod", C.c_int), ("num_cus", C.c_int), ("thin_wgs", C.c_int),
                ("tail_replicas", C.c_int), ("tail_max_blocks", C.c_int), ("epoch", C.c_uint), ("tail_err", C.c_void_p),
                ("side_stream", C.c_void_p), ("layer_split", C.POINTER(C.c_int32) * MAX_DIRS),
                ("fork_event", C.c_void_p), ("join_event", C.c_void_p)]


class WgradJob(C.Structure):
    _fields_ = [("dg", C.c_void_p), ("inp", C.c_void_p), ("d_weight", C.c_void_p), ("d_bias", C.c_void_p),
                ("ld_dg", C.c_int), ("ld_in", C.c_int), ("in_dim", C.c_int)]


class ColsumJob(C.Structure):
    _fields_ = [("x", C.c_void_p), ("weight", C.c_void_p), ("out", C.c_void_p), ("ld_x", C.c_int), ("cols", C.c_int)]


class BwdDataflowCell(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w_hh_t", "w_ih_t", "w_key", "alpha", "gi", "gh", "a", "b_hh", "h", "g_ext", "stat",
                                          "da_granules", "q_granules", "dgi_granules", "du_granules", "dgi", "dgh", "sigma",
                                          "edge_feat_grad")]


class BwdDataflowArgs(C.Structure):
    _fields_ = [("cell", (BwdDataflowCell * MAX_STACKED) * MAX_DIRS), ("num_stacked", C.c_int), ("dir_mask", C.c_int),
                ("H", C.c_int), ("ld_h", C.c_int), ("ld_g", C.c_int), ("gld", C.c_int), ("groups", C.c_int),
                ("epoch", C.c_uint), ("spin_limit", C.c_uint), ("schedule", C.c_void_p), ("records", C.c_void_p),
                ("err", C.c_void_p), ("plan_status", C.c_void_p), ("num_cus", C.c_int), ("xcc_table", C.c_void_p), ("xcd_first", C.c_int),
                ("stat_rows_written", C.c_int)]


class GatherJob0(C.Structure):   # (dagnn_gather_job inside dagnn_encode_args; `GatherJob` below has the same layout)
    _fields_ = [("h", C.c_void_p), ("ld_h", C.c_int), ("width", C.c_int), ("node_off", C.c_int), ("col_off", C.c_int)]


class EncodeArgs(C.Structure):
    _fields_ = [("plan", Plan), ("edge_index", C.c_void_p), ("layer_fwd", C.c_void_p), ("layer_bwd", C.c_void_p),
                ("batch", C.c_void_p), ("edge_attr", C.c_void_p), ("plan_status", C.c_void_p),
                ("gemm", GemmGroup * 2), ("num_gemm", C.c_int), ("gemm_cols", C.c_int), ("in_dim", C.c_int), ("ld_x", C.c_int),
                ("schedule", C.c_void_p), ("schedule_bytes", C.c_size_t), ("cost_layer", C.c_int), ("cost_row", C.c_int),
                ("df", DataflowArgs), ("jobs", GatherJob0 * 16), ("num_jobs", C.c_int), ("stride", C.c_int),
                ("hcat", C.c_void_p), ("ld_hcat", C.c_int), ("w_out", C.c_void_p), ("b_out", C.c_void_p), ("out", C.c_void_p),
                ("out_dim", C.c_int)]


AGG_ATTN, AGG_MATTN, AGG_GATED, AGG_ADD, AGG_MAX, AGG_GIVEN = range(6)
POOL_MAX, POOL_ADD, POOL_MEAN = range(3)


class VariantAggregator(C.Structure):
    _fields_ = [("mode", C.c_int32), ("lands", C.c_int32), ("val_dim", C.c_int32), ("aux_dim", C.c_int32),
                ("out_dim", C.c_int32), ("reserved", C.c_int32), ("vals", C.c_void_p), ("ld_vals", C.c_int64),
                ("node0", C.c_void_p), ("node1", C.c_void_p), ("ld_node", C.c_int64), ("edge_mat0", C.c_void_p),
                ("edge_vec0", C.c_void_p), ("edge_mat1", C.c_void_p), ("edge_vec1", C.c_void_p), ("out", C.c_void_p),
                ("ld_out", C.c_int64)]


class VariantMap(C.Structure):
    _fields_ = [("w_t", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("ld_out", C.c_int64),
                ("out_dim", C.c_int32), ("vid_mod", C.c_int32), ("vid_bias", C.c_void_p)]


class VariantCell(C.Structure):
    _fields_ = [("agg", VariantAggregator), ("recurrent", C.c_int32), ("in_dim", C.c_int32), ("input", C.c_void_p),
                ("ld_input", C.c_int64), ("w_in_t", C.c_void_p), ("w_agg_t", C.c_void_p), ("b_in", C.c_void_p),
                ("b_agg", C.c_void_p), ("h", C.c_void_p), ("ld_h", C.c_int64), ("map", VariantMap * 3),
                ("num_maps", C.c_int32), ("reserved", C.c_int32)]


class GatherJob(C.Structure):
    _fields_ = [("h", C.c_void_p), ("ld_h", C.c_int), ("width", C.c_int), ("node_off", C.c_int), ("col_off", C.c_int)]


class IpropLayer(C.Structure):
    _fields_ = [("w_ih", C.c_void_p), ("w_hh", C.c_void_p), ("b_ih", C.c_void_p), ("b_hh", C.c_void_p), ("in_dim", C.c_int)]


class VariantBwdCell(C.Structure):
    _fields_ = [("mode", C.c_int32), ("lands", C.c_int32), ("in_dim", C.c_int32), ("proj_dim", C.c_int32),
                ("recurrent", C.c_int32), ("reserved", C.c_int32)] + \
        [(k, C.c_void_p) for k in ("h", "a", "gi", "gh", "node0", "node1", "alpha", "edge_mat0", "edge_vec0", "edge_mat1",
                                   "edge_vec1", "w_node", "w_query", "w_hh", "w_ih", "g", "g_in", "da", "dgi", "dgh",
                                   "dnode0", "dnode1", "dlogit", "esum")]


class VariantBwdArgs(C.Structure):
    _fields_ = [("cell", (VariantBwdCell * MAX_STACKED) * MAX_DIRS), ("num_stacked", C.c_int), ("dir_mask", C.c_int),
                ("H", C.c_int)]


class VariantArgs(C.Structure):
    _fields_ = [("cell", (VariantCell * MAX_STACKED) * MAX_DIRS), ("num_stacked", C.c_int), ("dir_mask", C.c_int),
                ("H", C.c_int)]


# every symbol include/dagnn_hip.h declares: (restype, argtypes)
class PrepTable(C.Structure):
    _fields_ = [("type_emb", C.c_void_p), ("attr_emb", C.c_void_p), ("depth_emb", C.c_void_p), ("out", C.c_void_p),
                ("width", C.c_int), ("ld_out", C.c_int)]


PREPARE_MAX_TABLES = 3


class PrepareRows(C.Structure):
    _fields_ = [("x", C.c_void_p), ("depth", C.c_void_p), ("max_depth", C.c_int), ("num_tables", C.c_int),
                ("table", PrepTable * PREPARE_MAX_TABLES), ("stack_src", C.c_void_p * 4), ("stack_out", C.c_void_p)]


SYMBOLS = {
    "dagnn_version": (C.c_char_p, []),
    "dagnn_plan_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "dagnn_plan_is_small": (C.c_int, [C.c_int64, C.c_int64, C.c_int64]),
    "dagnn_plan_layout": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    "dagnn_plan_build": (C.c_int, [C.POINTER(Plan), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "dagnn_prepare": (C.c_int, [C.POINTER(Plan), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(PrepareRows), C.c_void_p]),
    "dagnn_encode_ast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "dagnn_gemm_nt_bias": (C.c_int, [C.POINTER(GemmGroup), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p]),
    "dagnn_pack_whh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dagnn_recurrence_layer": (C.c_int, [C.POINTER(Plan), C.POINTER(LayerArgs), C.c_int, C.c_int, C.c_void_p]),
    "dagnn_pack_slices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "dagnn_pack_batch": (C.c_int, [C.POINTER(PackJob), C.c_int, C.c_void_p]),
    "dagnn_pack_mfma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dagnn_frontier_run": (C.c_int, [C.POINTER(Plan), C.POINTER(FrontierArgs), C.POINTER(C.POINTER(C.c_int32)),
                                     C.POINTER(C.c_int32), C.c_void_p]),
    "dagnn_dataflow_groups": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "dagnn_dataflow_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int]),
    "dagnn_dataflow_layout": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    "dagnn_dataflow_schedule": (C.c_int, [C.POINTER(Plan), C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p]),
    "dagnn_dataflow_run": (C.c_int, [C.POINTER(Plan), C.POINTER(DataflowArgs), C.c_void_p]),
    "dagnn_dataflow_run_wide": (C.c_int, [C.POINTER(Plan), C.POINTER(DataflowArgs), C.c_void_p]),
    "dagnn_dataflow_run_x": (C.c_int, [C.POINTER(Plan), C.POINTER(DataflowArgs), C.c_void_p]),
    "dagnn_tiles_launches": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dagnn_tiles_run": (C.c_int, [C.POINTER(Plan), C.POINTER(TilesArgs), C.c_void_p]),
    "dagnn_pack_dataflow": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dagnn_pack_dataflow_transposed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "dagnn_score_parts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "dagnn_readout_max": (C.c_int, [C.POINTER(Plan), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p]),
    "dagnn_readout_max_batch": (C.c_int, [C.POINTER(Plan), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dagnn_readout_pool": (C.c_int, [C.POINTER(Plan), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_int, C.c_int, C.c_void_p]),
    "dagnn_backward_prepare": (C.c_int, [C.POINTER(Plan), C.POINTER(BackwardArgs), C.c_void_p]),
    "dagnn_backward_run": (C.c_int, [C.POINTER(Plan), C.POINTER(BackwardArgs), C.POINTER(C.POINTER(C.c_int32)),
                                     C.POINTER(C.c_int32), C.c_void_p]),
    "dagnn_bwd_dataflow_record_bytes": (C.c_size_t, [C.c_int64]),
    "dagnn_bwd_dataflow_static_bytes": (C.c_size_t, [C.c_int64]),
    "dagnn_bwd_dataflow_static_bytes_h": (C.c_size_t, [C.c_int64, C.c_int]),
    "dagnn_bwd_dataflow_prepare": (C.c_int, [C.POINTER(Plan), C.POINTER(BwdDataflowArgs), C.c_void_p]),
    "dagnn_bwd_dataflow_run": (C.c_int, [C.POINTER(Plan), C.POINTER(BwdDataflowArgs), C.c_void_p]),
    "dagnn_bwd_dataflow_run_wide": (C.c_int, [C.POINTER(Plan), C.POINTER(BwdDataflowArgs), C.c_void_p]),
    "dagnn_colsum_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "dagnn_colsum_run": (C.c_int, [C.POINTER(ColsumJob), C.c_int, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "dagnn_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "dagnn_wgrad_splits": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64]),
    "dagnn_wgrad_run": (C.c_int, [C.POINTER(WgradJob), C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_size_t, C.c_void_p]),
    "dagnn_readout_max_backward": (C.c_int, [C.POINTER(Plan), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dagnn_pack_dataflow_batch": (C.c_int, [C.POINTER(DfPackJob), C.c_int, C.c_int, C.c_void_p]),
    "dagnn_attn_grads_run": (C.c_int, [C.POINTER(AttnGradJob), C.c_int, C.c_void_p]),
    "dagnn_readout_max_backward_batch": (C.c_int, [C.POINTER(Plan), C.POINTER(ReadoutBwdJob), C.c_int, C.c_void_p, C.c_int,
                                                   C.c_void_p]),
    "dagnn_topo_layers": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "dagnn_variant_aggregate": (C.c_int, [C.POINTER(Plan), C.POINTER(VariantAggregator), C.c_int, C.c_int32, C.c_int32,
                                          C.c_void_p]),
    "dagnn_variant_run": (C.c_int, [C.POINTER(Plan), C.POINTER(VariantArgs), C.POINTER(C.POINTER(C.c_int32)),
                                    C.POINTER(C.c_int32), C.c_void_p]),
    "dagnn_variant_mattn_prepare": (C.c_int, [C.POINTER(Plan), C.POINTER(VariantBwdCell), C.c_int, C.c_int, C.c_int32,
                                              C.c_int32, C.c_void_p]),
    "dagnn_variant_aggregator_backward": (C.c_int, [C.POINTER(Plan), C.POINTER(VariantBwdCell), C.c_int, C.c_int, C.c_int32,
                                                    C.c_int32, C.c_void_p]),
    "dagnn_variant_backward_run": (C.c_int, [C.POINTER(Plan), C.POINTER(VariantBwdArgs), C.POINTER(C.POINTER(C.c_int32)),
                                             C.POINTER(C.c_int32), C.c_void_p]),
    "dagnn_iprop_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.POINTER(IpropLayer), C.c_int, C.c_void_p, C.c_void_p]),
    "dagnn_encode_forward": (C.c_int, [C.POINTER(EncodeArgs), C.c_void_p]),
    "dagnn_debug_occupy": (C.c_int, [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "dagnn_tn_product": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "dagnn_seq_ce": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "dagnn_opt_chunks": (C.c_int64, [C.c_void_p, C.c_int]),
    "dagnn_grad_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "dagnn_clip_adam": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, C.c_float,
                                  C.c_void_p, C.c_void_p]),
    "dagnn_score_parts_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "dagnn_param_fingerprint": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dagnn_gather_rows_batch": (C.c_int, [C.POINTER(GatherJob), C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dagnn_gather_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p]),
}

_lock = threading.Lock()
_lib = None


def lib_path() -> str:
    return _build.LIB


def load():
    """Load (building first if the in-tree .so is missing or stale and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = _build.LIB
        if _build.is_stale():
            try:
                _build.build()
            except Exception as exc:  # no hipcc and no prebuilt library: fail loudly
                if not os.path.exists(path):
                    raise DagnnHipError(
                        "libdagnn_hip.so is missing and could not be built (%s). There is no CPU or "
                        "PyTorch fallback for the DAGNN hot path." % exc) from exc
                # an OLDER library next to NEWER sources: its argument structs and plan layout may no longer match
                # the ctypes mirrors in this file - that is memory corruption, not an error message
                if os.environ.get("DAGNN_AMD_ALLOW_STALE", "0") != "1":
                    raise DagnnHipError(
                        "%s is older than its sources and the rebuild failed (%s); refusing to load a library whose "
                        "ABI may differ from this package (set DAGNN_AMD_ALLOW_STALE=1 to load it anyway)" % (path, exc)) from exc
        try:
            lib = C.CDLL(path)
        except OSError as exc:
            raise DagnnHipError("cannot load %s: %s (no fallback path exists)" % (path, exc)) from exc
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise DagnnHipError("%s does not export %s" % (path, name)) from exc
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


def check(code: int, what: str) -> None:
    if code == DAGNN_OK:
        return
    if code <= -1000:
        msg = "hipError_t %d" % (-code - 1000)
    else:
        msg = _ERRORS.get(code, "error %d" % code)
    raise DagnnHipError("%s failed: %s" % (what, msg))
