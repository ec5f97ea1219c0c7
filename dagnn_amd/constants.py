"""Aggregator / read-out name constants (the ctor vocabulary of `src/constants.py:12-27`)."""

NA_SUM = "add"
NA_MAX = "max"
NA_GATED_SUM = "gated_sum"
NA_SELF_ATTN_X = "self_attn_x"  # attention weights from the predecessors' inputs x
NA_SELF_ATTN_H = "self_attn_h"
NA_ATTN_X = "attn_x"  # query = own x, keys = predecessors' x
NA_ATTN_H = "attn_h"  # query = own x, keys = predecessors' hidden states (every BASELINE config)
NA_MATTN_H = "mattn_h"  # multiplicative attention

P_MEAN = "mean"
P_ADD = "add"
P_SUM = "sum"
P_MAX = "max"
P_ATTN = "attn"
EMB_POOLINGS = [P_MEAN, P_MAX, P_SUM]
POOLINGS = [P_MEAN, P_MAX, P_ATTN, P_ADD]
