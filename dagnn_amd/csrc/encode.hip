// encode.hip - one host entry point for a whole encoder pass over a collated batch (round 4).
//
// Reference path: `forward(G)` of dvae/dagnn.py:99-161 / dvae/dagnn_bn.py:98-152 up to the read-out (the evaluation pass of a
// D-VAE encoder: 64 x 8 or 128 x 10 nodes).  Such a batch is HOST-bound: its ~8 launches take the device ~95 us, issuing
// them one ctypes call at a time - argument structs, checks and bookkeeping in Python around every call - took the host
// 115-135 us (scripts/small_host_segments.py).  This entry point issues the SAME launches (plan, input-side GEMM, dataflow
// schedule, the persistent recurrence, the end-vertex gather, the final Linear) from ONE call: nothing new runs on the
// device, the host's share drops to the launches themselves.
#include "common.h"

extern "C" int dagnn_encode_forward(const dagnn_encode_args* a, void* stream) {
    if (!a || !a->plan.data || !a->edge_index || !a->layer_fwd || !a->layer_bwd || !a->batch || !a->plan_status || !a->schedule ||
        a->num_gemm < 1 || a->num_gemm > 2 || a->num_jobs < 0 || a->num_jobs > 16)
        return DAGNN_EINVAL;
    int rc = dagnn_plan_build(&a->plan, a->edge_index, a->layer_fwd, a->layer_bwd, a->batch, a->edge_attr, a->plan_status, stream);
    if (rc != DAGNN_OK) return rc;
    // stacked layer 0, every direction: gi0 = x W_ih^T + b_ih (independent of the plan)
    rc = dagnn_gemm_nt_bias(a->gemm, a->num_gemm, a->plan.N, a->gemm_cols, a->in_dim, a->ld_x, a->in_dim, a->gemm_cols, stream);
    if (rc != DAGNN_OK) return rc;
    rc = dagnn_dataflow_schedule(&a->plan, a->schedule, a->schedule_bytes, a->df.groups, a->cost_layer, a->cost_row, a->plan_status, stream);
    if (rc != DAGNN_OK) return rc;
    rc = dagnn_dataflow_run(&a->plan, &a->df, stream);
    if (rc != DAGNN_OK) return rc;
    if (a->num_jobs > 0) {
        rc = dagnn_gather_rows_batch(a->jobs, a->num_jobs, a->plan.B, a->stride, a->hcat, a->ld_hcat, stream);
        if (rc != DAGNN_OK) return rc;
    }
    if (a->w_out) {   // the model's final Linear on the gathered rows (hg_unify / out_linear)
        dagnn_gemm_group g = {a->hcat, a->w_out, a->b_out, a->out};
        rc = dagnn_gemm_nt_bias(&g, 1, a->plan.B, a->out_dim, a->ld_hcat, a->ld_hcat, a->ld_hcat, a->out_dim, stream);
        if (rc != DAGNN_OK) return rc;
    }
    return DAGNN_OK;
}
