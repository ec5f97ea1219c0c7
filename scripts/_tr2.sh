cd /tmp && export TMPDIR=/tmp
export DAGNN_AMD_PREPARE=0 DAGNN_AMD_PLAN_OVERLAP=0
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/tr_sep -o tr --output-format csv -- python $GRAFT_REPO_ROOT/scripts/plan_overlap_ab.py child > $GRAFT_REPO_ROOT/gpurun_out/tr_sep.log 2>&1
grep median $GRAFT_REPO_ROOT/gpurun_out/tr_sep.log
grep -E "plan_|df_" $GRAFT_REPO_ROOT/gpurun_out/tr_sep/tr_kernel_stats.csv | cut -c1-60,100-200
