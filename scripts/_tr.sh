cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tr_fused -o tr --output-format csv -- python $GRAFT_REPO_ROOT/scripts/plan_overlap_ab.py child > $GRAFT_REPO_ROOT/gpurun_out/tr_fused.log 2>&1
grep median $GRAFT_REPO_ROOT/gpurun_out/tr_fused.log
