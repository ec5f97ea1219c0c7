cd /tmp && export TMPDIR=/tmp
export DAGNN_AMD_PLAN_OVERLAP=$1 DAGNN_AMD_FOLD_INPUT=$2
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/tr_$1$2 -o tr --output-format csv -- python $GRAFT_REPO_ROOT/scripts/plan_overlap_ab.py child > $GRAFT_REPO_ROOT/gpurun_out/tr_$1$2.log 2>&1
grep median $GRAFT_REPO_ROOT/gpurun_out/tr_$1$2.log
