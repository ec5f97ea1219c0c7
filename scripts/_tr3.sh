cd /tmp && export TMPDIR=/tmp
for v in base; do
  rocprofv3 --kernel-trace --stats -d /tmp/pp_$v -o tr --output-format csv -- python $GRAFT_REPO_ROOT/scripts/plan_probe.py > /tmp/pp_$v.log 2>&1
  echo $v $(grep plan_graph /tmp/pp_$v/tr_kernel_stats.csv | awk -F, '{print $(NF-4)}')
done
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests -m gpu -x -q -k "plan or prepare or schedule_host or small_builds or contract or oracle_csr" 2>&1 | tail -2
