#!/bin/bash
# instruction / wave-cycle counters of the dominant kernel: separate rocprofv3 --pmc passes (<= 4 counters each, no trace
# domains next to them), averaged per launch of dataflow_kernel<16> -> profiles/<name>.json
name=${1:-r04_pmc_instructions}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 2 --warmup 1 --cpu-passes 0 --no-kernel-timer --train-steps 0 --other-configs 0"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rm -rf gpurun_out/${name}_p$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/${name}_p$i -o c -- $CMD > gpurun_out/${name}_p$i.log 2>&1
done
python - "$name" <<'PY'
import csv, glob, json, sys, os
name = sys.argv[1]
tot, cnt = {}, {}
for f in glob.glob("gpurun_out/%s_p*/**/*counter_collection.csv" % name, recursive=True):
    for r in csv.DictReader(open(f)):
        if "dataflow_kernel<16>" not in r["Kernel_Name"] or "bwd_" in r["Kernel_Name"]:
            continue
        k = r["Counter_Name"]
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
        cnt[k] = cnt.get(k, 0) + 1
out = {"note": "rocprofv3 --pmc passes (4 counters each, separate runs) over `python bench.py --steps 2 --warmup 1 --cpu-passes 0 "
               "--no-kernel-timer --train-steps 0 --other-configs 0`; dataflow_kernel<16> only, average per launch. SQ cycle "
               "counters are in units of 4 clocks summed over waves.",
       "launches": {k: cnt[k] for k in sorted(cnt)},
       "per_launch": {k: int(tot[k] / cnt[k]) for k in sorted(tot)}}
json.dump(out, open("profiles/%s.json" % name, "w"), indent=1)
json.dump(out, open("gpurun_out/%s.json" % name, "w"), indent=1)
print(json.dumps(out["per_launch"], indent=1))
PY
