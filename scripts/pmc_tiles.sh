#!/bin/bash
# PMC counters of the tile kernel on the cfg-5 forward (DAGNN_AMD_TILES=2: the kernel alone): separate rocprofv3 --pmc passes
# (no trace domains next to them), folded per tiles_kernel launch into profiles/r03_pmc_tiles.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export DAGNN_AMD_TILES=2
CMD="python scripts/tiles_time.py"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rm -rf gpurun_out/pmc_tiles_$i
  timeout 240 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_tiles_$i -o p -- $CMD > gpurun_out/pmc_tiles_$i.log 2>&1
  grep "ms per" gpurun_out/pmc_tiles_$i.log
done
python - <<'PY'
import csv, glob, json, collections
out = {"what": "rocprofv3 --pmc passes of python scripts/tiles_time.py with DAGNN_AMD_TILES=2 (cfg 5 forward on the tile kernel alone), "
               "per tiles_kernel launch: [stacked-layer-0 launch, launch of the four layers above]; FETCH_SIZE / WRITE_SIZE in KiB as "
               "reported (FETCH_SIZE doubled in hbm_bytes, as the microarch guide prescribes)", "counters": {}}
for f in sorted(glob.glob("gpurun_out/pmc_tiles_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "tiles_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, by in acc.items():
        ids = sorted(by)
        # launches alternate: chunk 0, chunk 1
        c0 = [by[i] for k, i in enumerate(ids) if k % 2 == 0]
        c1 = [by[i] for k, i in enumerate(ids) if k % 2 == 1]
        out["counters"][name] = [sum(c0) / max(len(c0), 1), sum(c1) / max(len(c1), 1)]
c = out["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm_bytes"] = [round((2 * c["FETCH_SIZE"][k] + c["WRITE_SIZE"][k]) * 1024) for k in range(2)]
json.dump(out, open("gpurun_out/r03_pmc_tiles.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
