#!/bin/bash
# Matrix-core utilisation of every kernel that issues MFMAs today: ONE counter set (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES,
# SQ_INSTS_VALU_MFMA_MOPS_F32, GRBM_GUI_ACTIVE), collected in separate `rocprofv3 --pmc` runs (no trace domains next to them)
# of four workloads: headline forward + training steps, the reference's training shape (emb_dim 300), cfg 5, cfg 5 on the
# tile kernel alone.  Folded per kernel into gpurun_out/<name>.json (copy to profiles/).   usage: scripts/pmc_mfma.sh r06_pmc_mfma
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
NAME=${1:-r05_pmc_mfma}
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"
run() {   # tag, command...
  tag=$1; shift
  rm -rf gpurun_out/${NAME}_$tag
  timeout 600 rocprofv3 --pmc $C --output-format csv -d gpurun_out/${NAME}_$tag -o p -- "$@" > gpurun_out/${NAME}_$tag.log 2>&1
  echo "$tag rc=$?"
}
run headline python bench.py --steps 3 --warmup 1 --cpu-passes 0 --no-kernel-timer --train-steps 3 --other-configs 0
run h300 env HS=300 python scripts/h300_probe.py
run cfg5 python bench.py --batch 256 --hidden 512 --layers 5 --steps 2 --warmup 1 --cpu-passes 0 --no-kernel-timer --train-steps 0 --other-configs 0
run cfg5_tiles env DAGNN_AMD_TILES=2 python scripts/tiles_time.py
python - "$NAME" <<'PY'
import csv, glob, json, re, sys, collections
name = sys.argv[1]
SIMDS, XCDS = 1024, 8
out = {"what": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE, one run per "
               "workload (scripts/pmc_mfma.sh), per kernel and launch.  mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 "
               "SIMDs): the share of the chip's matrix-pipe cycles in use while the kernel runs; mfma_util_busy_cus = MFMA busy "
               "cycles / (4 x SQ_BUSY_CU_CYCLES): the same over the CUs that hold a wave.  Calibration (round 1): the input GEMM's "
               "3.17 M v_mfma_f32_32x32x2_f32 of 64 cycles read 204.5 M against 203 M expected.", "workloads": {}}
for d in sorted(glob.glob("gpurun_out/%s_*/" % name)):
    tag = d.rstrip("/").split(name + "_")[1]
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur, seen = collections.defaultdict(int), collections.defaultdict(float), set()
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1) + (m.group(2) or "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[k] += 1
            dur[k] += int(r.get("End_Timestamp", 0) or 0) - int(r.get("Start_Timestamp", 0) or 0)
    per = {}
    for k, t in tot.items():
        busy, gui, cu = t.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), t.get("GRBM_GUI_ACTIVE", 0.0), t.get("SQ_BUSY_CU_CYCLES", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        per[k] = {"launches": cnt[k], "avg_us": round(dur[k] / max(cnt[k], 1) / 1e3, 2),
                  "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": int(busy / cnt[k]), "GRBM_GUI_ACTIVE_per_launch": int(gui / cnt[k]),
                  "SQ_BUSY_CU_CYCLES_per_launch": int(cu / cnt[k]),
                  "SQ_INSTS_VALU_MFMA_MOPS_F32_per_launch": int(t.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) / cnt[k]),
                  "mfma_util": round(busy / (gui / XCDS * SIMDS), 4),
                  "mfma_util_busy_cus": round(busy / (4 * cu), 4) if cu > 0 else None}
    out["workloads"][tag] = per
json.dump(out, open("gpurun_out/%s.json" % name, "w"), indent=1)
for tag, per in out["workloads"].items():
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] * kv[1]["launches"]):
        print("%-14s %-44s x%-5d %9.1f us  util %.3f  (busy CUs %.3f)" % (tag, k[:44], v["launches"], v["avg_us"], v["mfma_util"], v["mfma_util_busy_cus"] or 0))
PY
