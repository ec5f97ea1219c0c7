"""Developer probe: from a rocprofv3 --kernel-trace csv, the span of the per-layer launches and the duration of the
persistent kernel of the last forward (split mode: the two overlap)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tails = [r for r in rows if "frontier_tail_kernel" in r["Kernel_Name"]]
t = tails[-1]
t0, t1 = int(t["Start_Timestamp"]), int(t["End_Timestamp"])
eager = [r for r in rows if any(k in r["Kernel_Name"] for k in ("frontier_step_kernel", "frontier_mfma_kernel", "aggregate_rows_kernel"))
         and t0 - 200000 <= int(r["Start_Timestamp"]) <= t1 + 200000]
e0, e1 = min(int(r["Start_Timestamp"]) for r in eager), max(int(r["End_Timestamp"]) for r in eager)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in eager)
print("tail %.3f ms | per-layer launches: %d kernels, span %.3f ms, busy %.3f ms | union %.3f ms" % (
    (t1 - t0) / 1e6, len(eager), (e1 - e0) / 1e6, busy / 1e6, (max(t1, e1) - min(t0, e0)) / 1e6))
