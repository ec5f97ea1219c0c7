"""Developer probe: the LAST training step in a rocprofv3 --kernel-trace csv (step = from one bwd_prepare_kernel to the next) as
a timeline - every kernel with its start offset, duration and the idle gap in front of it."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "bwd_prepare_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-2], marks[-1]
step = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
cur = t0
print("step %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(step)))
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, max(0, s - cur) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:90]))
    cur = max(cur, e)
