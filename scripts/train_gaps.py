"""Developer probe: GPU busy / idle time of the last training step in a rocprofv3 --kernel-trace csv
(step = from one bwd_prepare_kernel to the next), and the idle gaps by the kernel that follows them."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["Start_Timestamp"]) for r in rows if "bwd_prepare_kernel" in r["Kernel_Name"]]
t0, t1 = marks[-2], marks[-1]
step = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
busy, cur_end, gaps = 0, t0, defaultdict(float)
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        gaps[r["Kernel_Name"][:70]] += s - cur_end
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
print("step %.3f ms, %d kernels, union busy %.3f ms, idle %.3f ms" % ((t1 - t0) / 1e6, len(step), busy / 1e6, (t1 - t0 - busy) / 1e6))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:12]:
    print("  idle before %-70s %.1f us" % (k, v / 1e3))
by = defaultdict(float)
for r in step:
    by[r["Kernel_Name"][:70]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("top kernels by time inside the step:")
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:25]:
    print("  %-70s %.1f us" % (k, v / 1e3))
