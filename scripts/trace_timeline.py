"""Timeline of one forward(G) from a rocprofv3 --kernel-trace CSV: kernel, queue, start offset and duration (us).

usage: python scripts/trace_timeline.py <kernel_trace.csv> [anchor-substring] [occurrence]
The anchor picks the forward: the window runs from the previous anchor kernel's end to this one's end."""
import csv
import sys

path = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "dataflow_kernel"
occ = int(sys.argv[3]) if len(sys.argv) > 3 else -2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[occ - 1], idx[occ]
t0 = int(rows[a]["End_Timestamp"])
print("window: %.1f us" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e3))
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:90]))
