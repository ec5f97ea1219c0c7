"""Phase stamps of the one-workgroup plan / schedule builds (csrc/small.hip built with -DPS_STAMPS):
    scripts/build_variant.sh stamps SRC=small.hip -DPS_STAMPS && DAGNN_AMD_LIB=scripts/tmp/lib_stamps.so python scripts/small_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dagnn_amd import engine, synth

dev = torch.device("cuda:0")
b1 = synth.dvae_batch([synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)])
b4 = synth.dvae_batch([synth.decode_bn_row(r) for r in synth.bn_rows(0, 128)])
PLAN = ["A stage+checks", "ptr", "B layers/hist", "scan ls", "lstart out", "C node order", "D row hist", "scan rp",
        "rowptr out", "E edge order", "F items", "G blptr", "H lbase", "I rowrec"]
SCHED = ["zero+stage", "assign", "reload", "count+lists", "prefix", "base", "glbase+padding", "tables out", "records", ]
for name, b, G in (("cfg1", b1, 42), ("cfg4", b4, 10)):
    B = int(b.batch.max()) + 1
    bl = b.bi_layer_index
    args = [t.to(dev) for t in (b.edge_index, bl[0][0], bl[1][0], b.batch)]
    for rep in range(6):
        plan = engine.build_plan(*args, B, None)
        sched = plan.dataflow_schedule(G)
        torch.cuda.synchronize()
    from dagnn_amd import host_plan
    lay = host_plan.plan_layout(plan.N, plan.E, plan.B, 0)
    st = plan.ws.cpu().numpy()[lay["cursor0"]:lay["cursor0"] + 32].view(np.uint64)[:15].astype(np.int64)
    ss = sched.cpu().numpy()[-64:].view(np.uint64)[:10].astype(np.int64)
    print(name, "plan  total %.2f us:" % ((st[14] - st[0]) / 100.0), "  ".join("%s %.2f" % (PLAN[i], (st[i + 1] - st[i]) / 100.0) for i in range(14)))
    print(name, "sched total %.2f us:" % ((ss[9] - ss[0]) / 100.0), "  ".join("%s %.2f" % (SCHED[i], (ss[i + 1] - ss[i]) / 100.0) for i in range(9)))
