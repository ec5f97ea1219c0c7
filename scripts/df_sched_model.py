"""Host schedule of the headline batch for G groups: blocks and the time model per group and per workgroup set."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dagnn_amd import synth, host_plan
G = int(sys.argv[1]); cl = int(sys.argv[2]) if len(sys.argv) > 2 else 8; cr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
c0, tau = float(os.environ.get("C0", 3.2)), float(os.environ.get("TAU", 1.48))
B = 128
b = synth.code2_batch(0, B)
ws, sched, splits = host_plan.build_plan_host(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, b.batch, B, b.edge_attr)
N, E = b.batch.shape[0], b.edge_index.shape[1]
R = b.edge_attr.reshape(E, -1).shape[1]
df = host_plan.build_dataflow_schedule_host(ws, N, E, B, R, G, cl, cr)
S = host_plan.dataflow_layout(N, B, G)
loff = df[S["loff"]:S["loff"] + G + 1]
for d in (0, 1):
    chain, blocks = [], []
    for g in range(G):
        pref = df[S["lcnt%d" % d] + loff[g]:S["lcnt%d" % d] + loff[g + 1]].astype(np.int64)
        nb_l = np.diff(pref) // 4
        nb_l = nb_l[nb_l > 0]
        chain.append(np.maximum(c0, tau * nb_l).sum()); blocks.append(nb_l.sum())
    chain, blocks = np.array(chain), np.array(blocks)
    print("dir %d chain-model per group: %s" % (d, " ".join("%.0f" % x for x in chain)))
    print("      blocks per group:       %s" % " ".join("%d" % x for x in blocks))
    if G % 2 == 0:
        pt = [max(chain[2 * k], chain[2 * k + 1], tau * (blocks[2 * k] + blocks[2 * k + 1])) for k in range(G // 2)]
        print("      per set max(chains, tau * blocks): %s" % " ".join("%.0f" % x for x in pt))
