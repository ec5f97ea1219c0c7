"""Builds the headline batch's plan (general kernels) a few times: run under rocprofv3 --kernel-trace --stats; with a
-DPG_STAMPS build prints the phase stamps (100 MHz ticks) of the largest graph's workgroups."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagnn_amd import engine, synth
dev = torch.device("cuda:0")
b = synth.code2_batch(seed=0, num_graphs=128)
ptr = b.ptr.clone()
b = b.to(dev)
for _ in range(20):
    plan = engine.build_plan(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, b.batch, 128, b.edge_attr)
    torch.cuda.synchronize()
if os.environ.get("PG_STAMPS"):
    lay = plan.layout()
    n = (ptr[1:] - ptr[:-1])
    ws = plan.ws.cpu().numpy()
    cur_off = {0: None}
    import ctypes
    # cursor offsets are not in layout(): recompute = items + 2B aligned... use the C layout through the known order
    off = lay["items"] + ((2 * 128 + 3) // 4 * 4)      # pos0
    N = int(b.x.shape[0]); al = lambda w: (w + 3) // 4 * 4
    pos0 = off; pos1 = pos0 + al(N); cur0 = pos1 + al(N); cur1 = cur0 + al(N + 128)
    for g in sorted(range(128), key=lambda g: -int(n[g]))[:3] + [5]:
        for d, c in ((0, cur0), (1, cur1)):
            st = ws[c + int(ptr[g]) + g: c + int(ptr[g]) + g + 9].astype("int64")
            print("graph", g, "n", int(n[g]), "dir", d, "phase ticks (10 ns):", list((st[1:] - st[:-1]) % (1 << 31)), "total", int((st[8] - st[0]) % (1 << 31)))
