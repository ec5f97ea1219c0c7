"""Time model of the dataflow kernel on the headline batch: per direction and group count,
   max over groups of  sum_layers max(c0, blocks * tau)  (us)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dagnn_amd import synth

B = 128
b = synth.code2_batch(0, B)
batch = b.batch.numpy()
lay = [b._bi_layer_idx0.numpy(), b._bi_layer_idx1.numpy()]
n_of = np.bincount(batch, minlength=B)
depth = [np.array([lay[d][batch == g].max() + 1 for g in range(B)]) for d in (0, 1)]
D = int(max(depth[0].max(), depth[1].max()))
cnt = [np.zeros((B, D), dtype=np.int64) for _ in (0, 1)]
for d in (0, 1):
    np.add.at(cnt[d], (batch, lay[d]), 1)

def lpt(G, cl, cr, dsel):
    order = np.argsort(-np.maximum(depth[0], depth[1]), kind="stable")
    load = np.zeros(G); empty = np.ones(G, bool); grp = np.zeros(B, int)
    for g in order:
        dg = dsel[g]
        cand = load + cr * n_of[g] + np.where(empty, cl * dg, 0)
        k = int(np.argmin(cand)); load[k] = cand[k]; empty[k] = False; grp[g] = k
    return grp

def model(d, grp, G, c0, tau):
    out = []
    for k in range(G):
        rows = cnt[d][grp == k].sum(axis=0)
        blocks = (rows + 3) // 4
        live = rows > 0
        out.append(float(np.where(live, np.maximum(c0, blocks * tau), 0).sum()))
    return out

c0, tau = float(os.environ.get("C0", 3.2)), float(os.environ.get("TAU", 1.48))
dm = np.maximum(depth[0], depth[1])
for G in (4, 5, 6, 7, 8):
    grp = lpt(G, 8, 1, dm)
    for d in (0, 1):
        t = model(d, grp, G, c0, tau)
        print("G=%d dir %d: max %.0f  mean %.0f  groups %s" % (G, d, max(t), np.mean(t), " ".join("%.0f" % x for x in t)))
for d in (0, 1):
    rows = cnt[d].sum(axis=0)
    print("dir %d: layers %d, rows/layer first 12: %s" % (d, (rows > 0).sum(), rows[:12]))
    print("   graphs alive at layer 10/20/40/80/160/320:", [(cnt[d][:, l] > 0).sum() for l in (10, 20, 40, 80, 160, 320)])
    print("   rows at layer 10/20/40/80/160/320:", [rows[l] for l in (10, 20, 40, 80, 160, 320)])
