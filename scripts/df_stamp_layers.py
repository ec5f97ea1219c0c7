"""Offline: block stamps of one workgroup (scripts/df_stamps.py DUMP=...) against the schedule of the headline batch:
time per topological layer vs its blocks and in-degrees."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dagnn_amd import synth, host_plan

path, d, group, G = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
grid = G * 6 * 8
B = 128
b = synth.code2_batch(0, B)
ws, sched, splits = host_plan.build_plan_host(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, b.batch, B, b.edge_attr)
N, E = b.batch.shape[0], b.edge_index.shape[1]
R = b.edge_attr.reshape(E, -1).shape[1]
df = host_plan.build_dataflow_schedule_host(ws, N, E, B, R, G, 8, 1)
S = host_plan.dataflow_layout(N, B, G)
loff = df[S["loff"]:S["loff"] + G + 1]
pref = df[S["lcnt%d" % d] + loff[group]:S["lcnt%d" % d] + loff[group + 1]]   # padded exclusive prefix; last = total
first, nblk = df[S["gtab%d" % d] + 2 * group], df[S["gtab%d" % d] + 2 * group + 1]
recs = df[S["grec%d" % d]:S["grec%d" % d] + 16 * (4 * N + 4)].reshape(-1, 16)[first:first + 4 * nblk]
deg = np.where(recs[:, 0] >= 0, recs[:, 2] - recs[:, 1], 0).reshape(nblk, 4)
live = (recs[:, 0] >= 0).reshape(nblk, 4).sum(axis=1)
lay_start_blk = pref[:-1] // 4
nl = len(lay_start_blk)
blk_layer = np.searchsorted(lay_start_blk, np.arange(nblk), side="right") - 1
st = np.load(path)
blk = st[2 * grid:2 * grid + 8 * nblk].reshape(nblk, 8).astype(np.float64) / 100.0
rdy, fma, sto, ldr, lst, lpd = (blk[:, i] for i in range(6))
print("blocks %d layers %d; span %.0f us" % (nblk, nl, sto[-1] - rdy[0]))
# time per layer = stores of its last block - stores of the previous layer's last block
last_blk = np.append(lay_start_blk[1:], nblk) - 1
tl = np.diff(np.concatenate([[rdy[0]], sto[last_blk]]))
nb_l = np.diff(np.append(lay_start_blk, nblk))
maxdeg_l = np.array([deg[lay_start_blk[i]:lay_start_blk[i] + nb_l[i]].max() for i in range(nl)])
for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 1000)):
    sel = (nb_l >= lo) & (nb_l <= hi)
    if sel.any():
        print("layers with %d..%d blocks: %d layers, total %.0f us, per layer %.2f, per block %.2f; model max(3.2, 1.48 nb) total %.0f" %
              (lo, hi, sel.sum(), tl[sel].sum(), tl[sel].mean(), tl[sel].sum() / nb_l[sel].sum(), np.maximum(3.2, 1.48 * nb_l[sel]).sum()))
one = nb_l == 1
for dlo, dhi in ((0, 1), (2, 4), (5, 8), (9, 16), (17, 10000)):
    sel = one & (maxdeg_l >= dlo) & (maxdeg_l <= dhi)
    if sel.any():
        print("  single-block layers with max in-degree %d..%d: %d layers, per layer %.2f us" % (dlo, dhi, sel.sum(), tl[sel].mean()))
trips = np.ceil(np.maximum(deg, 1) / 4).max(axis=1)
print("blocks by loader trips (max over the 4 rows):", {int(t): int((trips == t).sum()) for t in np.unique(trips)})
print("live rows per block:", {int(t): int((live == t).sum()) for t in np.unique(live)})
# hop decomposition over consecutive single-block layers
idx = [int(lay_start_blk[i]) for i in range(1, nl) if nb_l[i] == 1 and nb_l[i - 1] == 1]
idx = np.array(idx)
for lv in (1, 2, 3, 4):
    s = idx[live[idx] == lv]
    if len(s) == 0:
        continue
    f = lambda x: "%.2f" % np.median(x)
    tis = blk[s, 7]
    print("    prev store -> winning poll issued %s | winning poll issued -> done %s" % (f(tis - sto[s - 1]), f(lpd[s] - tis)))
    print("single-block hops, %d live rows (%d): prev store->poll done %s | poll done->signal %s | signal->seen %s | products %s | gates+stores %s | hop %s ; polls %.1f" %
          (lv, len(s), f(lpd[s] - sto[s - 1]), f(ldr[s] - lpd[s]), f(rdy[s] - ldr[s]), f(fma[s] - rdy[s]), f(sto[s] - fma[s]), f(sto[s] - sto[s - 1]),
           np.mean(st[2 * grid:2 * grid + 8 * nblk].reshape(nblk, 8)[s, 6])))
