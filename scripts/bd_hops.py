"""Hop decomposition of the reverse sweep (bwd_dataflow_kernel) on the headline batch, run on the GPU box with a -DBD_STAMPS
library (`scripts/build_variant.sh bstamps SRC=bwd_dataflow -DBD_STAMPS`, `DAGNN_AMD_LIB=scripts/tmp/lib_bstamps.so`).

The stamped workgroup is named by its role: workgroup set 0, kernel cell CELL (0 = state-gradient cell of the top stacked layer
of direction 0, 2 = the one of stacked layer 0, 3 / 5 = the same of direction 1 at L = 2), slice 0.  Loader wave 0 and compute
wave 0 stamp every block; blocks are matched against the host mirror of the schedule (the sweep walks a stream's blocks from the
last to the first) and every single-block layer whose successor layer is a single block too (= one dependent hop) is decomposed:
    previous block's stores issued -> winning poll issued -> rows landed -> pulls done -> flag -> compute sees it -> products
    -> stores issued.
"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagnn_amd import engine, synth, host_plan, _lib
from bench import build_model, fresh_inputs

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); H = int(os.environ.get("H", 256)); L = int(os.environ.get("L", 2))
CELL = int(os.environ.get("CELL", 0)); V, S5 = 5002, 5
b = synth.code2_batch(0, B); b.x[:, 1] %= 10030
model = build_model(H, L, V, S5, dev)
model.train()
y = torch.randint(0, V, (B, S5), generator=torch.Generator().manual_seed(1)).to(dev)
ce = torch.nn.CrossEntropyLoss()
ins = iter(fresh_inputs(b.clone().to(dev), 8))


def step():
    model.zero_grad(set_to_none=True)
    pred = model(next(ins))
    loss = sum(ce(pred[s], y[:, s]) for s in range(S5)) / S5
    loss.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
NW = 1 << 20
buf = torch.zeros(NW, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.dagnn_debug_bwd_stamps.argtypes = [ctypes.c_void_p, ctypes.c_uint]
lib.dagnn_debug_bwd_stamps.restype = None
lib.dagnn_debug_bwd_stamps(buf.data_ptr(), (0 << 10) | (CELL << 5) | 0)
step()
torch.cuda.synchronize()
lib.dagnn_debug_bwd_stamps(None, 0)
st = buf.cpu().numpy()
model.check()

d = 0 if CELL < 3 else 1
G = engine.dataflow_groups(dev, 2, L, H, B, training=True)
GRID = int(st[NW - 1])   # (the kernel leaves its grid size in the last word)
ncell, NS, sets = 2 * (2 * L - 1), H // 32, (G + 1) // 2
print("groups %d sets %d grid %d" % (G, sets, GRID))
wg = st[:2 * GRID].reshape(GRID, 2).astype(np.float64) / 100.0
act = wg[:, 0] > 0
t0 = wg[act, 0].min()
print("kernel span %.1f us; start skew %.1f us" % (wg[act, 1].max() - t0, wg[act, 0].max() - t0))

ws, sched, splits = host_plan.build_plan_host(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, b.batch, B, b.edge_attr)
N, E = b.batch.shape[0], b.edge_index.shape[1]
R = b.edge_attr.reshape(E, -1).shape[1]
df = host_plan.build_dataflow_schedule_host(ws, N, E, B, R, G, engine.DF_COST_LAYER, engine.DF_COST_ROW)
SL = host_plan.dataflow_layout(N, B, G)
NLS = 2
raw = st[2 * GRID:]
# out-degree of a node in direction d = its in-degree in direction 1 - d
deg_od = np.zeros(N, dtype=np.int64)
src, dst = b.edge_index[0].numpy(), b.edge_index[1].numpy()
np.add.at(deg_od, src if d == 0 else dst, 1)   # direction 0 aggregates over incoming edges (dst pulls src): successors of src
f = lambda x: "%.2f/%.2f" % (np.median(x), np.percentile(x, 90))
for sidx in range(NLS):
    group = sidx
    loff = df[SL["loff"]:SL["loff"] + G + 1]
    pref = df[SL["lcnt%d" % d] + loff[group]:SL["lcnt%d" % d] + loff[group + 1]]
    first, nblk = df[SL["gtab%d" % d] + 2 * group], df[SL["gtab%d" % d] + 2 * group + 1]
    recs = df[SL["grec%d" % d]:SL["grec%d" % d] + 16 * (4 * N + 4)].reshape(-1, 16)[first:first + 4 * nblk]
    vids = recs[:, 0].reshape(nblk, 4)
    live = (vids >= 0).sum(axis=1)
    sdeg = np.where(vids >= 0, deg_od[np.maximum(vids, 0)], 0)
    indeg = np.where(recs[:, 0] >= 0, recs[:, 2] - recs[:, 1], 0)
    chk = np.zeros(N, dtype=np.int64); np.add.at(chk, dst if d == 0 else src, 1)
    if sidx == 0 and not np.array_equal(indeg[recs[:, 0] >= 0], chk[recs[recs[:, 0] >= 0, 0]]):
        print("  (direction convention flipped: out-degrees taken from the other edge end)")
        deg_od = chk.copy(); np.add.at(deg_od, src if d == 0 else dst, 0); deg_od = np.zeros(N, dtype=np.int64); np.add.at(deg_od, dst if d == 0 else src, 1)
        sdeg = np.where(vids >= 0, deg_od[np.maximum(vids, 0)], 0)
    lay_start = pref[:-1] // 4
    nl = len(lay_start)
    nb_l = np.diff(np.append(lay_start, nblk))
    ent = raw[:16 * NLS * nblk].reshape(nblk, NLS, 16)[:, sidx, :]
    if (ent[:, 0] != 0).sum() < nblk - 2:
        print("stream %d: stamps cover %d of %d blocks - wrong role?" % (sidx, (ent[:, 0] != 0).sum(), nblk))
        continue
    t = ent.astype(np.float64) / 100.0
    # sweep block sb = forward block nblk - 1 - sb
    fwd_of = nblk - 1 - np.arange(nblk)
    row0, slot, trip1, landed, pulled, flag, outs, tissue = (t[:, k] for k in range(8))
    seen, prods, stored = t[:, 9], t[:, 10], t[:, 11]
    npoll = ent[:, 8]
    print("stream %d = group %d: blocks %d layers %d; first block seen %.0f us, last store %.0f us after kernel start (span %.0f)" %
          (sidx, group, nblk, nl, seen[0] - t0, stored[-1] - t0, stored[-1] - seen[0]))
    # per layer (reverse order): time from the previous layer's last store to this layer's last store
    lay_of_blk = np.repeat(np.arange(nl), nb_l)
    last_sweep_blk = nblk - 1 - lay_start           # sweep index of a layer's LAST processed block = its first forward block
    order = np.argsort(last_sweep_blk)
    tl = np.diff(np.concatenate([[seen[0]], stored[last_sweep_blk[order]]]))
    nbo = nb_l[order]
    for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 1000)):
        sel = (nbo >= lo) & (nbo <= hi)
        if sel.any():
            print("  layers with %d..%d blocks: %d layers, total %.0f us, per layer %.2f, per block %.2f" %
                  (lo, hi, sel.sum(), tl[sel].sum(), tl[sel].mean(), tl[sel].sum() / nbo[sel].sum()))
    # hops: forward layer i single block, forward layer i + 1 (processed just before) single block
    hop_f = np.array([int(lay_start[i]) for i in range(nl - 1) if nb_l[i] == 1 and nb_l[i + 1] == 1])
    s = nblk - 1 - hop_f                      # sweep indices; s - 1 = the successor layer's block
    for lv in (1, 2, 3, 4):
        for dlo, dhi in ((0, 1), (2, 4), (5, 10000)):
            md = sdeg[hop_f].max(axis=1)
            k = s[(live[hop_f] == lv) & (md >= dlo) & (md <= dhi)]
            k = k[k >= 1]
            if len(k) < 3:
                continue
            ti = np.where(tissue[k] > 0, tissue[k], trip1[k])
            print("  hops, %d live rows, out-degree %d..%d (%d): [med/p90 us] prev stores->win poll issued %s | issued->landed %s | "
                  "landed->pulled %s | pulled->flag %s | flag->seen %s | products %s | reduce+stores %s | hop %s ; polls %.1f" %
                  (lv, dlo, dhi, len(k), f(ti - stored[k - 1]), f(landed[k] - ti), f(pulled[k] - landed[k]), f(flag[k] - pulled[k]),
                   f(seen[k] - flag[k]), f(prods[k] - seen[k]), f(stored[k] - prods[k]), f(stored[k] - stored[k - 1]), np.mean(npoll[k])))
            print("      loader: row start -> slot free %s | -> first trip back %s | flag -> row outputs issued %s | row start relative to "
                  "previous stores %s (negative: the loader was waiting)" %
                  (f(slot[k] - row0[k]), f(trip1[k] - slot[k]), f(outs[k] - flag[k]), f(row0[k] - stored[k - 1])))
