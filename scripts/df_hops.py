"""Hop decomposition of the dataflow kernel on the headline batch, run on the GPU box with a -DDF_STAMPS library
(`scripts/build_variant.sh stamps -DDF_STAMPS`, `DAGNN_AMD_LIB=scripts/tmp/lib_stamps.so`).

For the stamped workgroup (DAGNN_AMD_DEBUG_WG: 0 = set 0, direction 0, stacked layer 0, slice 0; 128 = the same cell of
direction 1 under the XCD-aware placement of the headline configuration) the per-block stamps of loader wave 0 and compute
wave 0 are matched against the host mirror of the schedule (word for word the device's), and every single-block layer
whose predecessor layer is a single block too (= one dependent hop) is decomposed:
    previous block's stores issued -> winning poll issued -> poll done -> ready flag -> compute starts -> products done
    -> stores issued.
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagnn_amd import engine, synth, host_plan
from bench import build_model

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); H = int(os.environ.get("H", 256)); L = int(os.environ.get("L", 2))
WG = int(os.environ.get("DAGNN_AMD_DEBUG_WG", 0))
d = int(os.environ.get("DIR", 0)); stream_group = int(os.environ.get("GROUP", 0))
b = synth.code2_batch(0, B); b.x[:, 1] %= 10030
model = build_model(H, L, 64, 5, dev)
with torch.no_grad():
    model(b.clone().to(dev))
torch.cuda.synchronize()
NW = 1 << 20
engine.DEBUG_TIMING = torch.zeros(NW, dtype=torch.int64, device=dev)
with torch.no_grad():
    model(b.clone().to(dev))
torch.cuda.synchronize()
st = engine.DEBUG_TIMING.cpu().numpy()
engine.DEBUG_TIMING = None

cus = torch.cuda.get_device_properties(dev).multi_processor_count
G = engine.dataflow_groups(dev, 2, L, H, B)
ncell, NS, sets = 2 * (2 * L - 1), H // 32, (G + 1) // 2
# grid of the launch: the XCD-aware placement (dataflow.hip: dagnn_dataflow_run) or the linear one
grid = sets * ncell * NS
if engine.DF_XCD:
    cap, fill, top, ok = cus // 8, [0] * 8, 0, True
    units = [2 * NS] * (2 * sets) + [NS] * (2 * sets) if L == 2 else None
    if units is None:
        ok = False
    else:
        for size in units:
            x = 0
            while x < 8 and fill[x] + size > cap:
                x += 1
            if x == 8:
                ok = False
                break
            for _ in range(size):
                top = max(top, fill[x] * 8 + x + 1)
                fill[x] += 1
    if ok:
        grid = top
print("groups %d sets %d grid %d (stamped workgroup %d, direction %d, group %d)" % (G, sets, grid, WG, d, stream_group))

wg = st[:2 * grid].reshape(grid, 2).astype(np.float64) / 100.0
act = wg[:, 0] > 0
t0 = wg[act, 0].min()
print("kernel span %.1f us; start skew %.1f us" % (wg[act, 1].max() - t0, wg[act, 0].max() - t0))

ws, sched, splits = host_plan.build_plan_host(b.edge_index, b._bi_layer_idx0, b._bi_layer_idx1, b.batch, B, b.edge_attr)
N, E = b.batch.shape[0], b.edge_index.shape[1]
R = b.edge_attr.reshape(E, -1).shape[1]
df = host_plan.build_dataflow_schedule_host(ws, N, E, B, R, G, engine.DF_COST_LAYER, engine.DF_COST_ROW)
S = host_plan.dataflow_layout(N, B, G)
NLS = 2
raw = st[2 * grid:]
for sidx in range(NLS):
    group = stream_group + sidx
    loff = df[S["loff"]:S["loff"] + G + 1]
    pref = df[S["lcnt%d" % d] + loff[group]:S["lcnt%d" % d] + loff[group + 1]]
    first, nblk = df[S["gtab%d" % d] + 2 * group], df[S["gtab%d" % d] + 2 * group + 1]
    recs = df[S["grec%d" % d]:S["grec%d" % d] + 16 * (4 * N + 4)].reshape(-1, 16)[first:first + 4 * nblk]
    deg = np.where(recs[:, 0] >= 0, recs[:, 2] - recs[:, 1], 0).reshape(nblk, 4)
    live = (recs[:, 0] >= 0).reshape(nblk, 4).sum(axis=1)
    lay_start_blk = pref[:-1] // 4
    nl = len(lay_start_blk)
    ent = raw[:8 * NLS * nblk].reshape(nblk, NLS, 8)[:, sidx, :]
    xraw = st[2 * grid + (1 << 19):]
    xent = xraw[:8 * NLS * nblk].reshape(nblk, NLS, 8)[:, sidx, :] if xraw[:8 * NLS * 8].any() else None
    blk = ent.astype(np.float64) / 100.0
    rdy, fma, sto, ldr, lst, lpd = (blk[:, i] for i in range(6))
    if (ent[:, 0] != 0).sum() < nblk - 2:
        print("stream %d: stamps cover %d of %d blocks - wrong workgroup?" % (sidx, (ent[:, 0] != 0).sum(), nblk))
        continue
    print("stream %d = group %d: blocks %d layers %d; first ready %.0f us, last store %.0f us after kernel start (span %.0f)" %
          (sidx, group, nblk, nl, rdy[0] - t0, sto[-1] - t0, sto[-1] - rdy[0]))
    last_blk = np.append(lay_start_blk[1:], nblk) - 1
    tl = np.diff(np.concatenate([[rdy[0]], sto[last_blk]]))
    nb_l = np.diff(np.append(lay_start_blk, nblk))
    for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 1000)):
        sel = (nb_l >= lo) & (nb_l <= hi)
        if sel.any():
            print("  layers with %d..%d blocks: %d layers, total %.0f us, per layer %.2f, per block %.2f" %
                  (lo, hi, sel.sum(), tl[sel].sum(), tl[sel].mean(), tl[sel].sum() / nb_l[sel].sum()))
    idx = np.array([int(lay_start_blk[i]) for i in range(1, nl) if nb_l[i] == 1 and nb_l[i - 1] == 1])
    f = lambda x: "%.2f/%.2f" % (np.median(x), np.percentile(x, 90))
    maxdeg = deg.max(axis=1)
    for lv in (1, 2, 3, 4):
        for dlo, dhi in ((0, 1), (2, 4), (5, 10000)):
            s = idx[(live[idx] == lv) & (maxdeg[idx] >= dlo) & (maxdeg[idx] <= dhi)]
            if len(s) < 3:
                continue
            tis = blk[s, 7]
            print("  hops, %d live rows, in-degree %d..%d (%d): [med/p90 us] store->win-poll-issued %s | issued->done %s | done->flag %s | "
                  "flag->seen %s | products %s | gates+stores %s | hop %s ; polls %.1f" %
                  (lv, dlo, dhi, len(s), f(tis - sto[s - 1]), f(lpd[s] - tis), f(ldr[s] - lpd[s]), f(rdy[s] - ldr[s]), f(fma[s] - rdy[s]),
                   f(sto[s] - fma[s]), f(sto[s] - sto[s - 1]), np.mean(ent[s, 6])))
            if xent is not None:
                xb = xent[s].astype(np.float64) / 100.0
                print("      compute: seen->ids %s | ids->operands %s | operands->products %s | products->reduced %s | reduced->gates %s | gates->stores %s" %
                      (f(xb[:, 0] - rdy[s]), f(xb[:, 1] - xb[:, 0]), f(xb[:, 2] - xb[:, 1]), f(fma[s] - xb[:, 2]), f(xb[:, 3] - fma[s]), f(sto[s] - xb[:, 3])))
                print("      loader: poll done->fold done %s | fold done->LDS writes issued %s | ->flag %s" %
                      (f(xb[:, 4] - lpd[s]), f(xb[:, 5] - xb[:, 4]), f(ldr[s] - xb[:, 5])))
            # loader arrival: how long before the data did the loader start polling this block
            print("      loader start -> previous stores: %s (positive: the loader waited for the producer)" % f(sto[s - 1] - lst[s]))
