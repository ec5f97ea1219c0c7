"""Phase stamps of the dataflow kernel on the headline batch (workgroup 0 = group 0, cell (0,0), slice 0) and the
start / end of every workgroup."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dagnn_amd import engine, synth
from bench import build_model

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); H = int(os.environ.get("H", 256)); L = int(os.environ.get("L", 2))
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
if os.environ.get("DUMP"):   # raw stamps for offline analysis (scripts/df_stamp_layers.py)
    np.save(os.environ["DUMP"], st[:2 * 1024 + 8 * 4096])
cus = torch.cuda.get_device_properties(dev).multi_processor_count
ncell, NS = 2 * (2 * L - 1), H // 32
G = min(cus // (ncell * NS), B)
grid = G * ncell * NS
wg = st[:2 * grid].reshape(grid, 2).astype(np.float64) / 100.0   # us
t0 = wg[:, 0].min()
print("groups %d grid %d; kernel span %.1f us; start skew %.1f us" % (G, grid, wg[:, 1].max() - t0, wg[:, 0].max() - t0))
for k in range(G):
    sel = wg[k * ncell * NS:(k + 1) * ncell * NS]
    ends = [sel[c * NS:(c + 1) * NS, 1].max() - t0 for c in range(ncell)]
    print("  group %d: cell ends (us) %s" % (k, " ".join("%.0f" % e for e in ends)))
blk = st[2 * grid:]
hw = [int(blk[8 * w + 7]) & 0xffffffff for w in range(8)]
print("waves 0..7 (compute 0-3, loaders 4-7): simd %s cu %s" % ([(h >> 4) & 3 for h in hw], [(h >> 8) & 15 for h in hw]))
nb = int((blk[:len(blk) // 8 * 8].reshape(-1, 8)[:, 0] != 0).sum())
polls = blk[:8 * nb].reshape(nb, 8)[:, 6].copy()
blk = blk[:8 * nb].reshape(nb, 8).astype(np.float64) / 100.0
print("workgroup %s: %d blocks" % (os.environ.get("DAGNN_AMD_DEBUG_WG", "0"), nb))
if nb > 2:
    rdy, fma, sto, ldr = blk[:, 0], blk[:, 1], blk[:, 2], blk[:, 3]
    per = np.diff(rdy)
    print("  block period us: mean %.2f med %.2f p90 %.2f; total %.0f" % (per.mean(), np.median(per), np.percentile(per, 90), rdy[-1] - rdy[0]))
    print("  loader signal -> compute sees it: med %.2f" % np.median(rdy - ldr))
    print("  ready -> fma+reduce done: med %.2f" % np.median(fma - rdy))
    print("  reduce done -> stores issued: med %.2f" % np.median(sto - fma))
    print("  stores issued -> next block's loader done: med %.2f p90 %.2f" % (np.median(ldr[1:] - sto[:-1]), np.percentile(ldr[1:] - sto[:-1], 90)))
    lst, lpd = blk[:, 4], blk[:, 5]
    h2 = slice(nb // 2, nb)
    print("  second half: loader start -> poll complete med %.2f; poll complete -> signal med %.2f; polls/block mean %.2f" %
          (np.median((lpd - lst)[h2]), np.median((ldr - lpd)[h2]), polls[h2].mean()))
    print("  second half: prev stores issued -> loader start med %.2f; -> poll complete med %.2f" %
          (np.median((lst[1:] - sto[:-1])[nb // 2:]), np.median((lpd[1:] - sto[:-1])[nb // 2:])))
    print("  second half (thin chain): period med %.2f; store->next loader med %.2f" % (np.median(per[nb // 2:]), np.median((ldr[1:] - sto[:-1])[nb // 2:])))
    q = slice(nb // 8, nb // 4)
    print("  blocks %d..%d: period med %.2f; loader start->poll done med %.2f; poll done->signal %.2f; signal->seen %.2f; fma %.2f; gates %.2f; loader period med %.2f; polls %.2f" %
          (q.start, q.stop, np.median(per[q]), np.median((lpd - lst)[q]), np.median((ldr - lpd)[q]), np.median((rdy - ldr)[q]),
           np.median((fma - rdy)[q]), np.median((sto - fma)[q]), np.median(np.diff(ldr)[q]), polls[q].mean()))
