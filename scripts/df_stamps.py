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
print("workgroup sets %d grid %d; kernel span %.1f us; start skew %.1f us" % (G, grid, wg[:, 1].max() - t0, wg[:, 0].max() - t0))
for k in range(G):
    sel = wg[k * ncell * NS:(k + 1) * ncell * NS]
    ends = [sel[c * NS:(c + 1) * NS, 1].max() - t0 for c in range(ncell)]
    print("  set %d: cell ends (us) %s" % (k, " ".join("%.0f" % e for e in ends)))
raw = st[2 * grid:]
hw = [int(raw[8 * w + 7]) & 0xffffffff for w in range(8)]
print("waves 0..7 (compute 0-3, loaders 4-7): simd %s cu %s" % ([(h >> 4) & 3 for h in hw], [(h >> 8) & 15 for h in hw]))
NLS = int(os.environ.get("NLS", 2))
ent = raw[:len(raw) // (8 * NLS) * (8 * NLS)].reshape(-1, NLS, 8)   # [block][stream][stamp]
for sidx in range(NLS):
    e = ent[:, sidx, :]
    nb = int((e[:, 0] != 0).sum())
    if nb < 3:
        continue
    polls = e[:nb, 6].copy()
    blk = e[:nb].astype(np.float64) / 100.0
    rdy, fma, sto, ldr, lst, lpd = (blk[:, k] for k in range(6))
    per = np.diff(rdy)
    print("workgroup %s stream %d: %d blocks; first ready %.0f us, last store %.0f us after the kernel start" %
          (os.environ.get("DAGNN_AMD_DEBUG_WG", "0"), sidx, nb, rdy[0] - t0, sto[-1] - t0))
    print("  block period us: mean %.2f med %.2f p90 %.2f; total %.0f" % (per.mean(), np.median(per), np.percentile(per, 90), rdy[-1] - rdy[0]))
    print("  signal -> compute starts: med %.2f p90 %.2f | products med %.2f | gates+stores med %.2f" %
          (np.median(rdy - ldr), np.percentile(rdy - ldr, 90), np.median(fma - rdy), np.median(sto - fma)))
    print("  loader start -> first poll done med %.2f; poll done -> signal med %.2f; polls/block mean %.2f" %
          (np.median(lpd - lst), np.median(ldr - lpd), polls.mean()))
    h2 = slice(nb // 2, nb)
    print("  second half (thin chain): period med %.2f; prev stores -> poll done med %.2f; signal -> compute starts med %.2f" %
          (np.median(per[nb // 2:]), np.median((lpd[1:] - sto[:-1])[nb // 2:]), np.median((rdy - ldr)[h2])))
