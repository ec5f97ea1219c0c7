"""cfg 5 (B = 256, H = 512, L = 5): per-layer launches, the tile kernel alone, and the split (launches for the wide first layers,
tile kernel for the thin tail) over the split threshold; logits against the launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import warnings
warnings.simplefilter("ignore")
from dagnn_amd import engine, synth
from tests import helpers as Hh
from tests.test_gpu_parity import _headline_model
dev = torch.device("cuda:0")
model = _headline_model(H=512, L=int(os.environ.get("L", "5")), V=32, seed=5).to(dev)
G = synth.code2_batch(0, int(os.environ.get("B", "256"))).to(dev)
engine.TILES_MAX_NODES = int(os.environ.get("MAXN", engine.TILES_MAX_NODES))
def run(mode, rows=None):
    engine.TILES = mode
    if rows is not None:
        engine.TILES_TAIL_ROWS = rows
    for c in model._derived.values():
        c.invalidate()
    with torch.no_grad():
        out = model(G.clone())
        for _ in range(2):
            model(G.clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            model(G.clone())
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    model.check()
    return [o.clone() for o in out], ms
ref, ms0 = run(0)
print("launches: %.3f ms" % ms0, flush=True)
o2, ms2 = run(2)
print("tiles   : %.3f ms  (max diff %.3g)" % (ms2, max(Hh.maxdiff(a, b) for a, b in zip(o2, ref))), flush=True)
for rows in (32, 128):
    o, ms = run(1, rows)
    split = engine.tiles_tail_split(engine.build_plan.__self__ if False else model._last_plan, [0, 1]) if hasattr(model, "_last_plan") else None
    print("split at <= %3d rows: %.3f ms  (max diff %.3g)" % (rows, ms, max(Hh.maxdiff(a, b) for a, b in zip(o, ref))), flush=True)
