"""Small probe of the persistent fat launch (csrc/fat.hip): the cfg-5 shape on a 24-graph batch, every layer through the fat
path, against the per-step launches."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dagnn_amd import engine, synth
dev = torch.device("cuda:0")
engine.TILES = 0
engine.MFMA_MIN_ROWS = 1
engine.SPIN_LIMIT = int(os.environ.get("SPIN", "20000"))
model = bench.build_model(512, 5, 32, 5, dev)
b = synth.code2_batch(21, int(os.environ.get("NG", "24"))).to(dev)
outs = {}
for persist in (0, 1):
    engine.FAT_PERSIST = persist
    with torch.no_grad():
        t0 = time.time()
        out = model(b.clone())
        torch.cuda.synchronize()
        print("persist", persist, "forward done in %.2f s" % (time.time() - t0), flush=True)
        try:
            model.check()
        except Exception as exc:
            print("check:", exc, flush=True)
        outs[persist] = [o.clone() for o in out]
    for a in model._arenas.values():
        ws = a.__dict__.get("_fat_ws")
        if ws is not None:
            w = ws.cpu().numpy()
            print("ws header", w[:4].tolist(), "err", a._fat_err.cpu().tolist(), flush=True)
print("max diff", max(float((x - y).abs().max()) for x, y in zip(outs[0], outs[1])))
