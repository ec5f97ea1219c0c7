"""Dataflow kernel vs the lock-step launches on the headline batch: agreement and recurrence time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import engine, synth
from bench import build_model

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); H = int(os.environ.get("H", 256)); L = int(os.environ.get("L", 2))
b = synth.code2_batch(0, B)
b.x[:, 1] %= 10030
model = build_model(H, L, 64, 5, dev)
res = {}
for mode in (0, 1):
    engine.DATAFLOW = mode
    for c in model._derived.values():
        c.invalidate()
    with torch.no_grad():
        out = torch.stack(model(b.clone().to(dev)))
    torch.cuda.synchronize()
    for a in model._arenas.values():
        a.poll(block=True)
    res[mode] = out
    engine.TIMER = engine.KernelTimer()
    ts = []
    for it in range(12):
        G = b.clone().to(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            model(G)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    summ = engine.TIMER.summary()
    engine.TIMER = None
    print("DATAFLOW=%d forward ms: min %.3f med %.3f | %s" % (mode, min(ts), sorted(ts)[len(ts) // 2],
          {k: "%dx %.3f ms" % (n, ms) for k, (n, ms) in summ.items()}), flush=True)
print("max |dataflow - lockstep| = %.3e (scale %.3e)" % (float((res[0] - res[1]).abs().max()), float(res[0].abs().max())))
