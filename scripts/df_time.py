"""Headline batch, one library (DAGNN_AMD_LIB): recurrence time (HIP events around dagnn_dataflow_run), forward wall time,
and the logits against gpurun_out/_ref_logits.pt (written by the first run that finds none)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import engine, synth
from bench import build_model

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); H = int(os.environ.get("H", 256)); L = int(os.environ.get("L", 2))
tag = os.environ.get("TAG", "?")
b = synth.code2_batch(0, B)
b.x[:, 1] %= 10030
model = build_model(H, L, 64, 5, dev)
with torch.no_grad():
    out = torch.stack(model(b.clone().to(dev)))
torch.cuda.synchronize()
model.check()
ref_path = "gpurun_out/_ref_logits_%d_%d_%d.pt" % (B, H, L)
if os.path.exists(ref_path):
    ref = torch.load(ref_path).to(dev)
    diff = float((out - ref).abs().max())
else:
    torch.save(out.cpu(), ref_path)
    diff = 0.0
with torch.no_grad():
    out2 = torch.stack(model(b.clone().to(dev)))
rep = bool((out2 == out).all())
engine.TIMER = engine.KernelTimer()
ts = []
for it in range(int(os.environ.get("ITERS", 30))):
    G = b.clone().to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        model(G)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
summ = engine.TIMER.summary()
engine.TIMER = None
n, ms = summ.get("dataflow_run", (0, float("nan")))
print("%-10s recurrence %.4f ms | forward min %.3f med %.3f | max|logits - ref| %.2e | run-to-run bitwise %s" %
      (tag, ms, min(ts), sorted(ts)[len(ts) // 2], diff, rep), flush=True)
