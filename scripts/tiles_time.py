"""Forward time of the cfg-5 shape with whatever library DAGNN_AMD_LIB names (experiment builds of csrc/tiles.hip)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dagnn_amd import engine, synth
from tests.test_gpu_parity import _headline_model
dev = torch.device("cuda:0")
model = _headline_model(H=512, L=5, V=32, seed=5).to(dev)
G = synth.code2_batch(0, 256).to(dev)
engine.SPIN_LIMIT = 1 << 16
with torch.no_grad():
    for _ in range(2):
        model(G.clone())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        model(G.clone())
    torch.cuda.synchronize()
print("%s: %.3f ms per forward" % (os.environ.get("DAGNN_AMD_LIB", "default"), (time.perf_counter() - t0) / 5 * 1e3), flush=True)
try:
    model.check()
except Exception as e:
    print("  (check: %s)" % str(e)[:80])
