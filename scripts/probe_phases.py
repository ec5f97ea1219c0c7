"""Developer probe: per-phase time of the deepest (graph, direction) work item of the recurrence kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import build_model, fresh_inputs
from dagnn_amd import engine
from dagnn_amd.synth import code2_batch

dev = torch.device("cuda:0")
model = build_model(256, 2, 5002, 5, dev)
master = code2_batch(0, 128).to(dev)
ins = fresh_inputs(master, 6)
engine.DEBUG_TIMING = torch.zeros(8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for g in ins:
        model(g)
        torch.cuda.synchronize()
        t = engine.DEBUG_TIMING.cpu().tolist()
        n = max(t[3], 1)
        print("layer-launch(last): A %.2f us  B %.2f us  C %.2f us per chunk | chunks %d depth %d total %.1f us" % (
            t[0] / 100 / n, t[1] / 100 / n, t[2] / 100 / n, t[3], t[5], t[4] / 100))
print(torch.cuda.get_device_properties(0))
