"""Developer probe: in-kernel phase stamps of workgroup 0 for every launch of the lock-step schedule."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench import build_model, fresh_inputs
from dagnn_amd import engine
from dagnn_amd.synth import code2_batch

dev = torch.device("cuda:0")
model = build_model(256, 2, 5002, 5, dev)
master = code2_batch(0, 128).to(dev)
ins = fresh_inputs(master, 4)
engine.DEBUG_TIMING = torch.zeros(8 * 400, dtype=torch.int64, device=dev)
with torch.no_grad():
    for g in ins:
        model(g)
torch.cuda.synchronize()
t = engine.DEBUG_TIMING.cpu().numpy().reshape(400, 8)[:375].astype(np.float64) / 100.0  # us
names = ["wload-issue", "phaseA", "phaseB-fma", "reduce", "phaseC"]
for lo, hi in [(1, 5), (5, 20), (20, 60), (60, 150), (150, 374)]:
    seg = t[lo:hi]
    d = np.diff(seg[:, :6], axis=1)
    start_gap = np.diff(seg[:, 0])
    print("steps %3d-%3d  WGs %6.0f | " % (lo, hi, seg[:, 6].mean() * 100) +
          "  ".join("%s %.2f" % (n, v) for n, v in zip(names, d.mean(0))) +
          " | in-kernel %.2f  start-to-start %.2f us" % ((seg[:, 5] - seg[:, 0]).mean(), start_gap.mean()))
