"""Developer probe: phase stamps of workgroup 0 of frontier_mfma_kernel for the fat launches."""
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
t = engine.DEBUG_TIMING.cpu().numpy().reshape(400, 8).astype(np.float64) / 100.0
for lo, hi in [(1, 5), (5, 20), (20, 58)]:
    seg = t[lo:hi]
    d = np.diff(seg[:, :6], axis=1).mean(0)
    print("steps %2d-%2d WGs %5.0f | records+prefetch+stage %.2f  mfma %.2f  sync %.2f  tiles->LDS+sync %.2f  gates+store %.2f | in-kernel %.2f  start-to-start %.2f us"
          % ((lo, hi, seg[:, 6].mean() * 100) + tuple(d) + ((seg[:, 5] - seg[:, 0]).mean(), np.diff(seg[:, 0]).mean())))
