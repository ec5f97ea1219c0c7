"""Soak: the headline forward N times, every result compared bit for bit with the first (B, H, L, ITERS from the environment).
What it is for: hand-off protocol changes (round 6: 16-byte projection granules) - a torn or stale granule shows as a differing bit
or as a raised error word long before it shows in a tolerance test."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import synth
from bench import build_model

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); H = int(os.environ.get("H", 256)); L = int(os.environ.get("L", 2))
iters = int(os.environ.get("ITERS", 2000))
b = synth.code2_batch(0, B)
b.x[:, 1] %= 10030
model = build_model(H, L, 64, 5, dev)
master = b.clone().to(dev)
with torch.no_grad():
    ref = torch.stack(model(master.clone()))
    bad = 0
    for it in range(iters):
        out = torch.stack(model(master.clone()))
        if not torch.equal(out, ref):
            bad += 1
        if it % 256 == 255:
            model.check()
torch.cuda.synchronize()
model.check()
print("B=%d H=%d L=%d: %d forwards, %d differ from the first" % (B, H, L, iters, bad), flush=True)
