import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else ".")
import numpy as np, torch
from dagnn_amd import engine, synth
from bench import build_model
dev = torch.device("cuda:0")
b = synth.code2_batch(0, 128); b.x[:, 1] %= 10030
model = build_model(256, 2, 64, 5, dev)
with torch.no_grad(): model(b.clone().to(dev))
torch.cuda.synchronize()
engine.DEBUG_TIMING = torch.zeros(1 << 20, dtype=torch.int64, device=dev)
with torch.no_grad(): model(b.clone().to(dev))
torch.cuda.synchronize()
st = engine.DEBUG_TIMING.cpu().numpy()
blk = st[512:]
nb = int((blk[:len(blk)//8*8].reshape(-1, 8)[:, 0] != 0).sum())
x = blk[:8*nb].reshape(nb, 8).astype(np.float64) / 100
q = slice(nb//8, nb//4)
print("ready->ih done %.2f; ->pass0 fma %.2f; ->pass1 fma %.2f; ->reduce done %.2f" % (np.median((x[:,4]-x[:,0])[q]), np.median((x[:,5]-x[:,4])[q]), np.median((x[:,6]-x[:,5])[q]), np.median((x[:,1]-x[:,6])[q])))
