"""cfg 1 (CFG=1: NA, 64 graphs, h=128, L=2, one direction) or cfg 4 (CFG=4: BN, 128 graphs, h=256, L=2, both directions):
200 forwards - the command scripts/prof_small.sh profiles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import DAGNN_NA, DAGNN_BN, synth
dev = torch.device("cuda:0")
torch.manual_seed(0)
if os.environ.get("CFG", "1") == "1":
    m = DAGNN_NA(8, 128, 128, 8, 8, 0, 1, hs=128, nz=56, num_nodes=8, num_layers=2, bidirectional=False).eval().to(dev)
    b = synth.dvae_batch([synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)]).to(dev)
else:
    m = DAGNN_BN(10, 256, 256, 10, 10, 0, 1, hs=256, nz=56, num_nodes=10, num_layers=2, bidirectional=True).eval().to(dev)
    b = synth.dvae_batch([synth.decode_bn_row(r) for r in synth.bn_rows(0, 128)]).to(dev)
n = 200
with torch.no_grad():
    batches = [b.clone() for _ in range(n + 10)]
    for k in range(10):
        m(batches[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        m(batches[10 + k])
    torch.cuda.synchronize()
m.check()
print("cfg %s: %.1f us per forward" % (os.environ.get("CFG", "1"), (time.perf_counter() - t0) / n * 1e6))
