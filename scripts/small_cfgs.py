"""cfg 1 (NA) and cfg 4 (BN) forward passes alone: median HIP-event time, optionally one traced forward each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import DAGNN_NA, DAGNN_BN, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
na = DAGNN_NA(8, 128, 128, 8, 8, 0, 1, hs=128, nz=56, num_nodes=8, num_layers=2, bidirectional=False).eval().to(dev)
b1 = synth.dvae_batch([synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)]).to(dev)
bn = DAGNN_BN(10, 256, 256, 10, 10, 0, 1, hs=256, nz=56, num_nodes=10, num_layers=2, bidirectional=True).eval().to(dev)
b4 = synth.dvae_batch([synth.decode_bn_row(r) for r in synth.bn_rows(0, 128)]).to(dev)


def timed(fn, n=40, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        ts.append((a, b))
    torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ts)
    return ms[len(ms) // 2], ms[0]


with torch.no_grad():
    print("cfg1 NA  B=64  N=%d: median %.4f ms  min %.4f" % ((b1.x.shape[0],) + timed(lambda: na(b1.clone()))))
    print("cfg4 BN  B=128 N=%d: median %.4f ms  min %.4f" % ((b4.x.shape[0],) + timed(lambda: bn(b4.clone()))))
