"""D-VAE encoders at the reference's default width (dvae/train.py:55: --hs 501; 512 wide on the lock-step path): the BN
encoder (cfg 4's batch: 128 graphs x 10 nodes, L = 2, both directions) on the tile kernel (one launch) against the
per-layer launches, and the NA encoder (cfg 1's batch; vertex-id keys: per-layer launches only) for reference."""
import os, sys, time, warnings
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
warnings.simplefilter("ignore")
from dagnn_amd import DAGNN_NA, DAGNN_BN, engine, synth
dev = torch.device("cuda:0")


def timed(model, b, steps=50):
    with torch.no_grad():
        for _ in range(5):
            model(b.clone())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(b.clone())
        torch.cuda.synchronize()
    model.check()
    return (time.perf_counter() - t0) / steps * 1e3


for hs in (501,):
    bn = DAGNN_BN(10, hs, hs, 10, 10, 0, 1, hs=hs, nz=56, num_nodes=10, num_layers=2, bidirectional=True).eval().to(dev)
    for B in [int(v) for v in os.environ.get('BS', '128,512').split(',')]:
        b = synth.dvae_batch([synth.decode_bn_row(r) for r in synth.bn_rows(0, B)]).to(dev)
        res = {}
        for mode in (1, 0):
            engine.TILES = mode
            for c in bn._derived.values():
                c.invalidate()
            res[mode] = timed(bn, b)
        engine.TILES = 1
        print("BN hs=%d B=%d (N=%d): default %.3f ms, DAGNN_AMD_TILES=0 %.3f ms" % (hs, B, b.x.shape[0], res[1], res[0]), flush=True)
na = DAGNN_NA(8, 501, 501, 8, 8, 0, 1, hs=501, nz=56, num_nodes=8, num_layers=2, bidirectional=False).eval().to(dev)
b = synth.dvae_batch([synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)]).to(dev)
print("NA hs=501 B=64 (N=%d): %.3f ms" % (b.x.shape[0], timed(na, b)), flush=True)
