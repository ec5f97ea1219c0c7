"""Weight-stationary tile kernel (csrc/tiles.hip, H = 512) against the per-layer launch path and the reference fixture:
state rows of every cell on a small batch and on the cfg-5 batch (B = 256, L = 5), then the forward time of both paths.

    python scripts/tiles_check.py [--batch 256] [--layers 5] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dagnn_amd import engine, synth  # noqa: E402
from tests import helpers as Hh  # noqa: E402


def states(model, G):
    with torch.no_grad():
        out = model(G)
    model.check()
    return [o.clone() for o in out], [[h.clone() for h in hs] for hs in G.h]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    # 1. the reference fixture
    meta, arr = Hh.load("code2_h512_L5")
    model = Hh.code2_model(meta).to(dev)
    with torch.no_grad():
        out = model(Hh.code2_batch(arr, dev))
    model.check()
    print("fixture code2_h512_L5: max |hip - reference| = %.3g" % max(Hh.maxdiff(o, r) for o, r in zip(out, arr["pred"])), flush=True)
    # 2. both paths, cell by cell
    from tests.test_gpu_parity import _headline_model
    for B, mean_n in ((6, 30), (a.batch, 125)):
        model = _headline_model(H=512, L=a.layers, V=32, seed=5).to(dev)
        b = synth.code2_batch(3, B, mean_n)
        res = {}
        for mode in (2, 0):
            engine.TILES = mode
            for c in model._derived.values():
                c.invalidate()
            res[mode] = states(model, b.clone().to(dev))
        engine.TILES = 2
        worst = 0.0
        for d in range(2):
            for i in range(a.layers):
                df = float((res[2][1][d][i] - res[0][1][d][i]).abs().max())
                worst = max(worst, df)
                print("  B=%d cell (%d,%d): max |tiles - launches| = %.3g  (|h| max %.3g)" % (B, d, i, df, float(res[0][1][d][i].abs().max())))
        print("B=%d: states %.3g, logits %.3g" % (B, worst, max(Hh.maxdiff(x, y) for x, y in zip(res[2][0], res[0][0]))), flush=True)
        again = states(model, b.clone().to(dev))
        print("  run-to-run bitwise: %s" % all(torch.equal(x, y) for x, y in zip(again[0], res[2][0])))
        for mode in (2, 0):
            engine.TILES = mode
            for c in model._derived.values():
                c.invalidate()
            G = b.clone().to(dev)
            with torch.no_grad():
                for _ in range(3):
                    model(G.clone())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    model(G.clone())
                torch.cuda.synchronize()
            print("  B=%d %s: %.3f ms per forward" % (B, "tiles   " if mode else "launches", (time.perf_counter() - t0) / a.steps * 1e3), flush=True)
        engine.TILES = 2


if __name__ == "__main__":
    main()
