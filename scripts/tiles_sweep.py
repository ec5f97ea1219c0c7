"""Tile kernel (csrc/tiles.hip) vs per-layer launches at H = 512 over batch sizes: where each path wins."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dagnn_amd import engine, synth
from tests.test_gpu_parity import _headline_model
dev = torch.device("cuda:0")
import warnings
warnings.simplefilter("ignore")
for L in [int(v) for v in os.environ.get("LS", "5,2").split(",")]:
    model = _headline_model(H=512, L=L, V=32, seed=5).to(dev)
    for B, mean_n in ((8, 60), (16, 125), (32, 125), (64, 125), (128, 125), (256, 125)):
        G = synth.code2_batch(1, B, mean_n).to(dev)
        res = []
        for mode in ((2 if os.environ.get("FORCE") else 1), 0):
            engine.TILES = mode
            for c in model._derived.values():
                c.invalidate()
            with torch.no_grad():
                for _ in range(3):
                    model(G.clone())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    model(G.clone())
                torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / 8 * 1e3)
        engine.TILES = 1
        print("L=%d B=%4d N=%6d: tiles %7.3f ms  launches %7.3f ms  ratio %.2f" % (L, B, G.x.shape[0], res[0], res[1], res[0] / res[1]), flush=True)
