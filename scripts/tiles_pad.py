"""Hidden sizes between 256 and 512: per-layer launches at the 64-multiple width against the tile kernel at the width
padded to 512 (`DAGNN_AMD_TILES_PAD`, engine.state_width) - forward, and the whole training step."""
import os, sys, time, copy
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import warnings
warnings.simplefilter("ignore")
from dagnn_amd import engine, synth
from tests import helpers as Hh
from tests.test_gpu_parity import _headline_model, _train_step
dev = torch.device("cuda:0")
PADV = int(os.environ.get("PADV", "1"))
for H in (300, 384, 448):
    for L in [int(v) for v in os.environ.get('LS', '3,5').split(',')]:
        model = _headline_model(H=H, L=L, V=32, seed=5).to(dev)
        for B in (32, 128, 256):
            G = synth.code2_batch(1, B).to(dev)
            res = {}
            for pad in (0, 1):
                engine.TILES_PAD = pad * PADV
                for c in model._derived.values():
                    c.invalidate()
                with torch.no_grad():
                    out = model(G.clone())
                    for _ in range(2):
                        model(G.clone())
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(5):
                        model(G.clone())
                    torch.cuda.synchronize()
                res[pad] = ((time.perf_counter() - t0) / 5 * 1e3, [o.clone() for o in out])
            tr = {}
            y = torch.randint(0, 32, (B, 5), device=dev)
            for pad in (0, 1):
                engine.TILES_PAD = pad * PADV
                for c in model._derived.values():
                    c.invalidate()
                for _ in range(2):
                    _train_step(model, G.clone(), y)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(4):
                    _train_step(model, G.clone(), y)
                torch.cuda.synchronize()
                tr[pad] = (time.perf_counter() - t0) / 4 * 1e3
            model.eval()
            model.check()
            print("H=%d L=%d B=%3d: training step %7.3f -> %7.3f ms (%.2f)" % (H, L, B, tr[0], tr[1], tr[1] / tr[0]))
            print("H=%d L=%d B=%3d: launches %7.3f ms  padded to 512 on the tile kernel %7.3f ms  ratio %.2f  (max diff %.2g)"
                  % (H, L, B, res[0][0], res[1][0], res[1][0] / res[0][0], max(Hh.maxdiff(a, b) for a, b in zip(res[0][1], res[1][1]))), flush=True)
