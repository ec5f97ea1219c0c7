"""Phase sums of the tile kernel (csrc/tiles.hip built with -DT_STAMPS: scripts/build_variant.sh stamps SRC=tiles -DT_STAMPS,
run with DAGNN_AMD_LIB=scripts/tmp/lib_stamps.so): where a workgroup's loaders and compute waves spend the pass.

    DAGNN_AMD_LIB=scripts/tmp/lib_stamps.so python scripts/tiles_stamps.py [--batch 256] [--layers 5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dagnn_amd import engine, synth  # noqa: E402
from tests.test_gpu_parity import _headline_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--layers", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model = _headline_model(H=512, L=a.layers, V=32, seed=5).to(dev)
    b = synth.code2_batch(0, a.batch)
    G = b.clone().to(dev)
    with torch.no_grad():
        for _ in range(3):
            model(G.clone())
        engine.DEBUG_TIMING = torch.zeros(4 * 1024 * 32, dtype=torch.int64, device=dev)
        model(G.clone())
        torch.cuda.synchronize()
    t = engine.DEBUG_TIMING.cpu().view(4, 1024, 32).numpy().astype(float) / 100.0   # us
    engine.DEBUG_TIMING = None
    names = ["total", "poll", "aggregate", "barrier(L)", "tiles", "mfma", "red wait", "barrier(C)", "epilogue", "serial tiles",
             "begin", "w0 wait", "w0 aggregate", "w0 barrier", "mfma (wave 4)", "blocking low polls x100", "S wait low", "S wait own", "S issue", "S consume",
             "S barrier", "S w0 epi+publish", "S w0 wait low", "S w0 wait own"]
    for ch in range(4):
        live = t[ch][:, 0] > 0
        if not live.any():
            continue
        x = t[ch][live]
        print("launch %d: %d workgroups" % (ch, live.sum()))
        for k, nm in enumerate(names):
            if nm == "begin":
                col = x[:, k] - x[:, k].min()
            elif nm in ("tiles", "serial tiles"):
                col = x[:, k] * 100.0
            else:
                col = x[:, k]
            print("  %-14s mean %10.1f  min %10.1f  max %10.1f" % (nm, col.mean(), col.min(), col.max()))
        # per unit (cell): workgroup b -> unit b % units
        tiles = x[:, 4] * 100.0
        ser = x[:, 9] * 100.0
        print("  per SERIAL tile (us): wait low %.2f  wait own %.2f  issue %.2f  consume %.2f  barrier %.2f | wave 0: epilogue + publish %.2f  wait low %.2f  wait own %.2f"
              % tuple((x[:, k] / ser).mean() for k in (16, 17, 18, 19, 20, 21, 22, 23)))
        trace(t, ch)
        print("  per tile (mean over workgroups, us): total %.2f  poll %.2f  aggregate %.2f  barrier(L) %.2f | mfma %.2f  red %.2f  barrier(C) %.2f | epilogue %.2f"
              % tuple((x[:, k] / tiles).mean() for k in (0, 1, 2, 3, 5, 6, 7, 8)))


def trace(t, ch):
    """per-tile stamps of one workgroup (T_TRACE_WG), 256 tiles from T_TRACE_IT on: columns in us relative to the first"""
    import numpy as np
    x = t[ch][512:768]
    live = x[:, 0] > 0
    if not live.any():
        return
    x = x[live]
    t0 = x[0, 0]
    print("trace launch %d: it | C0: mfma-begin mfma-end red-ok barrier-out | C4: same | L0: a0 issue-done epi-done consume-done barrier-out | L3: same | pipelined k t" % ch)
    for r in x[:int(os.environ.get("TRACE_ROWS", "48"))]:
        f = lambda v: "%7.2f" % (v - t0)
        print("  " + " ".join(f(v) for v in r[0:4]) + " | " + " ".join(f(v) for v in r[4:8]) + " | " + " ".join(f(v) for v in r[8:13])
              + " | " + " ".join(f(v) for v in r[16:21]) + " | %d %d %d" % (r[21] * 100, r[22] * 100, r[23] * 100))


if __name__ == "__main__":
    main()
