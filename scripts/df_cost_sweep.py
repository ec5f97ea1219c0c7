"""Recurrence time of the dataflow kernel by LPT layer cost, over several synthetic batches (seeds / sizes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import engine, synth
from bench import build_model

dev = torch.device("cuda:0")
model = build_model(256, 2, 64, 5, dev)
for B, seed in ((128, 0), (128, 1), (128, 2), (128, 3), (64, 4), (256, 5), (512, 6)):
    b = synth.code2_batch(seed, B)
    b.x[:, 1] %= 10030
    row = []
    for cost in (2, 3, 4, 5, 6, 8, 12):
        engine.DF_COST_LAYER = cost
        engine.TIMER = engine.KernelTimer(only=("dataflow_run",))
        with torch.no_grad():
            for it in range(8):
                model(b.clone().to(dev))
        s = engine.TIMER.summary()
        engine.TIMER = None
        n, ms = s["dataflow_run"]
        row.append("%d: %.3f" % (cost, ms))
    print("B=%d seed=%d N=%d  " % (B, seed, b.x.shape[0]) + "  ".join(row), flush=True)
