"""Host time of the segments of a cfg 1 forward (perf_counter wrappers around the calls forward() makes; asynchronous
forwards, the GPU is never waited for) - what cProfile inflates is measured plainly here."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import DAGNN_NA, synth, engine, core, dvae

acc = collections.defaultdict(float)


def wrap(obj, name, tag=None):
    fn = getattr(obj, name)
    tag = tag or name

    def inner(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[tag] += time.perf_counter() - t0
    setattr(obj, name, inner)


dev = torch.device("cuda:0")
na = DAGNN_NA(8, 128, 128, 8, 8, 0, 1, hs=128, nz=56, num_nodes=8, num_layers=2, bidirectional=False).eval().to(dev)
b1 = synth.dvae_batch([synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)]).to(dev)
for o, n in ((engine, "build_plan"), (engine, "gemm_nt_bias"), (engine, "dataflow_run"), (engine.PlanHandle, "dataflow_schedule"),
             (engine.GranuleArena, "get"), (engine.GranuleArena, "watch"), (engine.GranuleArena, "poll"), (engine, "dataflow_groups"),
             (core, "pack_dataflow"), (dvae, "run_stack")):
    wrap(o, n)
wrap(na, "_cells"); wrap(na, "_readout"); wrap(na, "_training_pass"); wrap(na, "_arena_for")
n = 400
with torch.no_grad():
    batches = [b1.clone() for _ in range(n + 20)]
    for k in range(20):
        na(batches[k])
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    for k in range(n):
        na(batches[20 + k])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
print("forward: %.1f us (host issue)" % ((t1 - t0) / n * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-20s %6.1f us" % (k, v / n * 1e6))
