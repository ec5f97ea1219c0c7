"""How much of the headline recurrence is the deepest graph's chain alone?  The dataflow launch on (a) the headline batch, (b) the
batch without its deepest graph, (c) the deepest graph alone, (d) the two deepest alone."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dagnn_amd import engine, synth
dev = torch.device("cuda:0")
model = bench.build_model(256, 2, 5002, 5, dev)
graphs = synth.code2_graphs(0, 128, 125) if hasattr(synth, "code2_graphs") else None
full = synth.code2_batch(seed=0, num_graphs=128)
n = np.diff(full.ptr.numpy())
li = full._bi_layer_idx0.numpy()
depth = np.array([li[full.ptr[g]:full.ptr[g + 1]].max() + 1 for g in range(128)])
order = np.argsort(-depth)
print("deepest", depth[order[:4]], "nodes", n[order[:4]])
def sub(keep):
    gs = [graphs[g] for g in keep]
    return synth.GraphBatch.from_data_list(gs)
def time_rec(b, label):
    b = b.to(dev)
    ins = bench.fresh_inputs(b, 40)
    timer = engine.KernelTimer(only=["dataflow_run"])
    with torch.no_grad():
        for i in range(10):
            model(ins[i])
        torch.cuda.synchronize()
        engine.TIMER = timer
        for i in range(10, 40):
            model(ins[i])
        torch.cuda.synchronize()
        engine.TIMER = None
    nrec, ms = timer.summary()["dataflow_run"]
    T = int(b._bi_layer_idx0.max()) + 1
    print("%-28s graphs %3d nodes %6d layers %3d  recurrence %.4f ms  (%.2f us per layer)" % (label, b.num_graphs, b.x.shape[0], T, ms, ms / T * 1e3), flush=True)
if graphs is not None:
    chk = sub(list(range(128)))
    assert torch.equal(chk.edge_index, full.edge_index)
    time_rec(chk, "headline batch")
    time_rec(sub([g for g in range(128) if g != order[0]]), "without the deepest graph")
    time_rec(sub([order[0]]), "the deepest graph alone")
    time_rec(sub([order[0], order[1]]), "the two deepest")
    time_rec(sub([g for g in range(128) if g not in order[:2]]), "without the two deepest")
