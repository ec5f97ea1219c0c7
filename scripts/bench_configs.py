"""Timings of the other BASELINE.json configurations (not the headline metric): cfg 1 (NA), cfg 4 (BN), cfg 5."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import DAGNN_NA, DAGNN_BN, synth

dev = torch.device("cuda:0")


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
na = DAGNN_NA(8, 128, 128, 8, 8, 0, 1, hs=128, nz=56, num_nodes=8, num_layers=2, bidirectional=False).eval().to(dev)
na_graphs = [synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)]
bn = DAGNN_BN(10, 256, 256, 10, 10, 0, 1, hs=256, nz=56, num_nodes=10, num_layers=2, bidirectional=True).eval().to(dev)
bn_graphs = [synth.decode_bn_row(r) for r in synth.bn_rows(0, 128)]
out = {}
with torch.no_grad():
    b = synth.dvae_batch(na_graphs).to(dev)
    ms = timed(lambda: na(b.clone()))
    out["cfg1_NA_B64_h128_L2_unidir"] = {"ms_per_batch_forward_only": round(ms, 3), "graphs_per_s": round(64 / ms * 1e3)}
    ms = timed(lambda: na.encode(na_graphs))
    out["cfg1_NA_encode_incl_host_collation"] = {"ms_per_batch": round(ms, 3), "graphs_per_s": round(64 / ms * 1e3)}
    b = synth.dvae_batch(bn_graphs).to(dev)
    ms = timed(lambda: bn(b.clone()))
    out["cfg4_BN_B128_h256_L2_bidir"] = {"ms_per_batch_forward_only": round(ms, 3), "graphs_per_s": round(128 / ms * 1e3)}
print(json.dumps(out))
