"""Host-side cost of a cfg 1 / cfg 4 forward pass (cProfile over asynchronous forwards: the GPU is never waited for)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import DAGNN_NA, DAGNN_BN, synth

dev = torch.device("cuda:0")
torch.manual_seed(0)
na = DAGNN_NA(8, 128, 128, 8, 8, 0, 1, hs=128, nz=56, num_nodes=8, num_layers=2, bidirectional=False).eval().to(dev)
b1 = synth.dvae_batch([synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)]).to(dev)
bn = DAGNN_BN(10, 256, 256, 10, 10, 0, 1, hs=256, nz=56, num_nodes=10, num_layers=2, bidirectional=True).eval().to(dev)
b4 = synth.dvae_batch([synth.decode_bn_row(r) for r in synth.bn_rows(0, 128)]).to(dev)
n = 300
with torch.no_grad():
    for name, m, b in (("cfg1", na, b1), ("cfg4", bn, b4)):
        batches = [b.clone() for _ in range(n + 20)]
        for k in range(20):
            m(batches[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            m(batches[20 + k])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%s: host issue %.1f us / forward, with the final wait %.1f us / forward" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
    batches = [b1.clone() for _ in range(n)]
    pr = cProfile.Profile()
    pr.enable()
    for k in range(n):
        na(batches[k])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
