"""Developer probe: latency of ONE dependent hop of the recurrence.  A batch of `B` path graphs (node i -> i+1)
has exactly B rows per topological layer, so time / T is the per-layer latency of whichever kernel handles it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dagnn_amd import DAGNN, ASTNodeEncoder, GraphBatch, GraphData, engine
from dagnn_amd.dag_utils import add_order_info_01


def chain_batch(B, n, seed=0, leaves=0, skip=0):
    """B path graphs of n spine nodes; `leaves` > 0 hangs that many leaf children on every spine node, so the
    reverse direction sees a fan-in of leaves + 1 on the dependent chain."""
    rng = np.random.default_rng(seed)
    gs = []
    for _ in range(B):
        ei = np.stack([np.arange(n - 1), np.arange(1, n)])
        if skip:  # ladder: i -> i + 2 as well, fan-in 2 with one row per layer
            ei = np.concatenate([ei, np.stack([np.arange(n - 2), np.arange(2, n)])], 1)
        if leaves:
            src = np.repeat(np.arange(n), leaves)
            ei = np.concatenate([ei, np.stack([src, n + np.arange(n * leaves)])], 1)
        spine, n = n, n * (1 + leaves)
        g = GraphData(x=torch.from_numpy(np.stack([rng.integers(0, 98, n), rng.integers(0, 10030, n)], 1)).long(),
                      node_depth=torch.from_numpy(np.minimum(np.arange(n), 20)).long().view(-1, 1),
                      edge_index=torch.from_numpy(ei).long(), edge_attr=torch.zeros(ei.shape[1], 2))
        n = spine
        add_order_info_01(g)
        gs.append(g)
    return GraphBatch.from_data_list(gs)


def run(B, n, L, bidir, H=256, reps=5, leaves=0, skip=0):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = DAGNN(num_vocab=50, max_seq_len=2, emb_dim=H, hidden_dim=H, out_dim=None,
                  encoder=ASTNodeEncoder(H, 98, 10030, 20), num_layers=L, bidirectional=bidir, out_wx=False,
                  out_pool_all=False, out_pool="max").eval().to(dev)
    master = chain_batch(B, n, leaves=leaves, skip=skip).to(dev)
    best = 1e9
    with torch.no_grad():
        for _ in range(reps):
            g = master.clone()
            engine.TIMER = engine.KernelTimer()
            model(g)
            s = engine.TIMER.summary()
            engine.TIMER = None
            best = min(best, s["frontier_run"][1])
    print("B=%3d chains, %d leaves/node, T=%d, L=%d, %s: frontier_run %.3f ms = %.2f us per topological layer" %
          (B, leaves, n, L, "bidir" if bidir else "unidir", best, best * 1e3 / (n + L - 1)))


if __name__ == "__main__":
    for B in (1, 4, 16):
        run(B, 1000, 2, True)
    run(1, 1000, 1, False)
    run(1, 1000, 1, True)
    run(1, 1000, 2, False)
    for k in (1, 3, 4, 7, 15):
        run(1, 300, 2, True, leaves=k)


def stamps(n=600, L=1, bidir=False, leaves=0):
    """Phase stamps (100 MHz) of workgroup 0 of the persistent tail on one chain: it owns every layer's block."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    H = 256
    model = DAGNN(num_vocab=50, max_seq_len=2, emb_dim=H, hidden_dim=H, out_dim=None,
                  encoder=ASTNodeEncoder(H, 98, 10030, 20), num_layers=L, bidirectional=bidir, out_wx=False,
                  out_pool_all=False, out_pool="max").eval().to(dev)
    master = chain_batch(1, n, leaves=leaves).to(dev)
    engine.DEBUG_TIMING = torch.zeros(8 * (n + 8), dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            model(master.clone())
    torch.cuda.synchronize()
    t = engine.DEBUG_TIMING.cpu().numpy().reshape(-1, 8)[50:n - 50].astype(np.float64) / 100.0
    engine.DEBUG_TIMING = None
    d = np.diff(t[:, :6], axis=1).mean(0)
    print("tail WG0, leaves=%d L=%d: step-to-step %.2f us | to weights-issued %.2f, phase A (records, poll, aggregate) %.2f, "
          "FMA %.2f, reduce+sync %.2f, gates+store %.2f" % ((leaves, L, np.diff(t[:, 0]).mean()) + tuple(d)))


if __name__ == "__main__":
    run(1, 1000, 1, False)
    run(1, 1000, 1, False, skip=1)
    run(1, 1000, 2, False, skip=1)
    stamps()
    stamps(leaves=1, L=1, bidir=True)
