"""The reference's own training shape (scripts/ogb_tok.sh: emb_dim 300, batch 160, L = 2, bidirectional, clip 0.25): forward
and whole training step on the current path, and - for scale - the same batch at H = 256 on the dataflow kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import engine, synth
from bench import build_model, fresh_inputs

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 160)); L = int(os.environ.get("L", 2)); V, S = 5002, 5
b = synth.code2_batch(0, B); b.x[:, 1] %= 10030
b = b.to(dev)
print("batch: %d graphs, %d nodes, %d layers" % (B, b.x.shape[0], int(b._bi_layer_idx0.max()) + 1))
y = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(1)).to(dev)
for H in [int(h) for h in os.environ.get("HS", "300,256").split(",")]:
    model = build_model(H, L, V, S, dev)
    ins = iter(fresh_inputs(b, 40))
    with torch.no_grad():
        for _ in range(5):
            model(next(ins))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(next(ins))
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / 20 * 1e3
    model.check()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    ce = torch.nn.CrossEntropyLoss()
    ins = iter(fresh_inputs(b, 30))
    def step():
        opt.zero_grad(set_to_none=True)
        pred = model(next(ins))
        loss = sum(ce(pred[s], y[:, s]) for s in range(S)) / S
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25, foreach=True)
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(12):
        step()
    torch.cuda.synchronize()
    tr = (time.perf_counter() - t0) / 12 * 1e3
    engine.TIMER = engine.KernelTimer()
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    print("   spans per step (ms):", {k: round(n * ms / 6, 3) for k, (n, ms) in engine.TIMER.summary().items()})
    engine.TIMER = None
    model.check()
    print("H=%d (state width %d): forward %.3f ms, training step %.3f ms" % (H, engine.state_width(H, L, 2), fwd, tr), flush=True)
