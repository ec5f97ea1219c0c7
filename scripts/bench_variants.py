"""Constructor-string variants on the headline batch shape: csrc/variants.hip against the
torch-ops path, forward only.  Not the headline metric."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import DAGNN, ASTNodeEncoder, synth

dev = torch.device("cuda:0")
H = 256
master = synth.code2_batch(0, 128).to(dev)


def timed(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


out = {}
for kw in (dict(agg="gated_sum"), dict(agg="mattn_h"), dict(agg="add"), dict(agg="max"), dict(agg="attn_h", agg_x=True),
           dict(agg="attn_h", recurr=0)):
    torch.manual_seed(0)
    model = DAGNN(num_vocab=5002, max_seq_len=5, emb_dim=H, hidden_dim=H, out_dim=None,
                  encoder=ASTNodeEncoder(H, 98, 10030, 20), num_layers=2, bidirectional=True, out_pool_all=False,
                  **kw).eval().to(dev)
    row = {}
    with torch.no_grad():
        for backend, n, warm in (("hip", 10, 3),) + ((("torch", 2, 1),) if not os.environ.get("VARIANTS_HIP_ONLY") else ()):
            model.variant_backend = backend
            row[backend + "_ms"] = round(timed(lambda: model(master.clone()), n, warm), 2)
    if "torch_ms" in row:
        row["speedup"] = round(row["torch_ms"] / row["hip_ms"], 1)
    out["-".join("%s=%s" % kv for kv in kw.items())] = row
print(json.dumps(out))
