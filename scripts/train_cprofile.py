"""Host-side profile (cProfile) of the training step on the headline batch: where the Python time of a step goes."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import engine, synth
from bench import build_model, fresh_inputs

dev = torch.device("cuda:0")
B, H, L, V, S = 128, 256, 2, 5002, 5
b = synth.code2_batch(0, B); b.x[:, 1] %= 10030
b = b.to(dev)
y = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(1)).to(dev)
model = build_model(H, L, V, S, dev); model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
ce = torch.nn.CrossEntropyLoss()
NS = 60
ins = iter(fresh_inputs(b, NS + 10))


def step():
    opt.zero_grad(set_to_none=True)
    pred = model(next(ins))
    loss = sum(ce(pred[s], y[:, s]) for s in range(S)) / S
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25, foreach=True)
    opt.step()


for _ in range(8):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
print("step %.3f ms (no profiler)" % ((time.perf_counter() - t0) / 10 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(40):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
