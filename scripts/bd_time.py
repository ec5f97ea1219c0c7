"""Headline batch, one library (DAGNN_AMD_LIB): spans of a training pass (HIP events around the library calls), several repeats."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import engine, synth
from bench import build_model, fresh_inputs

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 128)); H = int(os.environ.get("H", 256)); L = int(os.environ.get("L", 2)); V, S = 5002, 5
tag = os.environ.get("TAG", "?")
b = synth.code2_batch(0, B); b.x[:, 1] %= 10030
b = b.to(dev)
y = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(1)).to(dev)
model = build_model(H, L, V, S, dev)
model.train()
ce = torch.nn.CrossEntropyLoss()
ins = iter(fresh_inputs(b, 40))


def step():
    model.zero_grad(set_to_none=True)
    pred = model(next(ins))
    loss = sum(ce(pred[s], y[:, s]) for s in range(S)) / S
    loss.backward()
    return loss


for _ in range(4):
    step()
torch.cuda.synchronize()
engine.TIMER = engine.KernelTimer()
for _ in range(20):
    loss = step()
torch.cuda.synchronize()
summ = engine.TIMER.summary()
engine.TIMER = None
model.check()
g = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
print("%-8s backward_run %.4f ms | forward %.4f | prepare %.3f epilogue %.3f | loss %.6f |grad| %.6f" %
      (tag, summ["backward_run"][1], summ["dataflow_run"][1], summ["backward_prepare"][1], summ["backward_epilogue"][1],
       float(loss), float(g.norm())), flush=True)
