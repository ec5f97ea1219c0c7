"""Time dagnn_amd.autograd._wgrad (dg^T @ u, the weight-gradient products of the backward epilogue) by split count."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd.autograd import _wgrad
dev = torch.device("cuda:0")
N = 16561
dg = torch.randn(N, 768, device=dev); u = torch.randn(N, 256, device=dev)
ref = (dg.double().t() @ u.double()).float()
for splits in (1, 4, 8, 12, 16, 24, 32, 48):
    for _ in range(3): out = _wgrad(dg, u, splits)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = _wgrad(dg, u, splits)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("splits %2d: %.1f us, max err %.2e" % (splits, dt * 1e6, float((out - ref).abs().max())))
