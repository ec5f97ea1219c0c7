"""The S vocabulary heads as one GEMM [B, out_dim] x [S * V, out_dim]^T: torch.addmm (hipBLASLt) against csrc/gemm_f32.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dagnn_amd import engine

dev = torch.device("cuda:0")
torch.manual_seed(0)
for M, Nc, K in ((128, 5 * 5002, 512), (128, 5 * 5002, 1024), (256, 5 * 5002, 2048)):
    a = torch.randn(M, K, device=dev); w = torch.randn(Nc, K, device=dev) * 0.05; b = torch.randn(Nc, device=dev)
    def timed(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); fn(); e.record(); ts.append((s, e))
        torch.cuda.synchronize()
        return sorted(x.elapsed_time(y) for x, y in ts)[n // 2] * 1e3
    t_lib = timed(lambda: torch.addmm(b, a, w.t()))
    t_own = timed(lambda: engine.gemm_nt_bias([a], [w], [b]))
    ref = torch.addmm(b, a, w.t()); own = engine.gemm_nt_bias([a], [w], [b])[0]
    print("M=%d Nc=%d K=%d: addmm %.1f us, gemm_nt_bias %.1f us, max diff %.2e" % (M, Nc, K, t_lib, t_own, float((ref - own).abs().max())))
