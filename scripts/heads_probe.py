"""The five vocabulary heads as one GEMM ([B, 2LH] x [S*V, 2LH]^T + bias): hipBLASLt (torch.addmm) against the library's fp32 MFMA
kernel (dagnn_gemm_nt_bias), per call and with a small kernel launched behind it (the bubble behind the library kernel)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagnn_amd import engine
dev = torch.device("cuda:0")
B, K, Nc = 128, 1024, 5 * 5002
for K in (1024, 512):
    a = torch.randn(B, K, device=dev); w = torch.randn(Nc, K, device=dev) * 0.05; b = torch.randn(Nc, device=dev)
    out = torch.empty(B, Nc, device=dev)
    z = torch.zeros(16, device=dev)
    def t_addmm():
        return torch.addmm(b, a, w.t())
    def t_mine():
        return engine.gemm_nt_bias([a], [w], [b], out=[out])[0]
    ref = t_addmm(); got = t_mine()
    print("K", K, "maxdiff", float((ref - got).abs().max()))
    for name, fn in (("addmm", t_addmm), ("mine", t_mine)):
        for tail in (0, 1):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50):
                fn()
                if tail:
                    z.add_(1.0)
            e.record(); torch.cuda.synchronize()
            print("  %-6s tail=%d  %.1f us per call" % (name, tail, s.elapsed_time(e) / 50 * 1e3))
