"""Time of dagnn_gemm_nt_bias alone on the headline / cfg-5 input products (HIP events over 20 calls)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dagnn_amd import engine
dev = torch.device("cuda:0")
for (M, Nc, K) in ((16561, 768, 256), (31053, 1536, 512), (20168, 960, 320)):
    A = torch.randn(M, K, device=dev)
    W = [torch.randn(Nc, K, device=dev) for _ in range(2)]
    b = [torch.randn(Nc, device=dev) for _ in range(2)]
    for _ in range(3):
        out = engine.gemm_nt_bias([A, A], W, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        out = engine.gemm_nt_bias([A, A], W, b)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    gf = 2 * 2 * M * Nc * K / 1e9
    ref = torch.nn.functional.linear(A.double(), W[0].double(), b[0].double())
    print("M %d Nc %d K %d (x2 groups): %.4f ms = %.1f TF; max err %.2e" % (M, Nc, K, ms, gf / ms, float((out[0].double() - ref).abs().max())))
