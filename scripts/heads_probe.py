"""The five vocabulary heads as one GEMM ([B, 2LH] x [S*V, 2LH]^T + bias): hipBLASLt's default pick (torch.addmm), the library's
fp32 MFMA kernel (dagnn_gemm_nt_bias) and torch's TunableOp search over the rocBLAS / hipBLASLt solutions."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dagnn_amd import engine
dev = torch.device("cuda:0")
B, K, Nc = 128, 1024, 5 * 5002
a = torch.randn(B, K, device=dev); w = torch.randn(Nc, K, device=dev) * 0.05; b = torch.randn(Nc, device=dev)
out = torch.empty(B, Nc, device=dev)
def t_addmm():
    return torch.addmm(b, a, w.t())
def t_mine():
    return engine.gemm_nt_bias([a], [w], [b], out=[out])[0]
def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print("addmm default  %.1f us" % timeit(t_addmm))
print("library kernel %.1f us" % timeit(t_mine))
try:
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(True)
    tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(20)
    t0 = time.time(); t_addmm(); torch.cuda.synchronize(); print("tuning took %.1f s" % (time.time() - t0))
    tun.tuning_enable(False)
    print("addmm tuned    %.1f us" % timeit(t_addmm))
    print(tun.get_results()[-3:])
except Exception as exc:
    print("tunable op:", repr(exc))
