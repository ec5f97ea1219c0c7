"""Micro-benchmark: weight-gradient GEMM dW[3H,H] = dg[N,3H]^T u[N,H] (K = N rows) - library variants."""
import torch, time
torch.manual_seed(0)
dev = "cuda"
N, H = 16561, 256
dg = torch.randn(N, 3 * H, device=dev)
u = torch.randn(N, H, device=dev)

def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

ref = dg.double().t() @ u.double()
def chk(x): return float((x.double() - ref).abs().max())
f1 = lambda: dg.t() @ u
print("dg.t() @ u           %8.1f us  err %.2e" % (timeit(f1), chk(f1())))
f2 = lambda: (u.t() @ dg).t()
print("(u.t() @ dg).t()     %8.1f us  err %.2e" % (timeit(f2), chk(f2())))
for S in (8, 16, 32):
    n = N // S * S
    def f3():
        p = torch.bmm(dg[:n].view(S, n // S, 3 * H).transpose(1, 2), u[:n].view(S, n // S, H)).sum(0)
        if n < N: p = p + dg[n:].t() @ u[n:]
        return p
    print("split-K bmm S=%-3d     %8.1f us  err %.2e" % (S, timeit(f3), chk(f3())))
try:
    torch.backends.cuda.preferred_blas_library("hipblaslt")
    print("hipblaslt dg.t() @ u %8.1f us  err %.2e" % (timeit(f1), chk(f1())))
    print("hipblaslt (u.t()@dg).t() %8.1f us" % timeit(f2))
except Exception as e:
    print("hipblaslt n/a", e)
sig = torch.randn(N, device=dev); h = torch.randn(N, 272, device=dev)[:, :256]
print("h.t() @ sigma        %8.1f us" % timeit(lambda: h.t() @ sig))
print("(h*sig[:,None]).sum(0) %6.1f us" % timeit(lambda: (h * sig[:, None]).sum(0)))
print("sig[None] @ h        %8.1f us" % timeit(lambda: sig[None] @ h))
