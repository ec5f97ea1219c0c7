"""Phase sums of the fat launches (csrc/fat.hip, -DFAT_STAMPS build): prologue / stage loop / k-half merge / gates,
averaged per workgroup over one cfg-5 forward.  `bash scripts/build_variant.sh stamps SRC=fat -DFAT_STAMPS` first."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dagnn_amd import engine, synth, _lib
from tests.test_gpu_parity import _headline_model
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 256))
model = _headline_model(H=512, L=5, V=32, seed=5).to(dev)
G = synth.code2_batch(0, B).to(dev)
lib = _lib.load()
lib.dagnn_fat_debug_read.restype = C.c_int
lib.dagnn_fat_debug_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(2):
        model(G.clone())
    torch.cuda.synchronize()
    lib.dagnn_fat_debug_read(buf, 1)
    model(G.clone())
    torch.cuda.synchronize()
    lib.dagnn_fat_debug_read(buf, 1)
n = buf[7]
names = ["prologue", "stage loop", "merge pass 2 + barrier", "gates", "epilogue operand issue", "merge pass 1 + barrier"]
print("workgroups %d" % n)
for i, nm in enumerate(names):
    print("  %-24s %8.2f us per workgroup" % (nm, buf[i] / max(n, 1) / 100.0))
print("  total                    %8.2f us" % (sum(buf[i] for i in range(6)) / max(n, 1) / 100.0))
if buf[1]:
    print("  shader clock during the stage loops: %.2f GHz" % (buf[6] / (buf[1] * 10.0)))
print("  other workgroups on the CU at start:", [int(buf[8 + k]) for k in range(4)], " at end:", [int(buf[12 + k]) for k in range(4)])
