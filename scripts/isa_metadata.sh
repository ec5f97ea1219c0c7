#!/bin/bash
# register / scratch / spill figures of the persistent kernels (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed) and where
# the dominant kernel's spilled SGPRs are touched -> profiles/<round>_isa_metadata.txt
R=${1:-r06}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/profiles/${R}_isa_metadata.txt
: > $out
for f in dataflow dataflow_w bwd_dataflow bwd_dataflow_w tiles fat gemm_f32 wgrad; do
  echo "== $f.hip" >> $out
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /dev/null \
      $root/dagnn_amd/csrc/$f.hip 2>&1 | grep -A8 "Function Name:.*\(dataflow_kernel\|tiles_kernel\|fat_layer\|gemm_nt_bias_k32\|wgrad_partial\)" | grep -v "^--" >> $out
done
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/_df.s $root/dagnn_amd/csrc/dataflow.hip 2>/dev/null
python3 - >> $out <<'PY'
import re
lines = open('/tmp/_df.s').read().split('\n')
a = next(i for i, l in enumerate(lines) if l.startswith('_ZN12_GLOBAL__N_115dataflow_kernelILi16E'))
b = next(i for i in range(a, len(lines)) if lines[i].startswith('.Lfunc_end'))
k = lines[a:b]
from collections import Counter
c = Counter(m.group(1) for l in k for m in [re.match(r'\s*v_writelane_b32 (v\d+),', l)] if m)
spill = c.most_common(1)[0][0] if c else None
print("== dataflow_kernel<16>: where the spilled SGPRs (lanes of %s) are read or written, per top-level loop of the kernel" % spill)
print("   (loops in code order: the placement handshake and the loader variants - lean REC0 / RECP / PROJ, then the generic ones -, and LAST the")
print("    eighteen compute-loop instances: 3 cell kinds x {rows handed over in L2 / write-through} x {no extra output / pre-activations / static rows})")
hdrs = [(i, re.match(r'^(\.LBB\d+_\d+):', l).group(1)) for i, l in enumerate(k) if '=>This Loop Header: Depth=1' in l]
for i, name in hdrs:
    n, cur = 0, False
    for l in k:
        if re.match(r'^\.LBB\d+_\d+:', l):
            cur = ('Header=%s ' % name[2:]) in l + ' ' or l.startswith(name + ':')
        if cur and spill and re.search(r'v_(read|write)lane_b32.*\b%s\b' % spill, l):
            n += 1
    print("   loop %-12s (line %6d of the function): %3d spill accesses in its blocks" % (name, i, n))
PY
tail -30 $out
