"""Builds libdagnn_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m dagnn_amd.build            # rebuild if sources are newer than the library

hipcc cross-compiles without a GPU, so this also runs in the build container.  The .so is
git-ignored but travels to the GPU box with the working tree.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.environ.get("DAGNN_AMD_LIB") or os.path.join(LIB_DIR, "libdagnn_hip.so")   # override: experiment builds
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))


def is_stale() -> bool:
    if os.environ.get("DAGNN_AMD_LIB"):
        return False
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in _deps())


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def _compile(src: str, obj: str, verbose: bool) -> None:
    """One object, written next to its final name and renamed on success: two ranks / pytest workers that both find a
    stale library never link each other's half-written object, and a compile killed mid-write leaves no object that
    looks fresh."""
    tmp = "%s.tmp.%d" % (obj, os.getpid())
    cmd = [hipcc_path(), "-O3", "-std=c++17", "--offload-arch=" + ARCH, "-fPIC", "-c", "-o", tmp, src] + \
        os.environ.get("DAGNN_AMD_HIPCC_FLAGS", "").split()
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, res.stdout, res.stderr))
    os.replace(tmp, obj)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (compiled in parallel, re-used while neither the source nor a header changed), then one
    link step; the objects live next to the library and are git-ignored like it."""
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    hdr_time = max([os.path.getmtime(h) for h in _deps() if h.endswith(".h")] + [0.0])
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        src_time = os.path.getmtime(src)
        with open(src) as fh:   # a translation unit that includes another source (dataflow_w.hip <- dataflow.hip) follows it
            for line in fh:
                if line.startswith("#include \"") and line.rstrip().endswith(".hip\""):
                    inc = os.path.join(CSRC, line.split("\"")[1])
                    if os.path.exists(inc):
                        src_time = max(src_time, os.path.getmtime(inc))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(src_time, hdr_time):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(8, max(len(jobs), 1))) as pool:
        for f in [pool.submit(_compile, s_, o_, verbose) for s_, o_ in jobs]:
            f.result()
    tmp = LIB + ".tmp.%d" % os.getpid()
    cmd = [hipcc_path(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
