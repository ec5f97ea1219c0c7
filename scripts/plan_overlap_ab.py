"""A/B harness of the headline forward pass: each variant = a set of environment knobs, run in its own process (the knobs are read at
import), median of 60 forwards after 10 warm-up passes.  Used for everything in DESIGN.md section 4h (fused pipeline, folded tables,
side / CU-masked streams, row shares, LPT costs, read-back variants).
  python scripts/plan_overlap_ab.py            # driver
"""
import os, subprocess, sys, json

def child():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from dagnn_amd import synth
    dev = torch.device("cuda:0")
    model = bench.build_model(256, 2, 5002, 5, dev)
    b = synth.code2_batch(seed=0, num_graphs=128).to(dev)
    n = 60
    ins = bench.fresh_inputs(b, n + 10)
    it = iter(ins)
    with torch.no_grad():
        for _ in range(10):
            model(next(it))
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); model(next(it)); e.record()
            ts.append((a, e))
        torch.cuda.synchronize()
    ms = sorted(x.elapsed_time(y) for x, y in ts)
    model.check()
    print(json.dumps({"median_ms": round(ms[len(ms) // 2], 4), "min_ms": round(ms[0], 4), "p90": round(ms[int(len(ms) * .9)], 4)}))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        variants = []
        for rep in range(2):   # (edit to taste: every variant is a set of environment knobs read at import)
            variants.append(("default", {}))
            variants.append(("separate_calls_no_folding", {"DAGNN_AMD_PREPARE": "0", "DAGNN_AMD_FOLD_INPUT": "0"}))
            variants.append(("fold_only", {"DAGNN_AMD_PREPARE": "0"}))
            variants.append(("fused_only", {"DAGNN_AMD_FOLD_INPUT": "0"}))
        for name, env in variants:
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
            print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("ERR " + r.stderr[-600:]), flush=True)
