"""The driver's bench.py contract: one JSON line with the agreed keys (checked on the committed round line here,
and on a live run in the GPU tier)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_line(j, n_gpus, steps, warmup, with_cpu=True):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["metric"] == "graphs/sec" and j["unit"] == "graphs/s" and j["higher_is_better"] is True
    assert j["n_gpus"] == n_gpus and j["steps"] == steps and j["warmup"] == warmup
    assert j["scaling"] == "weak" and j["data"] == "synthetic" and j["dtype"] == "f32" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    # value = graphs of all ranks / max-over-ranks time of the timed steps
    assert abs(j["value"] - j["config"]["global_batch"] / j["ms_per_step"] * 1e3) <= 1e-3 * j["value"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    if with_cpu:
        c = j["cpu_baseline"]
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] in ("port", "reference") and c["unit"] == j["unit"] and c["cores"] >= 1 and c["value"] > 0


@pytest.mark.parametrize("name", ["r01_final_bench.json.log", "r02_final_bench.json.log", "r03_final_bench.json.log",
                                  "r04_final_bench.json.log", "r05_final_bench.json.log", "r06_final_bench.json.log"])
def test_committed_round_line_keeps_the_contract(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip("%s is written at the end of its round" % name)
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    _check_line(j, 1, 30, 5)
    if name.startswith("r02"):   # round 2: where the traffic figure comes from, both CPU legs, the other configurations
        assert j["roofline"]["traffic_source"].startswith("profiles/r02_pmc_traffic.json")
        assert j["roofline"]["schedule"] == "dataflow" and "ms_per_step_median" in j
        assert j["cpu_baseline"]["vectorised"]["value"] > 0 and j["cpu_baseline"]["cpu"]
        assert set(j["other_configs"]) >= {"cfg1_NA_B64_h128_L2_unidir", "cfg4_BN_B128_h256_L2_bidir"}
        assert j["training_step"]["ms_per_step_median"] > 0
    if name.startswith("r03"):   # round 3: traffic tied to the kernel sources, one reverse launch, all three other configurations
        assert j["roofline"]["traffic_source"].startswith("profiles/r03_pmc_traffic.json") and j["roofline"]["traffic"] > 0
        assert j["roofline"]["schedule"] == "dataflow" and j["roofline"]["frac"] > 0.15
        assert set(j["other_configs"]) >= {"cfg1_NA_B64_h128_L2_unidir", "cfg4_BN_B128_h256_L2_bidir", "cfg5_code2_B256_h512_L5_bidir"}
        assert j["training_step"]["kernels_ms_per_step"]["backward_run"] < 3.0
    if name.startswith(("r04", "r05", "r06")):   # rounds 4-6: + the reference's own training shape, the full training step with clip_grad_norm
        rnd = name[:3]
        assert j["roofline"]["traffic_source"].startswith("profiles/%s_pmc_traffic.json" % rnd) and j["roofline"]["traffic"] > 0
        assert j["roofline"]["schedule"] == "dataflow" and j["roofline"]["frac"] > 0.18
        assert set(j["other_configs"]) >= {"cfg1_NA_B64_h128_L2_unidir", "cfg4_BN_B128_h256_L2_bidir", "cfg5_code2_B256_h512_L5_bidir",
                                           "ogb_tok_h300_L2_B160"}
        assert j["training_step"]["kernels_ms_per_step"]["backward_run"] < 2.0 and j["training_step"]["ms_per_step_median"] > 0
        assert j["loader_side_plan"]["ms_per_step"] < j["ms_per_step"]
    if name.startswith("r06"):   # round 6: the leaner kernel, the gather stream's own figure, the training tail through the library's entries
        r = j["roofline"]
        assert r["frac"] > 0.22 and r["recurrence_ms_per_forward"] < 1.2
        assert r["gather_GBps"] > 0 and abs(r["gather_frac_of_8TBps"] - r["gather_GBps"] / 8000.0) < 1e-3
        assert r["traffic_split"]["fetch_bytes"] + r["traffic_split"]["write_bytes"] == r["traffic"]
        assert "seq_cross_entropy" in j["training_step"]["loss"] and "ClipAdam" in j["training_step"]["optimizer"]
        h300 = j["other_configs"]["ogb_tok_h300_L2_B160"]["roofline"]
        assert h300["gflop_per_forward"] < h300["gflop_per_forward_padded"]   # priced on the model's own width
    if name.startswith(("r05", "r06")):   # round 5: the line says how the front of the recurrence is built, and times the unfused / unfolded path beside it
        assert "dagnn_prepare" in j["config"]["front_of_recurrence"] and "folded" in j["config"]["front_of_recurrence"]
        assert j["separate_calls_no_folding"]["ms_per_step"] > j["ms_per_step"] > 0
        assert j["kernels_ms_per_step"]["prepare"] < 0.15


def _run_bench(args, env_extra=None, launcher=()):
    env = dict(os.environ)
    env.update(env_extra or {})
    cmd = [sys.executable] + list(launcher) + [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]   # rank 0 prints ONE line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_gpu_line():
    j = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--cpu-passes", "0", "--train-steps", "0", "--other-configs", "0"])
    _check_line(j, 1, 3, 1, with_cpu=False)


@pytest.mark.gpu
def test_bench_two_ranks_under_torchrun():
    """The driver's N > 1 launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`);
    here two ranks share the one visible GPU and rendezvous over gloo (RCCL refuses two ranks on one device)."""
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29517"]
    j = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-passes", "0", "--train-steps", "2", "--other-configs", "0"],
                   env_extra={"DAGNN_BENCH_BACKEND": "gloo"}, launcher=launcher)
    _check_line(j, 2, 3, 1, with_cpu=False)
    assert j["config"]["global_batch"] == 256   # weak scaling: 128 graphs per rank
    # the line explains itself at N > 1: per-rank group counts / reserved CUs, the rank-seeds weak-scaling entry, the strong-scaling leg
    m = j["multi_gpu"]
    assert [r["rank"] for r in m["per_rank"]] == [0, 1]
    assert all(r["dataflow_groups_inference"] >= r["dataflow_groups_training"] > 0 and r["reserved_cus_training"] in (0, 64)
               for r in m["per_rank"])
    w = m["weak_scaling_rank_seeds"]
    assert w["graphs_per_s"] > 0 and len(w["nodes_layers_per_rank"]) == 2 and w["nodes_layers_per_rank"][0] != w["nodes_layers_per_rank"][1]
    assert j["strong_scaling"]["scaling"] == "strong" and "allreduce_exposed_ms" in j["training_step"]
