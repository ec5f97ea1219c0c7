#!/usr/bin/env python
"""Headline benchmark: graphs/sec + ms/batch of DAGNN.forward(G) on code2-like AST batches.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE end-to-end forward(G) over one batch (plan build + AST encoder + L x (input GEMM +
recurrence) + read-out + 5 vocabulary heads), inputs already resident in HBM.  Workload = cfg 2
of BASELINE.json: synthetic ogbg-code2-like ASTs, batch=128, h=256, L=2,
bidirectional, vocab 5002 x 5 heads, fp32.  Graph-parallel weak scaling: every rank runs its own copy
of the 128-graph headline batch (seed 0) - per-GPU work is literally fixed as N grows, so the N > 1
values measure the system, not the depth spread of the synthetic shards (seeds 0..7 have 195..439
topological layers; `--rank-seeds` draws batch `rank` on rank `rank` instead and the slowest shard
then sets the time).  No data-path collective (graphs are independent).

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (the recurrence: one persistent dataflow
launch per forward): duration from HIP events on the launching stream inside the timed region.  `cpu_baseline` is the
oracle's op-for-op restatement of the reference (per-node edge scan, full [N,H] scatter) timed on
this host - a reported baseline, never the thing measured.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
HBM_PEAK_GBPS = 8000.0


def build_model(H, L, V, S, device):
    from dagnn_amd import DAGNN, ASTNodeEncoder
    torch.manual_seed(0)  # random-init weights of the reference architecture (no checkpoints offline)
    enc = ASTNodeEncoder(H, 98, 10030, 20)
    model = DAGNN(num_vocab=V, max_seq_len=S, emb_dim=H, hidden_dim=H, out_dim=None, encoder=enc, w_edge_attr=True,
                  num_layers=L, bidirectional=True, agg="attn_h", out_wx=False, out_pool_all=False, out_pool="max",
                  dropout=0.0).eval()
    return model.to(device)


def fresh_inputs(master, n):
    """forward() mutates G (G.x becomes the embedding, node_depth is clamped): one namespace per step,
    cloned on the device BEFORE the timed region."""
    out = []
    for _ in range(n):
        out.append(SimpleNamespace(
            x=master.x.clone(), node_depth=master.node_depth.clone(), edge_index=master.edge_index,
            edge_attr=master.edge_attr, batch=master.batch, _bi_layer_idx0=master._bi_layer_idx0,
            _bi_layer_index0=master._bi_layer_index0, _bi_layer_idx1=master._bi_layer_idx1,
            _bi_layer_index1=master._bi_layer_index1, num_graphs=master.num_graphs))
    return out


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model_cpu_sd, batch, L, S, passes, threads):
    """The oracle on this host's cores: `faithful` = the reference's algorithm restated op
    for op (per-node edge scan, full [N,H] scatter: its cost profile), `vectorised` = the same arithmetic over a CSR."""
    import copy
    from oracle import dagnn_oracle as O
    if threads > 0:
        torch.set_num_threads(threads)

    def leg(mode, n):
        t0 = time.perf_counter()
        O.code2_forward(model_cpu_sd, copy.deepcopy(batch), num_layers=L, max_seq_len=S, mode=mode)  # warm-up
        warm = time.perf_counter() - t0
        times = []
        for _ in range(n):
            g = copy.deepcopy(batch)
            t0 = time.perf_counter()
            O.code2_forward(model_cpu_sd, g, num_layers=L, max_seq_len=S, mode=mode)
            times.append(time.perf_counter() - t0)
        return sorted(times)[len(times) // 2], warm

    med, warm = leg("faithful", passes)
    vmed, _ = leg("csr", max(passes, 4))
    return {"value": round(batch.num_graphs / med, 2), "unit": "graphs/s", "cores": torch.get_num_threads(),
            "kind": "port", "cpu": _cpu_model(),
            "sample": "%d full forward passes (median) over the same %d-graph seed-0 batch, oracle faithful mode "
                      "(per-node edge scan + full [N,H] scatter as ogbg-code/model/dagnn.py:144-182), torch %s CPU, "
                      "%.2f s/batch, first pass %.1f s" % (passes, batch.num_graphs, torch.__version__, med, warm),
            "ms_per_batch": round(med * 1e3, 1),
            "vectorised": {"value": round(batch.num_graphs / vmed, 2), "unit": "graphs/s", "ms_per_batch": round(vmed * 1e3, 1),
                           "what": "the oracle's CSR form of the same arithmetic (one gather / segment softmax per "
                                   "micro-step instead of the per-node edge scan), same threads"}}


def kernel_sources_sha16():
    """Fingerprint of the sources of the dominant kernel (what `roofline.traffic`'s PMC passes were collected with)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("dataflow.hip", "df_common.h", "common.h"):
        with open(os.path.join(ROOT, "dagnn_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def training_leg(model, inputs, B, S, V, steps, warmup, world, device, barrier):
    """One training step as ogbg-code/main_pyg.py:39-65 runs it: zero_grad, forward, mean-over-heads
    cross-entropy, backward, (N>1: the gradient all-reduce over RCCL in two buckets, the larger one overlapped), Adam step.
    Reported next to the headline forward metric, never as `value`."""
    from dagnn_amd import engine
    model.train()
    from dagnn_amd.train import OverlappedGradReducer
    params = [p for p in model.parameters() if p.requires_grad]
    nparams = sum(p.numel() for p in params)
    # N > 1: gradients live in two flat buckets - the vocabulary heads' (final before the reverse sweep starts: its
    # all-reduce leaves from a hook in the middle of backward() and hides behind the sweep) and the DAGNN core's
    # (exchanged after backward()).  N = 1: the reference's loop as it is (`optimizer.zero_grad()`, main_pyg.py:50).
    heads = list(model.graph_pred_linear_list.parameters()) if hasattr(model, "graph_pred_linear_list") else []
    red = OverlappedGradReducer(params, early=heads) if world > 1 else None
    # the reference's optimizer (main_pyg.py:179: optim.Adam, default hyper-parameters); `fused=True` is the same update
    # as ONE kernel over all parameters instead of five foreach passes
    # N = 1 default: the library's ClipAdam - the same update with the clip coefficient applied as the gradient is read (three
    # launches, csrc/optim.hip; DAGNN_BENCH_ADAM=fused / foreach time torch's pair clip_grad_norm_ + Adam instead)
    ADAM = os.environ.get("DAGNN_BENCH_ADAM", "clipadam" if world == 1 else "fused")
    CLIP = float(os.environ.get("DAGNN_BENCH_CLIP", "0.25"))   # the reference's training script: CLIP=0.25 (scripts/ogb_tok.sh:16)
    if ADAM == "clipadam" and world == 1:
        from dagnn_amd.train import ClipAdam
        opt = ClipAdam(params, lr=1e-3, max_norm=CLIP if CLIP > 0 else None)
    else:
        opt = torch.optim.Adam(params, lr=1e-3, fused=ADAM == "fused")
    y = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(1)).to(device)
    ce = torch.nn.CrossEntropyLoss()
    # the loss of main_pyg.py:55-60 (sum of the S heads' CrossEntropyLoss / S): through the library's one-launch entry
    # (dagnn_amd.train.seq_cross_entropy: same value, csrc/loss.hip) or - DAGNN_BENCH_LOSS=loop - the reference's loop verbatim
    from dagnn_amd.train import seq_cross_entropy
    LOSS_LOOP = os.environ.get("DAGNN_BENCH_LOSS", "fused") == "loop"
    exposed = []

    def step(G, timed=False):
        if red is not None:
            red.zero(local_count=B)   # graphs of this rank's shard: the global-batch mean (train.py)
        else:
            opt.zero_grad(set_to_none=True)
        pred = model(G)
        loss = sum(ce(pred[s], y[:, s]) for s in range(S)) / S if LOSS_LOOP else seq_cross_entropy(pred, y)
        loss.backward()
        if red is not None:
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            red.finish(clip=CLIP)
            if timed:
                b.record()
                exposed.append((a, b))
        elif CLIP > 0 and ADAM != "clipadam":   # main_pyg.py:63-64 (`--clip`, 0.25 in scripts/ogb_tok.sh:16): global-norm clip in front of the update
            torch.nn.utils.clip_grad_norm_(params, CLIP, foreach=True)
        opt.step()   # (ClipAdam: clip + update)
        return loss

    # the warm-up steps run under the same instrumentation as the timed ones (HIP event pairs around the library calls, the
    # exposed-collective events): torch creates its timing events lazily, and twelve hipEventCreate calls inside the FIRST timed
    # step made it 6.2-6.4 ms against 4.8 for every later one (round 6: that was the step's whole "p90 = max" tail)
    # ... and everything slow on the host (garbage collection, event objects) happens BEFORE the last warm-up steps, so that the
    # device is idle for microseconds, not tens of milliseconds, between the synchronisation and the first timed step: after a
    # long idle gap the first step ran at 6.3 ms, the second at 5.0, every later one at 4.8 (clocks ramping back up)
    import gc
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    engine.TIMER = engine.KernelTimer()
    for i in range(warmup):
        if i == max(warmup - 2, 0):
            gc.collect()
            gc.disable()   # a generation-2 collection in the middle of a 12 ms step is a 60-80 ms host stall (measured)
        step(inputs[i], timed=True)
    exposed.clear()
    timer = engine.KernelTimer()
    engine.TIMER = timer
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(warmup, warmup + steps):
        loss = step(inputs[i], timed=True)
        marks[i - warmup + 1].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    engine.TIMER = None
    per_step = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
    if os.environ.get("DAGNN_BENCH_VERBOSE"):
        print("training steps (ms):", " ".join("%.1f" % x for x in per_step), file=sys.stderr)
    per_step = sorted(per_step)
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    summ = timer.summary()
    model.eval()
    extra = {}
    if exposed:
        ex = sorted(a.elapsed_time(b) for a, b in exposed)
        extra = {"allreduce_exposed_ms": round(ex[len(ex) // 2], 4),
                 "allreduce": "two buckets over RCCL: the heads' %.1f M floats leave asynchronously from a hook in the "
                              "middle of backward() (hidden behind the reverse sweep), the core's %.1f M after it; "
                              "allreduce_exposed_ms = median stream time between the end of backward() and both buckets "
                              "reduced and normalised" % (red.early.numel / 1e6 if red.early else 0.0,
                                                          red.late.numel / 1e6 if red.late else 0.0)}
    return {"what": "zero_grad + forward + mean-CE over %d heads + backward%s + clip_grad_norm(%.2f) + Adam step (main_pyg.py:39-65)"
                    % (S, " + RCCL all-reduce of the %.1f M gradient floats in two buckets" % (nparams / 1e6)
                       if world > 1 else "", CLIP),
            "optimizer": "dagnn_amd.train.ClipAdam(max_norm=%.2f): clip_grad_norm_ + Adam's update in three launches "
                         "(DAGNN_BENCH_ADAM=fused times torch's pair)" % CLIP if (ADAM == "clipadam" and world == 1) else
                         "torch.nn.utils.clip_grad_norm_ + torch.optim.Adam(fused=%s)" % (ADAM == "fused"),
            "loss": "the reference's loop over the heads, verbatim (DAGNN_BENCH_LOSS=loop)" if LOSS_LOOP else
                    "dagnn_amd.train.seq_cross_entropy(pred_list, y_arr): the same value as the reference's loop over the heads "
                    "(main_pyg.py:55-60), loss + d logits in one launch (DAGNN_BENCH_LOSS=loop times the loop itself)",
            **extra,
            "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "ms_per_step_median": round(per_step[len(per_step) // 2], 4),
            "ms_per_step_p90": round(per_step[min(len(per_step) - 1, int(0.9 * len(per_step)))], 4),
            "ms_per_step_max": round(per_step[-1], 4),
            "graphs_per_s": round(world * B * steps / elapsed, 1), "final_loss": round(float(loss.detach()), 4),
            "kernels_ms_per_step": {k: round(n * ms / steps, 4) for k, (n, ms) in summ.items()}}


def ogb_tok_config(device, timed):
    """emb_dim 300, B = 160, L = 2, bidirectional: forward(G) end to end and one training step (zero_grad, forward, CE over 5
    heads, backward, clip_grad_norm 0.25, fused Adam), with the roofline of the forward's dataflow launch at H = 320."""
    from dagnn_amd import synth, engine
    H, B, L, V, S = 300, 160, 2, 5002, 5
    model = build_model(H, L, V, S, device)
    master = synth.code2_batch(seed=0, num_graphs=B)
    master.x[:, 1] %= 10030
    N, E = int(master.x.shape[0]), int(master.edge_index.shape[1])
    T = int(master._bi_layer_idx0.max()) + 1
    master = master.to(device)
    with torch.no_grad():
        it = iter(fresh_inputs(master, 30))
        fwd = timed(lambda: model(next(it)), 20, 5)
        engine.TIMER = engine.KernelTimer(only=("dataflow_run",))
        it = iter(fresh_inputs(master, 12))
        for _ in range(10):
            model(next(it))
        torch.cuda.synchronize()
        rec = engine.TIMER.summary().get("dataflow_run", (0, float("nan")))[1]
        engine.TIMER = None
    model.check()
    # the MODEL's algebra (emb_dim = hidden = 300), not the 320-wide rows the launch pads it to: the padding's products are
    # overhead of this implementation, so they must not count as achieved flops (the padded figure rides along for reference)
    Hp = 320
    gf = (2 * L * N * 6.0 * H * H + 2 * (L - 1) * N * 6.0 * H * H) / 1e9
    gf_padded = (2 * L * N * 6.0 * Hp * Hp + 2 * (L - 1) * N * 6.0 * Hp * Hp) / 1e9
    model.train()
    from dagnn_amd.train import ClipAdam
    opt = ClipAdam(model.parameters(), lr=1e-3, max_norm=0.25)   # (= clip_grad_norm_(0.25) + Adam: training_leg)
    from dagnn_amd.train import seq_cross_entropy
    y = torch.randint(0, V, (B, S), generator=torch.Generator().manual_seed(1)).to(device)
    it = iter(fresh_inputs(master, 24))

    def step():
        opt.zero_grad(set_to_none=True)
        pred = model(next(it))
        loss = seq_cross_entropy(pred, y)   # (= sum(ce(pred[s], y[:, s]) for s in range(S)) / S: training_leg)
        loss.backward()
        opt.step()
    train = timed(step, 12, 4)
    model.check()
    model.eval()
    return {"what": "ogbg-code/scripts/ogb_tok.sh: emb_dim = hidden = 300 (320-wide rows), batch 160, L = 2, bidirectional, clip 0.25",
            "nodes": N, "edges": E, "topo_layers": T,
            "forward_ms_per_batch": round(fwd, 4), "forward_graphs_per_s": round(B / fwd * 1e3, 1),
            "training_step_ms": round(train, 4), "training_graphs_per_s": round(B / train * 1e3, 1),
            "roofline": {"kernel": "dataflow_kernel<20> (dagnn_dataflow_run_wide)", "bound": "mfma", "achieved": round(gf / rec, 3),
                         "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(gf / rec / FP32_MATRIX_PEAK_TFLOPS, 5),
                         "flops": "hidden- and input-side products at the model's own width H = 300 (the launch computes them "
                                  "320 wide: gflop_per_forward_padded)",
                         "recurrence_ms_per_forward": round(rec, 4), "gflop_per_forward": round(gf, 2),
                         "gflop_per_forward_padded": round(gf_padded, 2),
                         "us_per_topological_layer": round(rec / T * 1e3, 3)}}


def other_configs(device):
    """BASELINE.json's other configurations, briefly (same code, forward only, inputs resident; medians of HIP-event
    times): cfg 1 (NA: 64 ENAS-shaped DAGs, h=128, L=2, unidirectional), cfg 4 (BN: 128 ten-node DAGs, h=256, L=2,
    bidirectional), cfg 5 (code2-like, B=256, h=512, L=5, bidirectional)."""
    from dagnn_amd import DAGNN_NA, DAGNN_BN, synth

    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            ts.append((a, b))
        torch.cuda.synchronize()
        ms = sorted(x.elapsed_time(y) for x, y in ts)
        return ms[len(ms) // 2]

    out = {}
    with torch.no_grad():
        torch.manual_seed(0)
        na = DAGNN_NA(8, 128, 128, 8, 8, 0, 1, hs=128, nz=56, num_nodes=8, num_layers=2, bidirectional=False).eval().to(device)
        b = synth.dvae_batch([synth.decode_enas_row(r) for r in synth.enas_rows(0, 64)]).to(device)
        def clones(batch, n):   # forward() mutates its batch: every timed call gets its own resident copy, made beforehand
            it = iter([batch.clone() for _ in range(n)])
            return lambda: next(it)
        nb = clones(b, 35)
        ms = timed(lambda: na(nb()), 30, 5)
        def small_roofline(ms, N, D, L, H, Din, T):
            # GRU products of the recurrence: D [2 N (Din + H) 3 H + (L - 1) 12 N H^2]  
            gf = D * (2.0 * N * (Din + H) * 3 * H + (L - 1) * 12.0 * N * H * H) / 1e9
            tf = gf / ms
            return {"bound": "mfma", "achieved": round(tf, 4), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / FP32_MATRIX_PEAK_TFLOPS, 6), "gru_gflop_per_batch": round(gf, 3), "topo_layers": T,
                    "note": "whole forward(G) over the GRU flops: a %d-node batch is launch- and latency-bound (about 25 "
                            "launches around one persistent recurrence kernel of %d dependent layers)" % (N, T)}
        T1 = int(b.bi_layer_index[0][0].max()) + 1
        out["cfg1_NA_B64_h128_L2_unidir"] = {"ms_per_batch": round(ms, 4), "graphs_per_s": round(64 / ms * 1e3, 1),
                                             "roofline": small_roofline(ms, int(b.x.shape[0]), 1, 2, 128, 8, T1)}
        bn = DAGNN_BN(10, 256, 256, 10, 10, 0, 1, hs=256, nz=56, num_nodes=10, num_layers=2, bidirectional=True).eval().to(device)
        b = synth.dvae_batch([synth.decode_bn_row(r) for r in synth.bn_rows(0, 128)]).to(device)
        nb4 = clones(b, 35)
        ms = timed(lambda: bn(nb4()), 30, 5)
        T4 = int(b.bi_layer_index[0][0].max()) + 1
        out["cfg4_BN_B128_h256_L2_bidir"] = {"ms_per_batch": round(ms, 4), "graphs_per_s": round(128 / ms * 1e3, 1),
                                             "roofline": small_roofline(ms, int(b.x.shape[0]), 2, 2, 256, 10, T4)}
        # the same two encoders at the reference's own default width (dvae/train.py:55: --hs 501; 512-wide rows, the tile
        # kernel of csrc/tiles.hip in one launch each) - not a BASELINE configuration, reported next to cfg 1 / cfg 4
        wide = {}
        for tag, cls, nn_, rows, dec, bidir, Bw in (("NA_B64_unidir", DAGNN_NA, 8, synth.enas_rows, synth.decode_enas_row, False, 64),
                                                    ("BN_B128_bidir", DAGNN_BN, 10, synth.bn_rows, synth.decode_bn_row, True, 128)):
            mw = cls(nn_, 501, 501, nn_, nn_, 0, 1, hs=501, nz=56, num_nodes=nn_, num_layers=2, bidirectional=bidir).eval().to(device)
            nbw = clones(synth.dvae_batch([dec(r) for r in rows(0, Bw)]).to(device), 25)
            msw = timed(lambda: mw(nbw()), 20, 5)
            mw.check()
            wide[tag] = {"ms_per_batch": round(msw, 4), "graphs_per_s": round(Bw / msw * 1e3, 1)}
        out["dvae_default_width_hs501_L2"] = wide
    # the reference's own training shape (scripts/ogb_tok.sh:17,63: --emb_dim=300, batch 160, 2 stacked layers, bidirectional,
    # --clip 0.25): hidden sizes 257..320 run 320 wide on the 8-wave shape of the dataflow kernels (csrc/dataflow_w.hip,
    # csrc/bwd_dataflow_w.hip) - forward and the whole training step
    try:
        out["ogb_tok_h300_L2_B160"] = ogb_tok_config(device, timed)
    except Exception as exc:   # (never lose the line over a side entry)
        out["ogb_tok_h300_L2_B160"] = {"error": repr(exc)}
    with torch.no_grad():
        m5 = build_model(512, 5, 5002, 5, device)
        b5 = synth.code2_batch(seed=0, num_graphs=256)
        N5, E5 = b5.x.shape[0], b5.edge_index.shape[1]
        b5 = b5.to(device)
        ins = fresh_inputs(b5, 9)
        it = iter(ins)
        ms = timed(lambda: m5(next(it)), 6, 3)
        gf = 2 * (2 * N5 * (512 + 512) * 3 * 512 + 4 * 12.0 * N5 * 512 * 512) / 1e9
        out["cfg5_code2_B256_h512_L5_bidir"] = {"ms_per_batch": round(ms, 4), "graphs_per_s": round(256 / ms * 1e3, 1),
                                                "gru_gflop_per_batch": round(gf, 1),
                                                "frac_of_fp32_peak": round(gf / ms / FP32_MATRIX_PEAK_TFLOPS, 4),
                                                "roofline": {"bound": "mfma", "achieved": round(gf / ms, 3), "peak": FP32_MATRIX_PEAK_TFLOPS,
                                                             "unit": "TFLOP/s", "frac": round(gf / ms / FP32_MATRIX_PEAK_TFLOPS, 5)},
                                                "path": "per-layer launches for the wide first layers, then the weight-stationary tile kernel "
                                                        "(csrc/tiles.hip) for the thin tail (the dataflow kernels cover h <= 256)"}
        # the weight-stationary tile kernel (csrc/tiles.hip) on the same batch (forced) and on HALF the batch, where it is
        # the default path - each next to the per-layer launches
        from dagnn_amd import engine as _eng
        def both(batch, n):
            r = {}
            for name, mode in (("split_ms", 1), ("tiles_ms", 2), ("launches_ms", 0)):
                old_mode, _eng.TILES = _eng.TILES, mode
                for c in m5._derived.values():
                    c.invalidate()
                it2 = iter(fresh_inputs(batch, n + 3))
                r[name] = round(timed(lambda: m5(next(it2)), n, 3), 4)
                _eng.TILES = old_mode
            for c in m5._derived.values():
                c.invalidate()
            return r
        tk = both(b5, 5)
        half = synth.code2_batch(seed=0, num_graphs=128).to(device)
        tk_half = both(half, 5)
        tk_half["nodes"] = int(half.x.shape[0])
        out["cfg5_code2_B256_h512_L5_bidir"]["tile_kernel"] = {
            "what": "forward(G) end to end on the default path (split_ms: dagnn_frontier_run for the wide first layers + "
                    "dagnn_tiles_run for the thin tail on batches above default_max_nodes, dagnn_tiles_run alone below), on "
                    "dagnn_tiles_run alone (tiles_ms: weights resident in registers, rows in tiles of 16, one persistent launch "
                    "per chunk of stacked layers) and on dagnn_frontier_run alone (launches_ms: a launch per topological layer)",
            "cfg5_batch": tk, "half_batch_B128": tk_half,
            "default_max_nodes": int(_eng.TILES_MAX_NODES)}
        # BASELINE.json quotes cfg 5 on 8 GPUs: under the reference's split of a batch over the devices (tg/dataloader.py:17-27)
        # a rank holds 32 of the 256 graphs - timed here on the first such shard, default policy and per-layer launches
        shard = synth.code2_batch(seed=0, num_graphs=32).to(device)   # (graphs are drawn one after another: the first 32 of the 256)
        tk_shard = both(shard, 5)
        out["cfg5_code2_B256_h512_L5_bidir"]["one_of_eight_shards_B32"] = {
            "what": "graphs 0..31 of the cfg 5 batch (one rank's share on 8 GPUs, no data-path collective): default policy "
                    "(split_ms: here the tile kernel alone), tile kernel forced, per-layer launches",
            "nodes": int(shard.x.shape[0]), **tk_shard,
            "graphs_per_s_x8": round(8 * 32 / tk_shard["split_ms"] * 1e3, 1)}
        del m5
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--vocab", type=int, default=5002)
    ap.add_argument("--cpu-passes", type=int, default=8, help="timed CPU forward passes (~1.4 s each); 0 disables the CPU baseline leg")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--rank-seeds", action="store_true",
                    help="rank r draws batch r (different depths per rank) instead of the headline batch everywhere")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="1: also time BASELINE.json's other configurations briefly (cfg 1 NA, cfg 4 BN, cfg 5 wide/deep) "
                         "and report them under `other_configs` (never as `value`); 0 disables")
    ap.add_argument("--train-steps", type=int, default=20,
                    help="also time this many full training steps (fwd + bwd + optimizer) after the headline "
                         "forward measurement and report them as `training_step`; 0 disables the leg")
    ap.add_argument("--schedule", choices=["lockstep", "pergraph"], default=None,
                    help="recurrence schedule (default: the library default, lock-step frontier launches)")
    ap.add_argument("--streams", type=int, default=1,
                    help="issue the timed forward passes round-robin on this many HIP streams (an evaluation loop with a "
                         "prefetching loader): plan / encoder / GEMMs / heads of one batch overlap the recurrence of "
                         "another; the all-resident persistent launches themselves are ordered device-wide "
                         "(engine.persistent_launch); 1 = strictly one batch after the other (the headline number)")
    ap.add_argument("--cpu-threads", type=int, default=8,
                    help="torch threads for the CPU baseline leg (8 = the thread count BASELINE.md was measured with; "
                         "the op-by-op path is slower with every core of a big host)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP for gfx950 and has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("DAGNN_BENCH_BACKEND", "nccl")  # "gloo": functional check of the N>1 path on fewer GPUs
    if local_rank >= ndev and backend == "nccl":
        raise SystemExit("rank %d has no GPU (%d visible): one process per GPU" % (local_rank, ndev))
    torch.cuda.set_device(local_rank % ndev)
    device = torch.device("cuda", local_rank % ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from dagnn_amd import engine
    from dagnn_amd.synth import code2_batch

    H, L, S, V, B = args.hidden, args.layers, 5, args.vocab, args.batch
    if args.schedule:
        os.environ["DAGNN_AMD_SCHEDULE"] = args.schedule
    # (--streams k: the persistent recurrence launches of different streams never overlap - engine.persistent_launch
    # orders them on the device - so nothing has to be shrunk; what overlaps is plan / encoder / GEMMs / heads of one
    # pass with the recurrence of another)
    model = build_model(H, L, V, S, device)
    # weak scaling: one B-graph batch per rank, by default the same headline batch everywhere (module docstring)
    batch_cpu = code2_batch(seed=rank if args.rank_seeds else 0, num_graphs=B)
    N, E = batch_cpu.x.shape[0], batch_cpu.edge_index.shape[1]
    T = int(batch_cpu._bi_layer_idx0.max()) + 1
    master = batch_cpu.clone().to(device)
    inputs = fresh_inputs(master, args.warmup + args.steps)

    def barrier():
        if world > 1:
            dist.barrier()

    streams = [torch.cuda.Stream(device) for _ in range(args.streams)] if args.streams > 1 else None

    def step(i):
        if streams is None:
            return model(inputs[i])
        with torch.cuda.stream(streams[i % args.streams]):
            return model(inputs[i])

    with torch.no_grad():
        for i in range(args.warmup):
            out = step(i)
        torch.cuda.synchronize()
        # The timed region carries ONE pair of HIP events per step - around the dominant kernel, on the stream it is
        # launched on (roofline.achieved).  Every further event pair is a barrier packet that drains the stream (~8 us
        # each, measured: 2.55 vs 2.41 ms per step with all spans on), so the per-kernel breakdown and the per-step
        # events are taken in a second, separately reported pass below.
        rec_span = "dataflow_run" if (model.schedule == "lockstep" and engine.DATAFLOW) else \
            ("frontier_run" if model.schedule == "lockstep" else "recurrence_layer")
        timer = None if args.no_kernel_timer else engine.KernelTimer(only=[rec_span, "frontier_run"])
        engine.TIMER = timer
        barrier()
        torch.cuda.synchronize()
        import gc
        gc.collect()
        gc.disable()   # no collector pauses inside the timed region (re-enabled right after it)
        t0 = time.perf_counter()
        for i in range(args.warmup, args.warmup + args.steps):
            out = step(i)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
        engine.TIMER = None
        # second pass (not the headline): every launch bracketed + one event per step boundary
        detail, marks = None, None
        if not args.no_kernel_timer and rank == 0 and streams is None:
            inputs2 = fresh_inputs(master, args.steps)
            detail = engine.KernelTimer()
            engine.TIMER = detail
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
            torch.cuda.synchronize()
            gc.collect()
            gc.disable()
            marks[0].record()
            for i in range(args.steps):
                model(inputs2[i])
                marks[i + 1].record()
            torch.cuda.synchronize()
            gc.enable()
            engine.TIMER = None
    assert all(torch.isfinite(o).all() for o in out)
    for a in model._arenas.values():
        a.poll(block=True)   # a device-side failure (expired wait, plan contract) invalidates the run
    per_step = sorted(a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])) if marks else None

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # the CPU baseline and the other configurations are single-process legs: rank 0 at N = 1 only (the other ranks would
    # sit in the final barrier for half a minute)
    cpu_leg = rank == 0 and world == 1 and args.cpu_passes > 0
    cpu_sd = {k: v.detach().cpu() for k, v in model.state_dict().items()} if cpu_leg else None
    # the same forward with the plan built by the loader (dagnn_amd.collate_with_plan): no plan
    # kernels and no device->host read inside the step.  Reported next to the headline, never as `value`.
    planned_res = None
    if args.streams == 1 and model.schedule == "lockstep":
        from dagnn_amd import attach_plan
        groups = engine.dataflow_groups(device, 2, L, (H + 63) // 64 * 64, B)
        pm = attach_plan(batch_cpu.clone(), dataflow_groups=groups, cost_layer=engine.DF_COST_LAYER,
                         cost_row=engine.DF_COST_ROW).to(device)
        pin = fresh_inputs(pm, args.warmup + args.steps)
        for g in pin:
            g._dagnn_plan, g._dagnn_plan_meta = pm._dagnn_plan, pm._dagnn_plan_meta
            g._dagnn_df = getattr(pm, "_dagnn_df", None)
        with torch.no_grad():
            for i in range(args.warmup):
                model(pin[i])
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.warmup, args.warmup + args.steps):
                model(pin[i])
            torch.cuda.synchronize()
            barrier()
            tp = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)
        planned_res = {"what": "forward(G) with the plan and the dataflow schedule built on the host by the loader "
                               "(attach_plan): no plan / schedule kernels inside the step",
                       "ms_per_step": round(float(tp) / args.steps * 1e3, 4),
                       "graphs_per_s": round(world * B * args.steps / float(tp), 1)}
    # consecutive batches on TWO HIP streams (what `--streams 2` times as `value`): the plan / schedule / encoder launches, the heads and
    # the host side of batch i + 1 run beside the recurrence of batch i; the all-resident recurrence launches themselves stay ordered
    # device-wide (engine.persistent_launch).  Reported next to the headline - `value` stays one batch strictly after the other.
    overlap_res = None
    if args.streams == 1 and model.schedule == "lockstep" and rank == 0 and world == 1:
        two = [torch.cuda.Stream(device) for _ in range(2)]
        oin = fresh_inputs(master, args.warmup + args.steps)
        with torch.no_grad():
            for i in range(args.warmup):
                with torch.cuda.stream(two[i % 2]):
                    model(oin[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.warmup, args.warmup + args.steps):
                with torch.cuda.stream(two[i % 2]):
                    model(oin[i])
            torch.cuda.synchronize()
            to_ = time.perf_counter() - t0
        model.check()
        overlap_res = {"what": "the same K forward(G) calls issued alternately on two HIP streams (bench.py --streams 2): the front and the "
                               "heads of one batch beside the recurrence of the other; never `value`",
                       "ms_per_step": round(to_ / args.steps * 1e3, 4), "graphs_per_s": round(B * args.steps / to_, 1)}
    # the same forward on round 5's front of the recurrence: 13 plan / schedule launches, the encoder, the [N, emb] x [emb, 3H]
    # input GEMM of both directions - no fused pipeline, no tables folded through W_ih.  Reported next to the headline so that
    # what the folding and the fusion buy stays visible; never `value`.
    front_res = None
    if args.streams == 1 and model.schedule == "lockstep" and rank == 0:
        saved = (engine.FOLD_INPUT, engine.PREPARE_FUSED, engine.PLAN_OVERLAP)
        engine.FOLD_INPUT, engine.PREPARE_FUSED, engine.PLAN_OVERLAP = 0, 0, 0
        fin = fresh_inputs(master, args.warmup + args.steps)
        with torch.no_grad():
            for i in range(args.warmup):
                model(fin[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.warmup, args.warmup + args.steps):
                model(fin[i])
            torch.cuda.synchronize()
            tf_ = time.perf_counter() - t0
        engine.FOLD_INPUT, engine.PREPARE_FUSED, engine.PLAN_OVERLAP = saved
        front_res = {"what": "forward(G) with DAGNN_AMD_PREPARE=0 DAGNN_AMD_FOLD_INPUT=0 DAGNN_AMD_PLAN_OVERLAP=0: plan and "
                             "dataflow schedule as 13 separate launches (the separate entry points), encoder, input GEMM of stacked layer 0",
                     "ms_per_step": round(tf_ / args.steps * 1e3, 4), "graphs_per_s": round(B * args.steps / tf_, 1)}
    # N > 1: the reference's own use of k devices - ONE global batch split by its Collater rule (tg/dataloader.py:17-27,
    # node-balanced contiguous shards) - next to the weak-scaling headline.  Bounded by the shard holding the deepest graph.
    strong_res = None
    if world > 1 and args.streams == 1:
        from dagnn_amd import collate_sharded
        from dagnn_amd.synth import code2_graphs
        shards = collate_sharded(code2_graphs(0, B), world)
        if len(shards) == world:
            mine = shards[rank]
            sm = mine.clone().to(device)
            sin = fresh_inputs(sm, args.warmup + args.steps)
            with torch.no_grad():
                for i in range(args.warmup):
                    model(sin[i])
                barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.warmup, args.warmup + args.steps):
                    model(sin[i])
                torch.cuda.synchronize()
                barrier()
                ts = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
            info = torch.tensor([mine.num_graphs, mine.x.shape[0], int(mine._bi_layer_idx0.max()) + 1], dtype=torch.int64, device=device)
            allinfo = [torch.zeros_like(info) for _ in range(world)]
            dist.all_gather(allinfo, info)
            strong_res = {"scaling": "strong", "what": "ONE %d-graph batch split over the ranks by the reference's Collater rule "
                          "(node-balanced contiguous shards); time = slowest rank" % B,
                          "ms_per_step": round(float(ts) / args.steps * 1e3, 4),
                          "graphs_per_s": round(B * args.steps / float(ts), 1),
                          "shards_graphs_nodes_layers": [[int(v) for v in a.tolist()] for a in allinfo],
                          "bound": "the rank holding the deepest graph walks all of its topological layers whatever the rank count: "
                                   "time >= max(layers) x the dependent hop of the dataflow kernel (~3 us) - strong scaling of ONE "
                                   "128-graph batch saturates near 1x; ranks scale by taking MORE graphs each (the weak-scaling headline)",
                          "deepest_shard_layers": max(int(a[2]) for a in allinfo)}
    # N > 1: what explains a scaling line - per rank the group count of the persistent launches (inference pass / training pass
    # under the active communicator), the CUs a training pass leaves to the collective, and a second weak-scaling entry on
    # DIFFERENT batches per rank (seed = rank: other depths, the slowest rank binds) next to the same-batch headline
    multi_res = None
    if world > 1 and args.streams == 1:
        from dagnn_amd import engine as _eng
        Hp_ = (H + 63) // 64 * 64
        mine_info = torch.tensor([_eng.dataflow_groups(device, 2, L, Hp_, B), _eng.dataflow_groups(device, 2, L, Hp_, B, training=True),
                                  _eng.reserved_cus(True), _eng._num_cus(device)], dtype=torch.int64, device=device)
        infos = [torch.zeros_like(mine_info) for _ in range(world)]
        dist.all_gather(infos, mine_info)
        multi_res = {"per_rank": [{"rank": r, "dataflow_groups_inference": int(a[0]), "dataflow_groups_training": int(a[1]),
                                   "reserved_cus_training": int(a[2]), "num_cus": int(a[3])} for r, a in enumerate(infos)],
                     "reserved_cus_why": _eng.reserved_cus_info(True)[1] + " (rank 0; DAGNN_AMD_RESERVED_CUS overrides; "
                                         "NCCL_MAX_NCHANNELS=16 would leave all five workgroup sets of the headline shape)",
                     "expected_weak_scaling": "N x the N = 1 value for the forward (no data-path collective, one process per GPU: "
                                              "DESIGN.md section 6 states the predicted N = 2 / 4 / 8 values); a training pass "
                                              "under the communicator runs dataflow_groups_training groups (reserved CUs) instead "
                                              "of dataflow_groups_inference"}
        if not args.rank_seeds:
            rb = code2_batch(seed=rank, num_graphs=B).to(device)
            rin = fresh_inputs(rb, args.warmup + args.steps)
            with torch.no_grad():
                for i in range(args.warmup):
                    model(rin[i])
                barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.warmup, args.warmup + args.steps):
                    model(rin[i])
                torch.cuda.synchronize()
                barrier()
                tr = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
            dist.all_reduce(tr, op=dist.ReduceOp.MAX)
            rinfo = torch.tensor([rb.x.shape[0], int(rb._bi_layer_idx0.max()) + 1], dtype=torch.int64, device=device)
            rall = [torch.zeros_like(rinfo) for _ in range(world)]
            dist.all_gather(rall, rinfo)
            multi_res["weak_scaling_rank_seeds"] = {
                "scaling": "weak", "what": "one %d-graph batch per rank drawn with seed = rank (different depths per rank); "
                                           "time = slowest rank" % B,
                "ms_per_step": round(float(tr) / args.steps * 1e3, 4), "graphs_per_s": round(world * B * args.steps / float(tr), 1),
                "nodes_layers_per_rank": [[int(v) for v in a.tolist()] for a in rall]}
    train_res = None
    if args.train_steps > 0 and args.streams == 1 and model.schedule == "lockstep":
        tw = 5
        train_res = training_leg(model, fresh_inputs(master, tw + args.train_steps), B, S, V, args.train_steps, tw,
                                 world, device, barrier)
    other_res = None
    if args.other_configs and rank == 0 and world == 1 and args.streams == 1:
        other_res = other_configs(device)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    result = {
        "metric": "graphs/sec", "value": round(value, 1), "unit": "graphs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "ogbg-code2-like synthetic ASTs (SURVEY.md App. E, %s), batch=%d h=%d L=%d "
                               "bidirectional attn_h, max-pool over output nodes, %d heads x vocab %d, forward(G) "
                               "end-to-end incl. plan build" % ("seed=rank" if args.rank_seeds else
                                                                "seed 0: the headline batch on every rank", B, H, L, S, V),
                   "front_of_recurrence": "dagnn_prepare: plan + dataflow schedule + encoder rows + side effect 1 in 7 launches; "
                                          "evaluation passes read stacked layer 0's input side from the embedding tables folded "
                                          "through W_ih once per weight version (constant folding of parameters: no input GEMM per "
                                          "batch; `separate_calls_no_folding` times the same forward without either)",
                   "global_batch": world * B, "nodes_per_batch": N, "edges_per_batch": E, "topo_layers": T,
                   "parallelism": "graph-parallel x%d, no data-path collective" % world,
                   "streams_per_gpu": args.streams},
    }
    if rank == 0:
        D = 2
        if timer is not None:
            summ = timer.summary()
            lock = model.schedule == "lockstep"
            df = lock and "dataflow_run" in summ
            n_rec, ms_rec = summ.get("dataflow_run" if df else ("frontier_run" if lock else "recurrence_layer"), (0, 0.0))
            calls_per_step = 1 if lock else L
            # algorithmic work of the recurrence per forward(G), SURVEY.md section 8(d): hidden-side GEMV 2*H*3H per
            # node-update + attention/gates (2NH + 2EH + 15NH); the dataflow / lock-step launches also do the
            # input-side GEMV of stacked layers > 0 (the per-graph schedule leaves it to the batched GEMM)
            flops = D * L * (N * 6.0 * H * H + 2.0 * N * H + 2.0 * E * H + 15.0 * N * H)
            if lock:
                flops += D * (L - 1) * N * 6.0 * H * H
            # compulsory HBM bytes: predecessor rows + own row write ((E+N)*4H), gi read (N*12H), CSR, scores
            byts = D * L * ((E + N) * 4.0 * H + 8.0 * E + 12.0 * N) + D * N * 12.0 * H * (1 if lock else L)
            ms_fwd = ms_rec * calls_per_step  # recurrence time per forward
            if ms_fwd > 0:
                tf = flops / (ms_fwd * 1e-3) / 1e12
                traffic, traffic_source = None, None
                tname = "r06_pmc_traffic.json" if df else "pmc_traffic.json"
                tpath = os.path.join(ROOT, "profiles", tname)
                if lock and os.path.exists(tpath):  # separate rocprofv3 --pmc passes of this command, see profiles/README.md
                    rec = json.load(open(tpath))
                    # the counters belong to the kernel sources they were collected with: a kernel change without a
                    # new PMC run must not leave a stale number in this line
                    if not df or rec.get("kernel_sources_sha16") == kernel_sources_sha16():
                        traffic = rec.get("recurrence_hbm_bytes_per_forward")
                        traffic_source = "profiles/" + tname + " (separate rocprofv3 --pmc passes, not measured in this run)"
                    else:
                        traffic_source = ("profiles/%s is older than the kernel sources (sha %s, now %s): re-run "
                                          "scripts/pmc_traffic.sh" % (tname, rec.get("kernel_sources_sha16"), kernel_sources_sha16()))
                if df:
                    kname = ("dataflow_kernel<H/16> (dagnn_dataflow_run): ONE persistent launch per forward(G) for the "
                             "whole recurrence - every (direction, stacked layer) cell plus the input-side projection "
                             "cells; graphs dealt to independent groups, two groups per workgroup set so that one "
                             "group's dependent hop hides behind the other's blocks; rows handed between workgroups "
                             "as tagged granules (through the shared L2 where a cell and its readers sit on one XCD - "
                             "checked at run time); code specialised per cell variant; products on v_mfma_f32_4x4x1")
                elif lock:
                    kname = ("recurrence = fat_layer_kernel (64-row MFMA tiles, gather fused) + frontier_step_kernel (one launch "
                             "per topological layer and direction) + frontier_tail_kernel / tiles_kernel (persistent, thin tail); "
                             "figures are per forward(G)")
                else:
                    kname = ("recurrence_kernel<KSL> (dagnn_recurrence_layer): %d launches per forward, one per stacked "
                             "GRU layer, persistent per-(graph, direction) workgroups" % L)
                result["roofline"] = {
                    "kernel": kname,
                    "bound": "mfma", "achieved": round(tf, 3), "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / FP32_MATRIX_PEAK_TFLOPS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "forwards_timed": n_rec // calls_per_step, "recurrence_ms_per_forward": round(ms_fwd, 4),
                    "algorithmic_flops_per_forward": flops, "algorithmic_bytes_per_forward": byts,
                    "hbm_frac_of_8TBps": round(byts / (ms_fwd * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                    # the gather stage on its own ("achieved_GBps = D L (E + N) 4H / t_recurrence"): the loader
                    # waves' stream of predecessor rows + the rows written, against the 8 TB/s HBM peak; `traffic_split` (when
                    # the PMC file is current) sets the counters' read and write totals beside it
                    "gather_GBps": round(D * L * (E + N) * 4.0 * H / (ms_fwd * 1e-3) / 1e9, 1),
                    "gather_frac_of_8TBps": round(D * L * (E + N) * 4.0 * H / (ms_fwd * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                    "traffic_split": ({"fetch_bytes": rec.get("fetch_bytes_per_forward"), "write_bytes": rec.get("write_bytes_per_forward"),
                                       "what": "FETCH_SIZE (doubled: MI355X_MICROARCH.md, HBM) = the loaders' polls of predecessor and "
                                               "projection granules that miss the XCD's L2 + weights / records / gi0; WRITE_SIZE = state "
                                               "rows (plain) + their 8-byte granules + the [N, H] 16-byte projection granules {tag, r, z, n}"}
                                      if traffic is not None else None),
                    "us_per_topological_layer": round(ms_fwd * 1e3 / max(T + L - 1, 1), 3),
                    "schedule": "dataflow" if df else model.schedule,
                    "note": "fp32 matrix peak (the path computes in fp32: 1e-4 after ~375 dependent steps); round 6: the kernel is "
                            "bound by the instruction issue of its specialised waves (ONE compute wave per SIMD pays ~5 cycles per "
                            "instruction around 96 products per 4-row block), then by dependent-chain latency (one granule "
                            "hand-off per topological layer of the deepest graph) - not by HBM bandwidth (hbm_frac, gather_GBps)",
                }
        if detail is not None:   # the instrumented pass: spans include their own event cost (~8 us a pair)
            dsum = detail.summary()
            result["kernels_ms_per_step"] = {k: round(n * ms / args.steps, 4) for k, (n, ms) in sorted(dsum.items())}
            result["kernels_ms_per_step"]["_pass"] = "second pass of %d steps with every launch bracketed by HIP events " \
                "(not the headline pass: the events themselves cost ~0.1 ms per step)" % args.steps
        if per_step:
            result["ms_per_step_median"] = round(per_step[len(per_step) // 2], 4)
            result["ms_per_step_p90"] = round(per_step[min(len(per_step) - 1, int(0.9 * len(per_step)))], 4)
            result["timing"] = "value / ms_per_step: wall clock over the K steps between two synchronisations (max over " \
                               "ranks), one HIP event pair per step around the dominant kernel only; median / p90 and " \
                               "kernels_ms_per_step: the instrumented second pass (HIP events around every step and launch)"
        if planned_res is not None:
            result["loader_side_plan"] = planned_res
        if front_res is not None:
            result["separate_calls_no_folding"] = front_res
        if overlap_res is not None:
            result["two_streams"] = overlap_res
        if strong_res is not None:
            result["strong_scaling"] = strong_res
        if multi_res is not None:
            result["multi_gpu"] = multi_res
        if train_res is not None:
            result["training_step"] = train_res
        if other_res is not None:
            result["other_configs"] = other_res
        if cpu_leg:
            result["cpu_baseline"] = cpu_baseline(cpu_sd, batch_cpu, L, S, args.cpu_passes, args.cpu_threads)
        print(json.dumps(result))
    barrier()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
