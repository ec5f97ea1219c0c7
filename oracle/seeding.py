"""Deterministic, torch-RNG-independent parameter fill used by the golden fixtures.

TEST INFRASTRUCTURE (oracle/): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package.

Golden fixtures do not store weights (a H=256 model is >6 MB): both the generating script (which
fills the *reference* model) and the tests (which fill our module and the oracle) call
`seeded_fill(module, seed)`; numpy's PCG64 `default_rng` stream is stable across versions and
platforms, and keys are visited in sorted order, so every party sees identical weights as long
as state_dict keys/shapes agree - which is itself part of the drop-in contract.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def seeded_fill(module: torch.nn.Module, seed: int) -> None:
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    new = {}
    by_storage = {}  # aliased parameters (dvae: cells_0 is grue_forward) must get ONE value
    for key in sorted(sd.keys()):
        t = sd[key]
        if not t.is_floating_point():
            new[key] = t
            continue
        alias = by_storage.get((t.data_ptr(), tuple(t.shape)))
        if alias is not None:
            new[key] = new[alias]
            continue
        by_storage[(t.data_ptr(), tuple(t.shape))] = key
        shape = tuple(t.shape)
        if key.startswith("encoder.") or ".encoder." in key:
            # embedding tables: N(0, 1) like torch.nn.Embedding
            v = rng.standard_normal(shape)
        elif t.dim() >= 2:
            a = 1.0 / math.sqrt(shape[-1])
            if "attn_lin" in key:
                a *= 8.0  # sharpen the attention so the softmax weights are far from uniform
            v = rng.uniform(-a, a, shape)
        else:
            v = rng.uniform(-0.05, 0.05, shape)
        new[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).to(t.dtype)
    module.load_state_dict(new)
