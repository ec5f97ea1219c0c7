"""Training-side plumbing for one-process-per-GPU data parallelism.

The reference trains with `tg/data_parallel.DataParallel`: one process, k threads, the 29.8 M parameters
re-broadcast and the gradients reduced to device 0 every step (`ogbg-code/tg/data_parallel.py:48-62`).  Here
every rank owns a full replica and its own shard of graphs; the only exchange of a training step is ONE
all-reduce (sum, then divide by the world size) of a flat gradient bucket over RCCL/xGMI.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch
import torch.distributed as dist


class GradBucket(object):
    """All gradients of `params` as views of one contiguous fp32 buffer.

    autograd accumulates into an existing `.grad` in place, so after `backward()` the bucket holds the
    step's gradients without any copy; `all_reduce_mean` is then a single collective of `numel` floats
    (119 MB at the headline configuration) instead of one per parameter."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket needs fp32 parameters on one device")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @property
    def numel(self) -> int:
        return self.flat.numel()

    def zero(self) -> None:
        """Replaces `optimizer.zero_grad()` (which would detach the views when it sets grads to None)."""
        self.flat.zero_()

    def all_reduce_mean(self, group: Optional[dist.ProcessGroup] = None) -> None:
        """Average over ranks: the reference's loss is the mean over the graphs of the GLOBAL batch
        (`main_pyg.py:55-60` on the gathered predictions), which for equal shard sizes is the mean of the
        per-rank mean losses."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
