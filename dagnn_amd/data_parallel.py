"""`DataParallel(list[Batch])`: the caller-side wrapper of the reference (`ogbg-code/tg/data_parallel.py:8-62`,
constructed at `main_pyg.py:295`, fed by `DataListLoader` + `Collater`, `tg/dataloader.py:13-35`).

The reference is ONE process: `forward(data_list)` takes the per-device `Batch` objects its `Collater` made, and
 * with no / one device calls `module(data_list[0].to(src_device))` (`:48-50`),
 * with k devices replicates the module and runs one Python thread per device (`:59-62`).
Here a process owns one GPU, so the k-device case is k processes under `torch.distributed` (RCCL): every rank gets
the same `data_list` from its loader (same sampler seed, as the reference's single loader) and runs ITS element;
the outputs stay on the rank - there is no gather to device 0, the loss is local and `reduce_gradients` is the one
collective of the step (`train.GradBucket`).  Same constructor, `.module`, and `module.`-prefixed `state_dict`
(`utils2.py:85-102` checkpoints load either way).
"""
from __future__ import annotations

import warnings
from typing import Optional, Sequence

import torch
import torch.distributed as dist

from .train import GradBucket


class DataParallel(torch.nn.Module):
    def __init__(self, module: torch.nn.Module, device_ids: Optional[Sequence] = None, output_device=None):
        super().__init__()
        self.module = module
        if device_ids is None:
            device_ids = [torch.cuda.current_device()] if torch.cuda.is_available() else []
        self.device_ids = [d.index if isinstance(d, torch.device) else int(d) for d in device_ids]
        self.output_device = output_device if output_device is not None else (self.device_ids[0] if self.device_ids else None)
        self.src_device = torch.device("cuda:%d" % self.device_ids[0]) if (self.device_ids and torch.cuda.is_available()) \
            else torch.device("cpu")   # (the reference allows a CPU-only run, data_parallel.py:39)
        self._bucket: Optional[GradBucket] = None

    @staticmethod
    def _world():
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def shard_of(self, data_list):
        """The element of `data_list` this process runs: the only one, or the one of its rank.  A list shorter than
        the world (the reference's `Collater` drops empty devices, `tg/dataloader.py:29-33`) leaves the last ranks
        without work: None."""
        rank, world = self._world()
        if world == 1:
            return data_list[0]   # the reference's one-device fallback runs the first element only (:48-50)
        if len(data_list) > world:
            raise ValueError("DataParallel got %d batches for %d processes: collate with collate_sharded(graphs, %d)"
                             % (len(data_list), world, world))
        return data_list[rank] if rank < len(data_list) else None

    def forward(self, data_list):
        if len(data_list) == 0:
            warnings.warn("DataParallel received an empty data list, which may result in unexpected behaviour.")
            return None
        data = self.shard_of(data_list)
        if data is None:
            return None
        return self.module(data.to(self.src_device))

    def reduce_gradients(self, local_count: Optional[int] = None, group=None) -> None:
        """Average the replicas' gradients of their local mean losses into the gradient of the mean over the global
        batch (`main_pyg.py:55-60` computes the loss on the gathered predictions): ONE all-reduce, weighted by the
        graphs each rank ran (0 for a rank without a shard).  The first call moves the gradients into a flat bucket."""
        if self._bucket is None:
            grads = {id(p): p.grad for p in self.module.parameters() if p.requires_grad and p.grad is not None}
            self._bucket = GradBucket(self.module.parameters())
            for p in self._bucket.params:   # keep what backward() already produced
                g = grads.get(id(p))
                if g is not None:
                    p.grad.copy_(g)
        self._bucket.all_reduce_mean(local_count, group)

    def zero_grad(self, set_to_none: bool = False) -> None:   # the bucket's views must survive
        if self._bucket is not None:
            self._bucket.zero()
        else:
            super().zero_grad(set_to_none=set_to_none)
