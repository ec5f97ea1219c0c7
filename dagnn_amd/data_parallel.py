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

from .train import GradBucket, clip_flat_grad_norm_


class DataParallel(torch.nn.Module):
    def __init__(self, module: torch.nn.Module, device_ids: Optional[Sequence] = None, output_device=None,
                 sync_check: bool = True):
        """`sync_check` (evaluation mode only): True - every forward ends with the module's blocking `check()`, i.e. a device
        synchronisation per evaluation batch; the reference's loop reads the predictions back right away
        (`main_pyg.py:91-124`), so nothing is lost there.  False - the forward stays asynchronous (failures of EARLIER
        passes still surface at the next forward, from the arenas' pinned rings); call `model.module.check()` once behind
        the last batch of the loop."""
        super().__init__()
        self.module = module
        self.sync_check = bool(sync_check)
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

    def _device(self) -> torch.device:
        """Where this process runs its shard: with one process per GPU that is the device the module's parameters
        live on (NOT `device_ids[0]`: the reference-style `DataParallel(model, list(range(k)))` under k processes would
        send every rank's data to cuda:0), else the reference's `src_device`."""
        if self._world()[1] > 1:
            for p in self.module.parameters():
                return p.device
        return self.src_device

    def shard_of(self, data_list):
        """The element of `data_list` this process runs under `torch.distributed`: the one of its rank.  A list shorter
        than the world (the reference's `Collater` drops empty devices, `tg/dataloader.py:29-33`) leaves the last ranks
        without work: None."""
        rank, world = self._world()
        if world == 1:
            return data_list[0]
        if len(data_list) > world:
            raise ValueError("DataParallel got %d batches for %d processes: collate with collate_sharded(graphs, %d)"
                             % (len(data_list), world, world))
        return data_list[rank] if rank < len(data_list) else None

    @staticmethod
    def _gather(outs):
        """`torch.nn.DataParallel.gather` for the shapes the DAGNN heads return: tensors are concatenated along dim 0,
        lists / tuples element-wise (`tg/data_parallel.py:62`)."""
        first = outs[0]
        if isinstance(first, torch.Tensor):
            return torch.cat(outs, dim=0)
        if isinstance(first, (list, tuple)):
            return type(first)(DataParallel._gather([o[k] for o in outs]) for k in range(len(first)))
        raise TypeError("cannot gather outputs of type %s" % type(first).__name__)

    def forward(self, data_list):
        if len(data_list) == 0:
            warnings.warn("DataParallel received an empty data list, which may result in unexpected behaviour.")
            return None
        rank, world = self._world()
        dev = self._device()
        if world == 1 and len(self.device_ids) > 1 and len(data_list) > 1:
            # the reference's k-device branch (`:52-62`: scatter the first min(k, len) elements, one replica each,
            # gather on the output device) inside ONE process that owns one GPU: the same elements one after the other
            # on that GPU, outputs concatenated in list order - nothing is dropped
            outs = [self.module(d.to(dev)) for d in data_list[:len(self.device_ids)]]
            out = self._gather(outs)
        else:
            data = self.shard_of(data_list)   # one device: the reference's fallback runs the first element only (:48-50)
            if data is None:
                return None
            out = self.module(data.to(dev))
        if not self.training and self.sync_check:
            # evaluation consumes the outputs right away (`main_pyg.py:91-124`): surface a device-side failure of THIS
            # pass now instead of at the next forward (a loop's last batch has no next forward)
            check = getattr(self.module, "check", None)
            if callable(check):
                check()
        return out

    def _attach_bucket(self) -> None:
        """(Re-)bind every parameter's `.grad` to its view of the flat bucket.  `optimizer.zero_grad()` (the
        reference's loop, `main_pyg.py:50`) sets the gradients to None, so the next `backward()` allocates fresh
        tensors outside the bucket: whatever `.grad` holds then is copied into the view and re-bound, so that the
        collective below reduces THIS step's gradients and `optimizer.step()` reads the reduced ones."""
        if self._bucket is None:
            grads = {id(p): p.grad for p in self.module.parameters() if p.requires_grad and p.grad is not None}
            self._bucket = GradBucket(self.module.parameters())
            for p in self._bucket.params:   # keep what backward() already produced
                g = grads.get(id(p))
                if g is not None:
                    p.grad.copy_(g)
            return
        self._bucket.rebind()

    def reduce_gradients(self, local_count: Optional[int] = None, group=None) -> None:
        """Average the replicas' gradients of their local mean losses into the gradient of the mean over the global
        batch (`main_pyg.py:55-60` computes the loss on the gathered predictions): ONE all-reduce, weighted by the
        graphs each rank ran (0 for a rank without a shard).  The first call moves the gradients into a flat bucket;
        every call checks that they still live there (see `_attach_bucket`)."""
        self._attach_bucket()
        self._bucket.all_reduce_mean(local_count, group)

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """The reference's `torch.nn.utils.clip_grad_norm(model.parameters(), args.clip)` (`main_pyg.py:63-64`): call it
        AFTER `reduce_gradients` (the norm is the global batch's, the same on every rank) and before `optimizer.step()`.
        With a bucket: one norm over the flat buffer; without (single process, no `reduce_gradients`): torch's own."""
        if self._bucket is not None:
            self._bucket.rebind()
            return clip_flat_grad_norm_([self._bucket.flat], max_norm)
        return torch.nn.utils.clip_grad_norm_(self.module.parameters(), max_norm)

    def zero_grad(self, set_to_none: bool = False) -> None:   # keeps the bucket's views (no copy on the next step)
        if self._bucket is not None:
            self._bucket.zero()
        else:
            super().zero_grad(set_to_none=set_to_none)
