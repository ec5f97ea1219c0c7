"""Training-side plumbing for one-process-per-GPU data parallelism.

The reference trains with `tg/data_parallel.DataParallel`: one process, k threads, the 29.8 M parameters
re-broadcast and the gradients reduced to device 0 every step (`ogbg-code/tg/data_parallel.py:48-62`).  Here
every rank owns a full replica and its own shard of graphs; the only exchange of a training step is ONE
all-reduce of a flat gradient bucket over RCCL/xGMI.  The reference's loss is the mean over the graphs of the GLOBAL
batch (`main_pyg.py:55-60` on the gathered predictions) and its `Collater` balances shards by nodes, not by graphs
(`tg/dataloader.py:17-27`: 12..18 graphs per shard on the seed-0 batch), so every rank's gradient of its LOCAL mean
loss is weighted by its graph count: grad = sum_k b_k grad_k / sum_k b_k.  The counts travel in the same collective.
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
        n = sum(p.numel() for p in self.params)
        self._buf = torch.zeros(n + 1, dtype=torch.float32, device=dev)   # one more float: this rank's graph count
        self.flat = self._buf[:n]
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket needs fp32 parameters on one device")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    @property
    def numel(self) -> int:
        return self.flat.numel()

    def rebind(self) -> None:
        """Make every `p.grad` alias its view of the flat buffer again.  A gradient that lives elsewhere (the
        optimizer's `zero_grad()` set it to None and `backward()` allocated a fresh one) is copied in first; a
        parameter without a gradient this step contributes zeros."""
        off = 0
        for p in self.params:
            view = self.flat[off:off + p.numel()].view_as(p)
            g = p.grad
            if g is None:
                view.zero_()
                p.grad = view
            elif g.data_ptr() != view.data_ptr() or g.shape != view.shape:
                view.copy_(g)
                p.grad = view
            off += p.numel()

    def zero(self) -> None:
        """Replaces `optimizer.zero_grad()` (which would detach the views when it sets grads to None)."""
        self.flat.zero_()

    def all_reduce_mean(self, local_count: Optional[int] = None, group: Optional[dist.ProcessGroup] = None) -> None:
        """Gradient of the mean loss over the GLOBAL batch from the ranks' gradients of their local mean losses:
        sum_k b_k grad_k / sum_k b_k with b_k = `local_count` (graphs in this rank's shard; 0 for a rank that skipped
        its shard, as `main_pyg.py:47` does for one-graph chunks).  Without `local_count` every rank weighs the same
        (equal shards).  ONE collective of numel + 1 floats."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
            return
        self.launch(1.0 if local_count is None else float(local_count), group).wait()
        self.normalise()

    def launch(self, count: float, group: Optional[dist.ProcessGroup] = None):
        """Start the count-weighted sum of this bucket (asynchronous: returns the collective's work handle; the
        backend orders it behind the kernels already queued on the current stream)."""
        if count != 1.0:
            self.flat.mul_(count)
        self._buf[-1] = count
        return dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def normalise(self) -> None:
        """After the collective: divide by the total count it carried."""
        self.flat.div_(self._buf[-1].clamp(min=1.0))


def clip_flat_grad_norm_(flats, max_norm: float, eps: float = 1e-6) -> torch.Tensor:
    """`torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)` (`ogbg-code/main_pyg.py:63-64`, `--clip`, 0.25 in
    `scripts/ogb_tok.sh:16`) over flat gradient buffers: the global 2-norm of ALL gradients, every buffer scaled by
    `min(1, max_norm / (norm + eps))`.  One norm kernel per buffer instead of one per parameter, no device->host read
    (the scale stays on the device).  Under data parallelism it must see the REDUCED gradients - the mean over the
    global batch, identical on every rank - so it sits behind the last collective of the step and in front of
    `optimizer.step()`; every rank then applies the same scale without another exchange.  Returns the norm."""
    flats = [f for f in flats if f is not None and f.numel() > 0]
    if not flats:
        return torch.zeros(())
    sq = torch.stack([torch.linalg.vector_norm(f, 2.0) for f in flats])
    total = torch.linalg.vector_norm(sq, 2.0)
    coef = torch.clamp(float(max_norm) / (total + eps), max=1.0)
    for f in flats:
        f.mul_(coef)
    return total


class OverlappedGradReducer(object):
    """Gradient exchange of one training step in TWO buckets so that most of it hides behind the reverse sweep.

    The 29.8 M parameters of the headline model are 25.6 M in the vocabulary heads (`graph_pred_linear_list`,
    `dagnn.py:106-112`) and 4.2 M in the DAGNN core + encoder.  `loss.backward()` finishes the heads first - their
    gradients are final before the recurrence's reverse sweep (the longest kernel of the step) even starts - so the
    heads' bucket (113 MB at cfg 2) is all-reduced ASYNCHRONOUSLY from a post-accumulate hook on the last head
    parameter, on the communication backend's own stream, while the sweep and its epilogue run; only the small core
    bucket (6.3 MB) is exchanged after `backward()` returns.  Same arithmetic as `GradBucket.all_reduce_mean`
    (count-weighted sum / total count), bucket by bucket: the result is identical to the single-bucket exchange.

        red = OverlappedGradReducer(model.parameters(), early=model.graph_pred_linear_list.parameters())
        red.zero(local_count=b_k); loss.backward(); red.finish(); optimizer.step()
    """

    def __init__(self, params: Iterable[torch.nn.Parameter], early: Iterable[torch.nn.Parameter],
                 group: Optional[dist.ProcessGroup] = None):
        early = [p for p in early if p.requires_grad]
        ids = {id(p) for p in early}
        rest = [p for p in params if p.requires_grad and id(p) not in ids]
        self.group = group
        from . import engine
        engine.register_collective(group)   # (the co-residency rule of the persistent kernels looks at THIS group: engine.reserved_cus)
        self.early = GradBucket(early) if early else None
        self.late = GradBucket(rest) if rest else None
        self._count = 1.0
        self._pending = 0
        self._work = None
        self.exposed_ms = None
        self._hooks = []
        if self.early is not None:
            for p in self.early.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_early_grad))

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def zero(self, local_count: Optional[int] = None) -> None:
        """Start of a step.  `local_count` (graphs of this rank's shard) must be known here: the early bucket leaves
        in the middle of `backward()`."""
        for b in (self.early, self.late):
            if b is not None:
                b.zero()
        self._count = 1.0 if local_count is None else float(local_count)
        self._pending = len(self.early.params) if self.early is not None else 0
        self._work = None

    def _on_early_grad(self, p) -> None:
        self._pending -= 1
        if self._pending == 0 and self._active():
            self.early.rebind()   # (a gradient autograd allocated outside the bucket is copied in first)
            self._work = self.early.launch(self._count, self.group)

    def finish(self, clip: float = 0.0) -> Optional[torch.Tensor]:
        """After `backward()`: exchange the late bucket, wait for the early one, normalise both - and, with `clip` > 0, the
        reference's `clip_grad_norm(model.parameters(), clip)` (`main_pyg.py:63-64`) on the reduced gradients of BOTH
        buckets (it needs the global norm, so it cannot start before the last collective has landed); then
        `optimizer.step()`.  Returns the gradient norm when clipping."""
        self._exchange()
        if clip and clip > 0:
            for b in (self.early, self.late):   # (single process: the gradients may live outside the buckets)
                if b is not None and not self._active():
                    b.rebind()
            return clip_flat_grad_norm_([b.flat for b in (self.early, self.late) if b is not None], clip)
        return None

    def _exchange(self) -> None:
        if not self._active():
            return
        if self.early is not None and self._work is None:   # no hook fired (a head without gradient): exchange it now
            self.early.rebind()
            self._work = self.early.launch(self._count, self.group)
        if self.late is not None:
            self.late.rebind()
            w = self.late.launch(self._count, self.group)
            w.wait()
            self.late.normalise()
        if self._work is not None:
            self._work.wait()
            self.early.normalise()
            self._work = None

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


# ----------------------------------------------------------------------------- the TOK task's loss (main_pyg.py:55-60)
_CE_COUNTERS = {}


class _SeqCE(torch.autograd.Function):
    """Mean cross-entropy over the S heads' logits laid side by side in ONE [B, S * V] tensor, loss and gradient in one HIP
    launch (`dagnn_seq_ce`, csrc/loss.hip); the backward is one multiplication by the incoming scalar."""

    @staticmethod
    def forward(ctx, base, y, S, V):
        from . import engine
        lib = engine._lib.load()
        B = base.shape[0]
        dev = base.device
        need_grad = base.requires_grad
        # (d logits with rows pitched to a multiple of 4 floats: what the heads' weight-gradient product reads 16 bytes at a time)
        dl = torch.empty(B, (S * V + 3) // 4 * 4, dtype=torch.float32, device=dev)[:, :S * V] if need_grad else None
        row = torch.empty(B * S, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        key = (dev.index, engine._stream(base))
        cnt = _CE_COUNTERS.get(key)
        if cnt is None:
            cnt = _CE_COUNTERS[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        engine.check(lib.dagnn_seq_ce(base.data_ptr(), base.stride(0), y.data_ptr(), B, S, V, None if dl is None else dl.data_ptr(),
                                      0 if dl is None else dl.stride(0), row.data_ptr(), loss.data_ptr(), cnt.data_ptr(),
                                      engine._stream(base)), "dagnn_seq_ce")
        ctx.dl = dl
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dl, ctx.dl = ctx.dl, None
        return dl.mul_(g), None, None, None


def seq_cross_entropy(pred_list, y_arr: torch.Tensor) -> torch.Tensor:
    """`sum_i CrossEntropyLoss()(pred_list[i], y_arr[:, i]) / len(pred_list)` - the loss of the reference's training loop
    (ogbg-code/main_pyg.py:55-60).  When the list is what `DAGNN.forward` returns on a GPU - views of one [B, S * V] tensor, the
    heads' outputs side by side - loss and gradient are ONE launch (`dagnn_seq_ce`) and the backward pass meets the heads as
    one matrix; any other list takes the plain loop.  Targets must lie in [0, V) (no `ignore_index`)."""
    S = len(pred_list)
    p0 = pred_list[0]
    base = getattr(p0, "_base", None)
    fused = base is not None and base.is_cuda and base.dtype == torch.float32 and base.dim() == 2 and base.stride(1) == 1 \
        and y_arr.is_cuda and y_arr.dtype == torch.int64 and y_arr.dim() == 2 and y_arr.shape[1] == S and y_arr.is_contiguous()
    if fused:
        V = p0.shape[1]
        for i, p in enumerate(pred_list):
            if getattr(p, "_base", None) is not base or p.shape != p0.shape or p.stride() != base.stride() or \
                    p.data_ptr() != base.data_ptr() + 4 * i * V:
                fused = False
                break
        fused = fused and base.shape[1] == S * V and y_arr.shape[0] == base.shape[0]
    if not fused:
        loss = 0
        for i in range(S):
            loss = loss + torch.nn.functional.cross_entropy(pred_list[i].to(torch.float32), y_arr[:, i])
        return loss / S
    return _SeqCE.apply(base, y_arr, S, V)


# ----------------------------------------------------------------------------- clip_grad_norm_ + Adam (main_pyg.py:63-65)
class ClipAdam(torch.optim.Optimizer):
    """`torch.nn.utils.clip_grad_norm_(params, max_norm)` followed by `torch.optim.Adam(params, ...).step()` - the tail of the
    reference's training step (ogbg-code/main_pyg.py:63-65, optimizer at :179) - as three launches (csrc/optim.hip): the global
    gradient 2-norm as a device float (fixed summation order), then Adam's update with the clip coefficient applied to the
    gradient as it is read.  Every tensor crosses the memory bus once; torch's pair is ~12 launches that move the gradients
    three times and walk the small tensors at a fraction of the memory rate (0.36 -> 0.2 ms at the headline shape).

    Same update as `torch.optim.Adam` (no amsgrad, no maximize; `weight_decay` is Adam's L2 term) and the same `state_dict`
    layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so checkpoints move between the two.  `max_norm=None` (or <= 0)
    leaves the gradients unclipped.  Gradients themselves are NOT scaled in place (unlike `clip_grad_norm_`): the coefficient
    only enters the update.  `last_norm` is the device float of the last step's total norm (what `clip_grad_norm_` returns).
    fp32 parameters on one ROCm device."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 max_norm: Optional[float] = None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.max_norm = max_norm
        self.last_norm = None
        self._scratch = None

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes as C
        from . import engine
        from ._lib import OptTensor
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = engine._lib.load()
        work = []   # (group, [params with a gradient])
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if ps:
                work.append((group, ps))
        if not work:
            return loss
        allp = [p for _, ps in work for p in ps]
        dev = allp[0].device
        for p in allp:
            if not p.is_cuda or p.device != dev or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                raise engine.DagnnHipError("ClipAdam needs dense fp32 parameters and gradients on one ROCm device")
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in allp]
        st = engine._stream(allp[0])
        cap = 48
        clip = self.max_norm is not None and self.max_norm > 0
        if clip:
            numel = (C.c_int64 * len(allp))(*[g.numel() for g in grads])
            chunks = int(lib.dagnn_opt_chunks(numel, len(allp)))
            sc = self._scratch
            if sc is None or sc[0].device != dev or sc[0].numel() < chunks:
                sc = self._scratch = (torch.empty(max(chunks, 1), dtype=torch.float32, device=dev),
                                      torch.empty(2, dtype=torch.float32, device=dev))
            for o in range(0, len(grads), cap):
                part = grads[o:o + cap]
                ptrs = (C.c_void_p * len(part))(*[g.data_ptr() for g in part])
                nn_ = (C.c_int64 * len(part))(*[g.numel() for g in part])
                engine.check(lib.dagnn_grad_norm(ptrs, nn_, len(part), sc[0].data_ptr(), sc[0].numel(), sc[1].data_ptr(),
                                                 1 if o > 0 else 0, sc[1].data_ptr() + 4, st), "dagnn_grad_norm")
            self.last_norm = sc[1][1]
        k = 0
        for group, ps in work:
            b1, b2 = group["betas"]
            gs = grads[k:k + len(ps)]
            k += len(ps)
            for p in ps:
                state = self.state[p]
                if not state:
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            # (one step counter per group, as torch keeps them in lock-step; parameters that joined later start their own)
            steps = {}
            for p, g in zip(ps, gs):
                state = self.state[p]
                if state["step"].is_cuda:   # (a checkpoint written by torch's fused Adam keeps its counters on the device)
                    state["step"] = state["step"].cpu()
                state["step"] += 1
                steps.setdefault(int(state["step"]), []).append((p, g, state))
            for stepno, items in steps.items():
                for o in range(0, len(items), cap):
                    part = items[o:o + cap]
                    arr = (OptTensor * len(part))()
                    for j, (p, g, state) in enumerate(part):
                        m, v = state["exp_avg"], state["exp_avg_sq"]
                        if not (p.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                            raise engine.DagnnHipError("ClipAdam needs contiguous parameters")
                        arr[j].param, arr[j].grad, arr[j].exp_avg, arr[j].exp_avg_sq, arr[j].numel = \
                            p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                    engine.check(lib.dagnn_clip_adam(arr, len(part), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                     float(group["weight_decay"]), stepno, float(self.max_norm) if clip else 0.0,
                                                     self._scratch[1].data_ptr() + 4 if clip else None, st), "dagnn_clip_adam")
        return loss
