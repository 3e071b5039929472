"""Data-parallel training step pieces: flat parameter/gradient buffers, the
gradient exchange over RCCL/xGMI and a fused Adam update.

The reference trains on one device with ``torch.optim.Adam`` + ``ExponentialLR``
(``src/main/trainer.py:407-415``).  For N ranks (one process per GPU, ``torch.distributed`` backend
``nccl`` = RCCL) this module keeps every parameter as a view into ONE float32 buffer (3.95 MB for the
default model) and every ``.grad`` as a view into a second one, so a step is:

    reduce_scatter(grad buffer)  ->  Adam on this rank's 1/N shard  ->  all_gather(param buffer)

i.e. the all-reduce split into its two halves with the optimizer in between: the same bytes on the
wire as an all-reduce (each half moves (N-1)/N of the buffer per rank), 1/N of the optimizer work and
state per rank, and one collective per direction per step instead of one per parameter.  On xGMI
(point-to-point links, a ring is per-link bound) a single 4 MB message is what RCCL's direct
algorithms want; nothing here mirrors an NCCL call pattern of the reference (it has none).

On the HIP device the update is ``aft_adam_step_f32`` (one kernel over the shard).  On CPU tensors the
same formula runs through torch ops, which is what the world_size-2 gloo test exercises.
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist


class FlatParameters:
    """Re-home ``params`` as views into one contiguous float32 buffer (and their grads into another)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], pad_to: int = 1) -> None:
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if dt != torch.float32 or any(p.dtype != dt or p.device != dev for p in self.params):
            raise ValueError("FlatParameters needs float32 parameters on one device")
        # every tensor starts on a 64-byte boundary of the flat buffers (vector loads of the kernels)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 15) // 16 * 16
        self.numel = off
        self.padded = (self.numel + pad_to - 1) // pad_to * pad_to
        self.data = torch.zeros(self.padded, dtype=dt, device=dev)
        self.grad = torch.zeros(self.padded, dtype=dt, device=dev)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)

    def zero_grad(self) -> None:
        """Keep the grad views alive (set_to_none would detach them from the flat buffer)."""
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad[off:off + n].data_ptr():
                p.grad = self.grad[off:off + n].view(p.shape)


class ShardedFlatAdam(torch.optim.Optimizer):
    """Adam (torch.optim.Adam semantics, amsgrad off) over a FlatParameters, sharded across
    ``process_group``.  ``step()`` = reduce-scatter grads (mean) -> optional global-norm clipping ->
    update own shard -> all-gather params.  A ``torch.optim.Optimizer``, so the reference's
    ``ExponentialLR(optimizer, gamma=0.995)`` (trainer.py:414) drives ``param_groups[0]["lr"]`` as usual.

    ``max_grad_norm`` replaces the trainer's ``clip_grad_norm_`` call (trainer.py:222-223): with sharded
    gradients the norm has to be taken after the reduction, on the averaged gradient."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, max_grad_norm: Optional[float] = None,
                 process_group: Optional[dist.ProcessGroup] = None) -> None:
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if self.world > 1 else 0
        self.flat = FlatParameters(params, pad_to=self.world * 64)
        self.shard = self.flat.padded // self.world
        lo = self.rank * self.shard
        self.p_shard = self.flat.data[lo:lo + self.shard]
        self.g_shard = torch.zeros(self.shard, dtype=torch.float32, device=self.flat.data.device) if self.world > 1 \
            else self.flat.grad[lo:lo + self.shard]
        self.exp_avg = torch.zeros_like(self.p_shard)
        self.exp_avg_sq = torch.zeros_like(self.p_shard)
        self.max_grad_norm = max_grad_norm
        self.steps = 0
        if self.flat.data.device.type == "cuda":
            from . import training
            training.ACCUMULATE_INTO_GRAD = True   # the encoder backward adds into the flat grad views

    def zero_grad(self, set_to_none: bool = False) -> None:   # the views must stay attached to the flat buffer
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.steps += 1
        if self.world > 1:
            dist.reduce_scatter_tensor(self.g_shard, self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
        scale = 1.0 / self.world
        if self.max_grad_norm is not None:
            sq = (self.g_shard.double() ** 2).sum() * scale * scale
            if self.world > 1:
                dist.all_reduce(sq, group=self.group)
            norm = float(sq.sqrt())
            scale *= min(1.0, self.max_grad_norm / (norm + 1e-6))   # clip_grad_norm_'s coefficient
        self._adam(scale)
        if self.world > 1:
            dist.all_gather_into_tensor(self.flat.data, self.p_shard.clone(), group=self.group)
        return loss

    def _adam(self, grad_scale: float) -> None:
        g0 = self.param_groups[0]
        lr, (b1, b2), eps, wd = g0["lr"], g0["betas"], g0["eps"], g0["weight_decay"]
        if self.p_shard.device.type == "cuda":
            from . import _lib
            lib = _lib.load()
            _lib.check(lib.aft_adam_step_f32(self.p_shard.data_ptr(), self.g_shard.data_ptr(), self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), self.shard, lr, b1, b2, eps, wd, grad_scale,
                                             self.steps, _lib.current_stream_ptr(self.p_shard.device)))
            return
        g = self.g_shard * grad_scale + wd * self.p_shard
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.steps, 1 - b2 ** self.steps
        self.p_shard.addcdiv_(self.exp_avg, self.exp_avg_sq.sqrt() / bc2 ** 0.5 + eps, value=-lr / bc1)

    # checkpoints (reference trainer.py saves optimizer.state_dict()): this rank's shard of the moments
    def state_dict(self):
        return {"steps": self.steps, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "lr": self.param_groups[0]["lr"], "world": self.world, "rank": self.rank}

    def load_state_dict(self, sd) -> None:
        if sd["world"] != self.world or sd["rank"] != self.rank:
            raise ValueError("optimizer shard was saved for a different world size / rank")
        self.steps = int(sd["steps"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.param_groups[0]["lr"] = sd["lr"]
