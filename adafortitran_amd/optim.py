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

Every rank must take the same branch around the collectives.  Two ways to drive a step:
  * plain ``loss.backward(); optimizer.step()``: the reduction happens inside ``step()``; gradient
    clipping is the optimizer's ``max_grad_norm`` (applied AFTER the reduction, on the averaged gradient);
  * the reference's GradScaler branch (trainer.py:207-223: ``scaler.unscale_``, ``clip_grad_norm_``,
    ``scaler.step``): with more than one rank call ``optimizer.reduce_gradients()`` right after
    ``backward()``.  It all-reduces (averages) the flat gradient buffer in place, so the inf check, the
    unscale and the clip all see the same averaged gradient on every rank and ``found_inf`` agrees across
    ranks; ``step()`` then skips its own reduction.  Driving ``step()`` through a GradScaler with
    unreduced gradients on more than one rank raises instead of risking mismatched collectives.

On the HIP device the update is ``aft_adam_step_f32`` (one kernel over the shard).  On CPU tensors the
same formula runs through torch ops, which is what the world_size-2 gloo tests exercise.
"""
from __future__ import annotations

import itertools
import weakref
from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist


_TOKENS = itertools.count(1)
_OWNERS: "weakref.WeakValueDictionary[int, FlatParameters]" = weakref.WeakValueDictionary()


def flat_owner(p: torch.nn.Parameter) -> Optional["FlatParameters"]:
    """The live FlatParameters that tagged ``p`` (None when there is none, or it was released / collected; a token that
    arrived through pickle from another process matches nothing here)."""
    tok = getattr(p, "_aft_flat_owner", None)
    return _OWNERS.get(tok) if isinstance(tok, int) else None


class FlatParameters:
    """Re-home ``params`` as views into one contiguous float32 buffer (and their grads into another).

    ``direct_accumulation``: tag the parameters so that the library's backward kernels add straight into
    the flat ``.grad`` views (training.direct_grad_ok).  The tag is a weak reference to this object: it
    stops applying when this object is released or ``release()`` is called."""

    def __init__(self, params: Iterable[torch.nn.Parameter], pad_to: int = 1, direct_accumulation: bool = False) -> None:
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
        self.direct_accumulation = bool(direct_accumulation)
        # the tag on a parameter is a plain integer token (picklable: Parameter.__reduce_ex__ pickles __dict__, and a
        # weakref there made torch.save(model) / mp.spawn / DataLoader workers fail while an optimizer was alive);
        # the token -> owner map lives on this side and holds the owner weakly
        self.token = next(_TOKENS)
        _OWNERS[self.token] = self
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.data[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            p._aft_flat_owner, p._aft_flat_off = self.token, off

    def owns_grad(self, p: torch.nn.Parameter) -> bool:
        g = p.grad
        return (g is not None and g.dtype == torch.float32 and g.is_contiguous()
                and g.data_ptr() == self.grad.data_ptr() + 4 * getattr(p, "_aft_flat_off", -1))

    def release(self) -> None:
        """Drop the tags (parameters and grads stay where they are)."""
        for p in self.params:
            if getattr(p, "_aft_flat_owner", None) == self.token:
                p._aft_flat_owner = None
        _OWNERS.pop(self.token, None)

    def zero_grad(self) -> None:
        """Keep the grad views alive (set_to_none would detach them from the flat buffer)."""
        self.grad.zero_()
        base = self.grad.data_ptr()     # (one pointer compare per parameter: this runs every step, on the host's critical path at small batches)
        for p, off in zip(self.params, self.offsets):
            g = p.grad
            if g is None or g.data_ptr() != base + self.grad.element_size() * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)


class ShardedFlatAdam(torch.optim.Optimizer):
    """Adam (torch.optim.Adam semantics, amsgrad off) over a FlatParameters, sharded across
    ``process_group``.  ``step()`` = reduce-scatter grads (mean) -> optional global-norm clipping ->
    update own shard -> all-gather params.  A ``torch.optim.Optimizer``, so the reference's
    ``ExponentialLR(optimizer, gamma=0.995)`` (trainer.py:414) drives ``param_groups[0]["lr"]`` as usual.

    ``max_grad_norm`` replaces the trainer's ``clip_grad_norm_`` call (trainer.py:222-223): with sharded
    gradients the norm has to be taken after the reduction, on the averaged gradient."""

    _step_supports_amp_scaling = True   # GradScaler hands step() its found_inf / grad_scale instead of deciding itself

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.0, max_grad_norm: Optional[float] = None,
                 process_group: Optional[dist.ProcessGroup] = None, direct_accumulation: Optional[bool] = None) -> None:
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized()   # collectives run whenever a group exists,
        self.world = dist.get_world_size(process_group) if self.distributed else 1   # world_size 1 included
        self.rank = dist.get_rank(process_group) if self.distributed else 0
        on_hip = params[0].device.type == "cuda"
        self.flat = FlatParameters(params, pad_to=self.world * 64,
                                   direct_accumulation=on_hip if direct_accumulation is None else direct_accumulation)
        self.shard = self.flat.padded // self.world
        lo = self.rank * self.shard
        self.lo = lo
        self.p_shard = self.flat.data[lo:lo + self.shard]
        self.g_local = self.flat.grad[lo:lo + self.shard]         # this rank's slice of the (local or all-reduced) gradients
        self.g_shard = torch.zeros(self.shard, dtype=torch.float32, device=self.flat.data.device) if self.distributed \
            else self.g_local                                     # reduce-scatter output
        self.exp_avg = torch.zeros_like(self.p_shard)
        self.exp_avg_sq = torch.zeros_like(self.p_shard)
        self.max_grad_norm = max_grad_norm
        self.steps = 0
        self._grads_reduced = False

    def zero_grad(self, set_to_none: bool = False) -> None:   # the views must stay attached to the flat buffer
        self.flat.zero_grad()
        self._grads_reduced = False

    @torch.no_grad()
    def reduce_gradients(self) -> None:
        """Average the flat gradient buffer over the ranks IN PLACE (one all-reduce), like DDP leaves ``.grad``:
        call it ONCE per step, after the LAST ``backward()`` (gradient accumulation included) and before
        ``scaler.unscale_`` / ``clip_grad_norm_`` / any inspection of gradients when more than one rank trains (module
        docstring).  ``step()`` then skips its reduction.  A second call before ``step()`` / ``zero_grad()`` raises."""
        if self._grads_reduced:
            # a second call would average a buffer that further backward() calls may have added rank-local gradients to
            # (gradient accumulation): step() would then update with a mix of averaged and local gradients, silently
            raise RuntimeError("reduce_gradients() was already called for this step: call it ONCE, after the last "
                               "backward() of the step (gradient accumulation: accumulate first, reduce once)")
        if self.distributed:
            dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            if self.world > 1:
                self.flat.grad.mul_(1.0 / self.world)
        self._grads_reduced = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        found_inf = getattr(self, "found_inf", None)     # set by torch.amp.GradScaler.step around this call
        grad_scale = getattr(self, "grad_scale", None)   # None when scaler.unscale_ already ran (the reference's order)
        if found_inf is not None:
            if self.world > 1 and not self._grads_reduced:
                raise RuntimeError("ShardedFlatAdam.step() driven through a GradScaler on more than one rank: call "
                                   "optimizer.reduce_gradients() after backward() so that every rank checks the same "
                                   "(averaged) gradients for inf/nan and takes the same branch")
            if float(found_inf) > 0:                     # the scaler skips the step; identical on every rank
                self._grads_reduced = False
                return loss
        self.steps += 1
        scale = 1.0 if grad_scale is None else 1.0 / float(grad_scale)
        if self._grads_reduced:
            g = self.g_local                             # already the mean over ranks
        else:
            if self.distributed:
                dist.reduce_scatter_tensor(self.g_shard, self.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            g = self.g_shard
            scale /= self.world
        if self.max_grad_norm is not None:
            sq = (g.double() ** 2).sum() * scale * scale
            if self.distributed:
                dist.all_reduce(sq, group=self.group)
            norm = float(sq.sqrt())
            scale *= min(1.0, self.max_grad_norm / (norm + 1e-6))   # clip_grad_norm_'s coefficient
        self._adam(g, scale)
        if self.distributed:
            dist.all_gather_into_tensor(self.flat.data, self.p_shard.clone(), group=self.group)
        # the fused kernel and the all-gather write the flat buffer behind the parameters' backs: move their autograd
        # version counters, as an in-place torch op would (saved-tensor checks, the HipEngine's packed-weight cache)
        try:
            torch.autograd.graph.increment_version(self.flat.params)
        except (AttributeError, TypeError):
            for p in self.flat.params:
                torch._C._increment_version(p)
        self._grads_reduced = False
        return loss

    def _adam(self, g_shard: torch.Tensor, grad_scale: float) -> None:
        g0 = self.param_groups[0]
        lr, (b1, b2), eps, wd = g0["lr"], g0["betas"], g0["eps"], g0["weight_decay"]
        if self.p_shard.device.type == "cuda":
            from . import _lib
            lib = _lib.load()
            _lib.check(lib.aft_adam_step_f32(self.p_shard.data_ptr(), g_shard.data_ptr(), self.exp_avg.data_ptr(),
                                             self.exp_avg_sq.data_ptr(), self.shard, lr, b1, b2, eps, wd, grad_scale,
                                             self.steps, _lib.current_stream_ptr(self.p_shard.device)))
            return
        g = g_shard * grad_scale + wd * self.p_shard
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.steps, 1 - b2 ** self.steps
        self.p_shard.addcdiv_(self.exp_avg, self.exp_avg_sq.sqrt() / bc2 ** 0.5 + eps, value=-lr / bc1)

    # ---- checkpoints (reference trainer.py:663-674 saves optimizer.state_dict() from one process) ----
    def _full_moments(self):
        if not self.distributed or self.world == 1:
            return self.exp_avg, self.exp_avg_sq
        full = [torch.empty(self.flat.padded, dtype=torch.float32, device=self.exp_avg.device) for _ in range(2)]
        dist.all_gather_into_tensor(full[0], self.exp_avg, group=self.group)
        dist.all_gather_into_tensor(full[1], self.exp_avg_sq, group=self.group)
        return full

    def state_dict(self):
        """``torch.optim.Adam``-format state (``state[i] = {step, exp_avg, exp_avg_sq}`` per parameter in
        ``param_groups[0]["params"]`` order + ``param_groups``), holding the moments of ALL ranks.  WITH MORE THAN ONE
        RANK THIS IS A COLLECTIVE: every rank must call it (each gets the full state; rank 0 writes the file).  The
        reference's call pattern -- only the process that saves calls ``optimizer.state_dict()``
        (``_save_checkpoint``, trainer.py:663-674) -- would hang in the all-gather; call it on all ranks and save on one.  Loads into ``torch.optim.Adam`` and
        into a ShardedFlatAdam of any world size."""
        m, v = self._full_moments()
        state = {}
        for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.steps)), "exp_avg": m[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": v[off:off + n].view(p.shape).clone()}
        g0 = self.param_groups[0]
        group = {k: val for k, val in g0.items() if k != "params"}
        group.update(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     decoupled_weight_decay=False)
        group["params"] = list(range(len(self.flat.params)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd) -> None:
        """Accepts the format above (also a ``torch.optim.Adam`` state_dict over the same parameters): the moments
        are laid into the flat order and this rank keeps its shard, whatever world size wrote the file."""
        state, groups = sd["state"], sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.params):
            raise ValueError("optimizer state does not match this model's parameter list")
        dev = self.exp_avg.device
        m = torch.zeros(self.flat.padded, dtype=torch.float32, device=dev)
        v = torch.zeros_like(m)
        steps = 0
        for i, (p, off) in enumerate(zip(self.flat.params, self.flat.offsets)):
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
            n = p.numel()
            m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps = max(steps, int(float(st["step"])))
        self.exp_avg.copy_(m[self.lo:self.lo + self.shard])
        self.exp_avg_sq.copy_(v[self.lo:self.lo + self.shard])
        self.steps = steps
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in groups[0]:
                self.param_groups[0][k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
