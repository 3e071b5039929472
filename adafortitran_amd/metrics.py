"""Channel-estimation MSE metric, as the reference's evaluator defines it
(reference src/utils.py:164-180 ``concat_complex_channel`` + ``nn.MSELoss`` and the
``2 * loss * B`` accumulation of src/main/trainer.py:338-347, ``to_db`` utils.py:233-245):

    MSE = mean over all complex grid elements of |h_est - h_ref|^2 ,   dB = 10 log10(MSE)

:class:`MseAccumulator` keeps the running sum on the device (one HIP reduction kernel per batch,
no ``.item()`` sync per batch as the reference does) and closes a sweep with ONE all-gather of the
per-rank ``(sum|e|^2, n_elements)`` pair -- RCCL over xGMI on GPUs, gloo on CPU.
"""
from __future__ import annotations

import math
from typing import Optional

import torch


def to_db(val: float) -> float:
    return 10.0 * math.log10(val)


class MseAccumulator:
    def __init__(self, device) -> None:
        self.device = torch.device(device)
        self.sum_sq = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.n_elements = 0

    def update(self, est: torch.Tensor, ref: torch.Tensor) -> None:
        """Add one batch (complex64 [B,S,T] each); ``ref`` is moved to the estimate's device
        (the reference forgets to)."""
        ref = ref.to(est.device)
        if est.device.type == "cuda":
            from .hip_ops import mse_sum   # hand-written reduction; raises if the extension is missing
            mse_sum(est, ref, self.sum_sq)
        else:
            d = torch.view_as_real(est).double() - torch.view_as_real(ref).double()
            self.sum_sq += (d * d).sum()
        self.n_elements += est.numel()

    def local_pair(self) -> torch.Tensor:
        return torch.stack([self.sum_sq[0], torch.tensor(float(self.n_elements), dtype=torch.float64,
                                                          device=self.device)])

    def result(self, group: Optional["torch.distributed.ProcessGroup"] = None) -> float:
        """Global MSE over all ranks: all-gather the 16-byte pairs, reduce locally (identical on
        every rank; equals the reference's sample-weighted mean up to summation order)."""
        pair = self.local_pair()
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():   # world_size 1 included: same code path on every launch
            gathered = [torch.empty_like(pair) for _ in range(dist.get_world_size(group))]
            dist.all_gather(gathered, pair, group=group)
            pair = torch.stack(gathered).sum(dim=0)
        return float(pair[0] / pair[1])

    def result_db(self, group=None) -> float:
        return to_db(self.result(group))


def shard_bounds(n_frames: int, world_size: int, rank: int):
    """Contiguous frame shard of rank r: [r*n/R, (r+1)*n/R) (SURVEY.md 8e partitioning)."""
    lo = (n_frames * rank) // world_size
    hi = (n_frames * (rank + 1)) // world_size
    return lo, hi
