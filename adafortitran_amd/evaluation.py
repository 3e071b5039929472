"""Evaluation sweep without per-batch host syncs.

Mirrors ``ModelEvaluator._evaluate_dataloader`` / ``get_test_stats`` of the reference
(src/main/trainer.py:311-347): same dispatch on the estimator class, same sample-weighted mean of
``2 * MSELoss(cat(Re, Im))`` per loader, same ``{int(val): dB}`` dictionary sorted by the integer in
the loader's name -- but the squared error accumulates on the device (metrics.MseAccumulator) and is
read back once per loader (plus one all-gather when several ranks each evaluate a shard).

With an ``ingest.PackedLoader`` on a HIP device the reference's pilot-count ``ValueError`` ("Expected 24 pilot values, got 25",
dataset.py:128-132) surfaces ONE BATCH LATE -- while the next batch is prepared, at the end of the sweep, or when the iterator is
closed early -- because the counts come back asynchronously; the sweep therefore never returns a value computed from a bad frame
without raising, but the exception's traceback points at the loader, not at the forward of the offending batch."""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .estimators import AdaFortiTranEstimator
from .metrics import MseAccumulator, to_db


def forward_pass(model: torch.nn.Module, pilots: torch.Tensor, meta_data: Optional[tuple]) -> torch.Tensor:
    """The reference's ``_forward_pass`` (trainer.py:267-288): AdaFortiTran needs ``meta_data``."""
    if isinstance(model, AdaFortiTranEstimator):
        if meta_data is None:
            raise ValueError("AdaFortiTranEstimator requires meta_data but it was not provided")
        return model(pilots, meta_data)
    return model(pilots)


def evaluate_dataloader(model: torch.nn.Module, dataloader: Iterable, group=None) -> float:
    """Mean |h_est - h|^2 over every complex element of the loader (one host sync at the end)."""
    model.eval()
    device = next(model.parameters()).device
    acc = MseAccumulator(device)
    with torch.no_grad():
        for pilots, ideal, meta in dataloader:
            acc.update(forward_pass(model, pilots, meta), ideal)
    return acc.result(group)


def get_test_stats(model: torch.nn.Module, test_dataloaders: List[Tuple[str, Iterable]], group=None) -> Dict[int, float]:
    stats: Dict[int, float] = {}
    for name, loader in sorted(test_dataloaders, key=lambda x: int(x[0].split("_")[1])):
        stats[int(name.split("_")[1])] = to_db(evaluate_dataloader(model, loader, group))
    return stats
