"""Training-step harness around the hot path -- SURVEY.md section 8(f) rank 4.

Mirrors the reference's loop: ``optim.AdamW(lr, weight_decay)`` + ``ExponentialLR(gamma)`` + ``CrossEntropyLoss``
(generic_train.py:24-26), one ``zero_grad / forward / loss / backward / step`` per batch (evaluations.py:41-72) and one
scheduler step per epoch (evaluations.py: train_and_test_models), with the optimizer replaced by ``FusedAdamW`` and, under
``torch.distributed``, the gradient mean taken by ``BucketedGradReducer`` between backward and step.  Everything stays on
the launch stream: the only host synchronisation is the once-per-epoch read of the accumulated loss (the reference reads
``loss.item()`` every batch).  Datasets, metrics, checkpoints and plots stay out of scope (section 8: control plane).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from .optim import FusedAdamW


def train_step(model: nn.Module, data: torch.Tensor, target: torch.Tensor, optimizer: torch.optim.Optimizer,
               criterion: Optional[nn.Module] = None, reducer=None) -> torch.Tensor:
    """One batch: returns the (detached, device-resident) loss."""
    criterion = criterion if criterion is not None else nn.CrossEntropyLoss()
    optimizer.zero_grad()
    loss = criterion(model(data), target)
    loss.backward()
    if reducer is not None:
        reducer.finish()
    optimizer.step()
    return loss.detach()


def train_model_generic(model: nn.Module, train_batches: Iterable[Tuple[torch.Tensor, torch.Tensor]], device="cuda",
                        learning_rate: float = 1e-3, weight_decay: float = 1e-4, gamma: float = 0.8, epochs: int = 15,
                        reducer=None) -> List[float]:
    """generic_train.py:19-30 without the dataset / metric / checkpoint plumbing; returns the average loss per epoch.
    ``train_batches`` is re-iterated every epoch (a list of (data, target) pairs or a DataLoader)."""
    model.to(device).train()
    optimizer = FusedAdamW(model.parameters(), lr=learning_rate, weight_decay=weight_decay)
    scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=gamma)
    criterion = nn.CrossEntropyLoss()
    history = []
    for _ in range(epochs):
        total, n = torch.zeros((), device=device), 0
        for data, target in train_batches:
            total += train_step(model, data.to(device), target.to(device), optimizer, criterion, reducer)
            n += 1
        history.append(float(total) / max(n, 1))
        scheduler.step()
    return history
