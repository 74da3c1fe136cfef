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


class GraphedStep:
    """zero_grad / forward / loss / backward of ``model`` on fixed shapes, recorded ONCE into a HIP graph and replayed per batch.

    Every launch of the hot path goes through ctypes onto torch's current stream, so ``torch.cuda.graph`` captures the conv-KAN kernels next to
    torch's own; a replay costs one graph launch instead of the host's per-kernel work.  That pays where the step is launch-bound -- small models and
    single layers (BASELINE config 2: 0.23 ms replayed against 0.43-0.48 ms eager, bit-identical); KAN-VGG11 at bs 256 keeps the GPU busy either way.
    The weight packs are recorded unconditionally (``ops.always_pack``): the replay reads the weights as they are at replay time, so an optimizer may
    update them in place between replays.  Keep the optimizer step OUTSIDE the graph (AdamW's bias correction takes the step count as a kernel
    argument) and do not call ``zero_grad`` between replays: the recorded backward assigns ``p.grad`` (static tensors) rather than accumulating.

        step = GraphedStep(model, example_data, example_target)
        for data, target in batches:
            loss = step(data, target)          # device-resident, overwritten by the next call
            optimizer.step()
    """

    def __init__(self, model: nn.Module, example_data: torch.Tensor, example_target: torch.Tensor, criterion: Optional[nn.Module] = None,
                 warmup: int = 3):
        from . import ops
        if not example_data.is_cuda:
            raise ValueError("GraphedStep needs device tensors")
        self.model = model
        self.criterion = criterion if criterion is not None else nn.CrossEntropyLoss()
        self.data, self.target = example_data.clone(), example_target.clone()
        side = torch.cuda.Stream(device=example_data.device)
        side.wait_stream(torch.cuda.current_stream(example_data.device))
        with torch.cuda.stream(side):                      # warm-up off the capture stream: plans, workspaces, torch's lazy initialisations
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(example_data.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        self.model.zero_grad(set_to_none=True)             # recorded backward then ASSIGNS the gradients (tensors of the graph's pool)
        with ops.always_pack(), torch.cuda.graph(self.graph):
            self.loss = self._eager()

    def _eager(self) -> torch.Tensor:
        self.model.zero_grad(set_to_none=True)
        loss = self.criterion(self.model(self.data), self.target)
        loss.backward()
        return loss.detach()

    def __call__(self, data: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        if data.shape != self.data.shape or target.shape != self.target.shape:
            raise ValueError(f"GraphedStep was recorded for {tuple(self.data.shape)} / {tuple(self.target.shape)}, got {tuple(data.shape)} / {tuple(target.shape)}")
        self.data.copy_(data, non_blocking=True)
        self.target.copy_(target, non_blocking=True)
        self.graph.replay()
        return self.loss


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
