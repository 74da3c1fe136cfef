"""Weight-decay wrappers the reference's factories put around a layer when ``l1_decay > 0``
(utils/regularization.py:57-159; used at layers/kan_conv.py:66-68 and in every sibling factory).

Semantics restated: the wrapper forwards to the wrapped module and registers a full backward hook on it; when the hook
fires, every selected parameter whose ``.grad`` is still ``None`` or all zeros gets ``.grad = penalty(parameter)``
(L1: ``weight_decay * sign(p)``, L2: ``weight_decay * p``), so the gradient autograd accumulates afterwards lands on top
of the penalty term.  (The all-zeros test reads the gradient back to the host, as the reference's does.)  Parameters and
``state_dict`` keys gain the ``module.`` prefix, exactly as with the reference wrapper.
"""
from __future__ import annotations

import abc
from typing import Optional

import torch
import torch.nn as nn


class WeightDecay(nn.Module):
    def __init__(self, module: nn.Module, weight_decay: float, name: Optional[str] = None):
        if weight_decay < 0.0:
            raise ValueError("Regularization's weight_decay should be greater than 0.0, got {}".format(weight_decay))
        super().__init__()
        self.module, self.weight_decay, self.name = module, weight_decay, name
        self.hook = self.module.register_full_backward_hook(self._weight_decay_hook)

    def remove(self):
        self.hook.remove()

    def _selected(self):
        if self.name is None:
            return list(self.module.parameters())
        return [p for n, p in self.module.named_parameters() if self.name in n]

    def _weight_decay_hook(self, *_):
        for p in self._selected():
            if p.grad is None or bool(torch.all(p.grad == 0.0)):
                p.grad = self.regularize(p)

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def extra_repr(self) -> str:
        return "weight_decay={}".format(self.weight_decay) + (", name={}".format(self.name) if self.name is not None else "")

    @abc.abstractmethod
    def regularize(self, parameter):
        ...


class L2(WeightDecay):
    def regularize(self, parameter):
        return self.weight_decay * parameter.data


class L1(WeightDecay):
    def regularize(self, parameter):
        return self.weight_decay * torch.sign(parameter.data)
