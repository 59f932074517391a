"""Optimiser / scheduler factory with the reference's semantics (utils/optim.py:20-49): Adam(eps=1e-8) and an
ExponentialLR whose gamma takes the learning rate from lr to lr_end over max_step steps."""
from __future__ import annotations

import torch
from torch.optim import SGD, Adam, AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR


def get_parameters(models):
    if isinstance(models, (list, tuple)):
        return [p for m in models for p in get_parameters(m)]
    if isinstance(models, dict):
        return [p for m in models.values() for p in get_parameters(m)]
    return list(models.parameters())


def get_optimizer(type, lr, models):
    params = get_parameters(models)
    if type == "sgd":
        return SGD(params, lr=lr)
    if type == "adam":
        return Adam(params, lr=lr, eps=1e-8)
    if type == "adamw":
        return AdamW(params, lr=lr)
    raise ValueError("optimizer not recognized!")


def get_scheduler(type, lr, lr_end, max_step, optimizer):
    if type == "cosine":
        return CosineAnnealingLR(optimizer, T_max=max_step, eta_min=1e-8)
    cls = getattr(torch.optim.lr_scheduler, type)
    assert type == "ExponentialLR" and lr_end, "only ExponentialLR with lr_end is configured by the reference"
    return cls(optimizer, gamma=(lr_end / lr) ** (1.0 / max_step))


def get_learning_rate(optimizer):
    for g in optimizer.param_groups:
        return g["lr"]
