"""Optimiser / scheduler factory with the reference's semantics (utils/optim.py:20-49): Adam(eps=1e-8) and an
ExponentialLR whose gamma takes the learning rate from lr to lr_end over max_step steps."""
from __future__ import annotations

import math

import torch
from torch.optim import SGD, Adam, AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR


class FlatAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (lr, betas, eps; no weight decay / amsgrad) on ONE flat fp32 buffer, updated by the
    upnerf_adam HIP kernel.

    * every parameter's storage becomes a view into `flat_p` (so one kernel covers all tensors, and a data-parallel
      run has a single gradient buffer to all-reduce);
    * parameters whose `.grad` is None are skipped exactly like torch.optim.Adam skips them -- their moments and their
      per-parameter step count do not advance (which heads receive gradients depends on the schedule
      phase); the update is launched once per maximal run of consecutive live parameters with equal step count.
    Only fp32 CUDA parameters are supported (the HIP path has no CPU fallback)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps or any((not p.is_cuda) or p.dtype != torch.float32 for p in ps):
            raise ValueError("FlatAdam needs fp32 CUDA parameters")
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        self.flat_p = torch.empty(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        self.flat_m = torch.zeros(n, device=dev)
        self.flat_v = torch.zeros(n, device=dev)
        self._spans, off = [], 0
        with torch.no_grad():
            for p in ps:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)
                self._spans.append((p, off, k))
                off += k
        self._steps = [0] * len(ps)

    def _plan(self):
        """Maximal runs of consecutive parameters that have a gradient and share a step count."""
        live = [i for i, (p, _, _) in enumerate(self._spans) if p.grad is not None]
        runs = []
        if live:
            run = [live[0]]
            for i in live[1:]:
                if i == run[-1] + 1 and self._steps[i] == self._steps[run[0]]:
                    run.append(i)
                else:
                    runs.append(run)
                    run = [i]
            runs.append(run)
        return live, runs

    def _scalars(self, first: int):
        """(step_size, sqrt(bias_corr2)) of the update that takes parameter `first` to its NEXT step count, formed in
        double precision like torch.optim.Adam's Python scalars (the kernel receives them rounded to fp32)."""
        group = self.param_groups[0]
        b1, b2 = group["betas"]
        t = self._steps[first] + 1
        return (group["lr"] / (1 - b1 ** t), math.sqrt(1 - b2 ** t))

    @torch.no_grad()
    def step_device(self):
        """The device half of step(): launch the update(s).  When every live gradient is a contiguous fp32 CUDA tensor the
        kernels read the gradients where autograd left them (`self.in_place` = True: `flat_g` is NOT written and holds stale
        values -- read `p.grad`); otherwise the gradients are first gathered into `flat_g`.  No host state changes, so it can
        be captured in a HIP graph: under a capture (step_scalars.current()) the launches read the step size and bias
        correction from device memory, refreshed before every replay."""
        from . import step_scalars
        from ._lib import check, lib, ptr, stream
        group = self.param_groups[0]
        (b1, b2), eps = group["betas"], group["eps"]
        live, runs = self._plan()
        if not live:
            return []
        from ._lib import MAX_ADAM_DESC, AdamDesc
        dyn = step_scalars.current()
        grads = {i: self._spans[i][0].grad for i in live}
        in_place = all(g.is_contiguous() and g.dtype == torch.float32 and g.is_cuda and g.numel() == self._spans[i][2]
                       for i, g in grads.items())
        self.in_place = in_place  # (whether flat_g is current after this call)
        if not in_place:  # (strided or foreign gradients: gather them into the flat buffer first)
            torch._foreach_copy_([self.flat_g[self._spans[i][1]:self._spans[i][1] + self._spans[i][2]].view_as(self._spans[i][0])
                                  for i in live], [grads[i] for i in live])
        for run in runs:
            step_size, bc2s = self._scalars(run[0])
            dyn2 = dyn.ptr_fn(2, lambda first=run[0]: self._scalars(first)) if dyn is not None else None
            if in_place:  # the update reads every gradient where autograd left it: one launch per run of <= 96 tensors
                for c0 in range(0, len(run), MAX_ADAM_DESC):
                    part = run[c0:c0 + MAX_ADAM_DESC]
                    arr = (AdamDesc * len(part))(*[AdamDesc(grads[i].data_ptr(), self._spans[i][1], self._spans[i][2]) for i in part])
                    check(lib.upnerf_adam_gather(ptr(self.flat_p), ptr(self.flat_m), ptr(self.flat_v), arr, len(part), b1, b2, eps,
                                                 step_size, bc2s, dyn2, stream()), "upnerf_adam_gather")
                continue
            a, b = self._spans[run[0]][1], self._spans[run[-1]][1] + self._spans[run[-1]][2]
            p, g, m, v = self.flat_p[a:b], self.flat_g[a:b], self.flat_m[a:b], self.flat_v[a:b]
            check(lib.upnerf_adam(b - a, ptr(p), ptr(g), ptr(m), ptr(v), b1, b2, eps, step_size, bc2s, dyn2, stream()),
                  "upnerf_adam")
        return runs

    def step_host(self, runs):
        """The host half of step(): advance the per-parameter step counts of the runs step_device() updated."""
        for run in runs:
            for i in run:
                self._steps[i] += 1
        self._opt_called = True  # what torch's LR schedulers look at to order optimizer.step() / scheduler.step()

    @torch.no_grad()
    def step(self, closure=None):
        self.step_host(self.step_device())
        return None

    def state_dict(self):
        """torch.optim.Adam-shaped state (per-parameter step / exp_avg / exp_avg_sq) for checkpoint compatibility."""
        state = {i: {"step": torch.tensor(float(self._steps[i])), "exp_avg": self.flat_m[o:o + k].view_as(p).clone(),
                     "exp_avg_sq": self.flat_v[o:o + k].view_as(p).clone()}
                 for i, (p, o, k) in enumerate(self._spans) if self._steps[i] > 0}
        groups = [{**{k: v for k, v in g.items() if k != "params"}, "params": list(range(len(self._spans)))}
                  for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        for i, st in sd["state"].items():
            p, o, k = self._spans[int(i)]
            self._steps[int(i)] = int(st["step"])
            self.flat_m[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.flat_v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
        for g, gs in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in gs.items() if k != "params"})


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
        if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
            return FlatAdam(params, lr=lr, eps=1e-8)  # same update rule, one HIP launch per live run
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
