"""Data parallelism over the GPUs of one node: rays shard across ranks, all model state is replicated, and ONE
all-reduce of a flat fp32 gradient buffer per step averages the gradients (the reference gets the same effect from
Lightning's implicit DDP, train.py:70-72).

One process per GPU; torch.distributed backend "nccl" is RCCL on ROCm (xGMI inside the node), "gloo" on CPU for
the unit tests.  ~9 MB per step: latency bound, so a single flat buffer (not per-tensor calls, not DDP buckets).

Unused parameters: which parameters receive a gradient depends only on the schedule phase, which
every rank derives from the same host step counter -- the set is identical on all ranks by construction, and it is
checked (cheaply, by count and total size) in debug mode."""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from torchrun's environment; initialises the default process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # UPNERF_DIST_BACKEND: diagnostic override (e.g. "gloo" to exercise the multi-rank path of bench.py on a
        # single-GPU box, together with UPNERF_SHARE_DEVICE=1); the product path is "nccl" = RCCL
        backend = backend or os.environ.get("UPNERF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if os.environ.get("UPNERF_SHARE_DEVICE"):
        local = 0  # diagnostic: every rank on cuda:0
    return rank, local, world


class GradSync:
    """Average the gradients of `params` across ranks with one flat all-reduce."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, check: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group, self.check = group, check
        self._flat = None

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def __call__(self) -> int:
        """All-reduce (mean) every gradient that exists; returns the number of floats exchanged."""
        if self.world == 1:
            return 0
        live = [p for p in self.params if p.grad is not None]
        n = sum(p.grad.numel() for p in live)
        if n == 0:
            return 0
        if self._flat is None or self._flat.numel() < n or self._flat.device != live[0].grad.device:
            self._flat = torch.empty(n, device=live[0].grad.device, dtype=torch.float32)
        flat = self._flat[:n]
        if self.check:
            sig = torch.tensor([len(live), n], device=flat.device, dtype=torch.int64)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not (torch.equal(lo, sig) and torch.equal(hi, sig)):
                raise RuntimeError("ranks disagree on which parameters received gradients")
        off = 0
        views = []
        for p in live:
            k = p.grad.numel()
            views.append(flat[off:off + k].view_as(p.grad))
            off += k
        torch._foreach_copy_(views, [p.grad for p in live])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)
        torch._foreach_copy_([p.grad for p in live], views)
        return n
