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
from typing import Iterable, List, Optional

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
        self._live, self._views, self._n = [], [], 0

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    # The exchange in three pieces (pack / reduce / unpack) so that a captured training step can keep the collective
    # OUTSIDE its two HIP graphs: graph 1 ends with pack(), reduce() runs eagerly, graph 2 starts with unpack().
    def pack(self) -> int:
        """Copy every gradient that exists into the flat buffer; returns the number of floats (0: nothing to do)."""
        self._live = [p for p in self.params if p.grad is not None]
        n = sum(p.grad.numel() for p in self._live)
        self._n = n
        if n == 0 or self.world == 1:
            return 0
        dev = self._live[0].grad.device
        if self._flat is None or self._flat.device != dev:
            # allocated ONCE at the size of all parameters: captured graphs keep pointers into it
            self._flat = torch.empty(sum(p.numel() for p in self.params), device=dev, dtype=torch.float32)
        flat = self._flat[:n]
        off, self._views = 0, []
        for p in self._live:
            k = p.grad.numel()
            self._views.append(flat[off:off + k].view_as(p.grad))
            off += k
        torch._foreach_copy_(self._views, [p.grad for p in self._live])
        return n

    def reduce(self, n: Optional[int] = None) -> None:
        """All-reduce (mean) of the first n floats of the packed buffer (default: what the last pack() wrote)."""
        n = self._n if n is None else n
        if n == 0 or self.world == 1:
            return
        flat = self._flat[:n]
        if self.check:
            sig = torch.tensor([len(self._live), n], device=flat.device, dtype=torch.int64)
            lo, hi = sig.clone(), sig.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            if not (torch.equal(lo, sig) and torch.equal(hi, sig)):
                raise RuntimeError("ranks disagree on which parameters received gradients")
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.mul_(1.0 / self.world)

    def unpack(self) -> None:
        """Averaged gradients back into the parameters' .grad tensors."""
        if self._n == 0 or self.world == 1:
            return
        torch._foreach_copy_([p.grad for p in self._live], self._views)

    def __call__(self) -> int:
        """All-reduce (mean) every gradient that exists; returns the number of floats exchanged."""
        n = self.pack()
        self.reduce()
        self.unpack()
        return n
