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
import time
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, local_rank, world_size) from torchrun's environment; initialises the default process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # the host driver of the pool only supports dmabuf IPC: without this RCCL's buffer exchange fails with
        # `hipIpcGetMemHandle: invalid argument`.  Read when the HSA runtime starts, i.e. at this process's first HIP call --
        # which has not happened yet in a rank that comes here first (bench.py, the Trainer); exported already where the launcher did
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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


def _gather(flat: torch.Tensor, views, tensors, scatter: bool = False) -> None:
    """views[i] <- tensors[i] (scatter: tensors[i] <- views[i]); `views` tile the start of `flat` in order.  On the GPU through
    upnerf_pack, 48 tensors per launch (three launches of ~8 us for the step's ~130 gradients): torch._foreach_copy_ spends two
    33 us multi-tensor launches on the same 9 MB, and pack + unpack were 0.14 of the 0.36 ms the exchange costs a replayed step
    on one device (tools/rccl_probe.py).  CPU tensors (gloo tests) and anything that is not contiguous
    fp32 keep the torch path."""
    ok = os.environ.get("UPNERF_GATHER", "1") != "0" and flat.is_cuda and all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() < 2 ** 31 for t in tensors)
    if ok:
        try:
            from ._lib import PackDesc, check, lib, ptr, stream
        except Exception:  # noqa: BLE001 -- the exchange does not depend on the extension
            ok = False
    if not ok:
        if scatter:
            torch._foreach_copy_(list(tensors), list(views))
        else:
            torch._foreach_copy_(list(views), list(tensors))
        return
    base = flat.data_ptr()
    descs = []
    for v, t in zip(views, tensors):
        n = t.numel()
        if n:
            descs.append(PackDesc(ptr=t.data_ptr(), rows=1, cols=n, src_ld=n, dst_off=(v.data_ptr() - base) // 4, dst_ld=n, accumulate=0))
    for i in range(0, len(descs), 48):
        chunk = descs[i:i + 48]
        arr = (PackDesc * len(chunk))(*chunk)
        check(lib.upnerf_pack(ptr(flat), arr, len(chunk), 1 if scatter else 0, stream()), "upnerf_pack")


class GradSync:
    """Average the gradients of `params` across ranks with one flat all-reduce -- or, with `early` given, two: the
    parameters in `early` (the fine field: autograd produces its gradients first) are reduced on a side stream
    as soon as the last of them has its gradient, while the backward pass of the coarse field, the tables and the pose
    still runs; everything else follows at the end.

    Which parameters receive a gradient depends only on the schedule phase, so the number of `early`
    gradients to wait for is LEARNED per phase key (`begin(key)`): the first step of a phase reduces everything at the end,
    later steps launch the early bucket from a post-accumulate-grad hook.  Gradients filled at the end of the backward pass
    (ops._DeferredWgrads / _DeferredEmbeds) must not be in `early`."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None, check: bool = False,
                 early: Iterable[torch.nn.Parameter] = ()):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group, self.check = group, check
        self._flat = None
        self._live, self._views, self._n = [], [], 0
        early_ids = {id(p) for p in early}
        self.early: List[torch.nn.Parameter] = [p for p in self.params if id(p) in early_ids]
        self._expected = {}        # phase key -> number of early parameters that receive a gradient
        self._key, self._seen, self._work = None, 0, None
        self._early_flat, self._early_live, self._stream = None, [], None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.early]
        self.stats = {"early_launches": 0, "late_only": 0}
        # communication attribution (bench.py `comm`): with `timing` on, every all-reduce is bracketed -- by HIP events on the
        # compute stream (no host synchronisation: the pairs are read back in comm_summary()), by the host clock for a
        # CPU group -- and the floats it moved are counted
        self.timing = False
        self._pairs, self._host_s, self._floats, self._reduces = [], 0.0, 0, 0

    # ---- early bucket ------------------------------------------------------------------------------------------------------
    def begin(self, key) -> None:
        """Start of a backward pass in schedule phase `key`."""
        self._key, self._seen, self._work, self._early_live = key, 0, None, []

    def _on_grad(self, p) -> None:
        if self._key is None or self.world == 1:
            return
        self._seen += 1
        if self._expected.get(self._key) == self._seen:
            self._launch_early()

    def _launch_early(self) -> None:
        live = [p for p in self.early if p.grad is not None]
        n = sum(p.grad.numel() for p in live)
        dev = live[0].grad.device
        if self._early_flat is None or self._early_flat.device != dev:
            self._early_flat = torch.empty(sum(p.numel() for p in self.early), device=dev, dtype=torch.float32)
        flat = self._early_flat[:n]
        views, off = [], 0
        for p in live:
            k = p.grad.numel()
            views.append(flat[off:off + k].view_as(p.grad))
            off += k
        if dev.type == "cuda":
            if self._stream is None:
                self._stream = torch.cuda.Stream(dev)
            self._stream.wait_stream(torch.cuda.current_stream(dev))  # the gradients are complete on the compute stream
            with torch.cuda.stream(self._stream):
                _gather(flat, views, [p.grad for p in live])
                self._work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                # counted here, timed in _finish_early: an asynchronous collective runs on the backend's own stream, events
                # around its enqueue would bracket nothing -- what the step pays for it is the wait at the end of backward
                self._mark_end(None, dev, n)
        else:
            _gather(flat, views, [p.grad for p in live])
            self._work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._mark_end(None, dev, n)  # (asynchronous on a CPU group: its wait is timed in _finish_early)
        self._early_live, self._early_views = live, views
        self.stats["early_launches"] += 1

    def _finish_early(self) -> set:
        """Wait for the early all-reduce and write the averages back; returns the ids of the parameters it covered."""
        if self._work is None:
            return set()
        t0 = time.perf_counter()
        dev = self._early_flat.device
        ev = self._mark(dev) if dev.type == "cuda" else None
        self._work.wait()
        if self.timing and dev.type != "cuda":
            self._host_s += time.perf_counter() - t0
        flat = self._early_flat[:sum(v.numel() for v in self._early_views)]
        if flat.is_cuda:
            torch.cuda.current_stream(dev).wait_stream(self._stream)
            if ev is not None:  # the EXPOSED part of the early all-reduce: how long the compute stream stood waiting for it
                end = torch.cuda.Event(enable_timing=True)
                end.record(torch.cuda.current_stream(dev))
                self._pairs.append((ev, end))
        flat.mul_(1.0 / self.world)
        _gather(flat, self._early_views, [p.grad for p in self._early_live], scatter=True)
        self._work = None
        return {id(p) for p in self._early_live}

    @property
    def world(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    # The exchange in three pieces (pack / reduce / unpack) so that a captured training step can keep the collective
    # OUTSIDE its two HIP graphs: graph 1 ends with pack(), reduce() runs eagerly, graph 2 starts with unpack().
    def pack(self, skip: set = frozenset()) -> int:
        """Copy every gradient that exists (and is not in `skip`) into the flat buffer; returns the number of floats."""
        self._live = [p for p in self.params if p.grad is not None and id(p) not in skip]
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
        _gather(flat, self._views, [p.grad for p in self._live])
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
        ev = self._mark(flat.device)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        self._mark_end(ev, flat.device, n)
        flat.mul_(1.0 / self.world)

    # ---- communication attribution -------------------------------------------------------------------------------------------
    def _mark(self, dev):
        if not self.timing:
            return None
        if dev.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            return ev
        return time.perf_counter()

    def _mark_end(self, start, dev, n) -> None:
        if not self.timing:
            return
        self._floats += n
        self._reduces += 1
        if start is None:
            return
        if dev.type == "cuda":
            end = torch.cuda.Event(enable_timing=True)
            end.record(torch.cuda.current_stream(dev))
            self._pairs.append((start, end))
        else:
            self._host_s += time.perf_counter() - start

    def comm_reset(self, timing: bool = True) -> None:
        self.timing = timing
        self._pairs, self._host_s, self._floats, self._reduces = [], 0.0, 0, 0
        self.stats["early_launches"] = self.stats["late_only"] = 0

    def comm_summary(self, steps: int) -> dict:
        """Per-step communication of the steps since comm_reset(): time inside the all-reduce calls (HIP events on their stream;
        the caller has synchronised the device), bytes they moved, how many of the steps launched the early bucket."""
        ms = sum(a.elapsed_time(b) for a, b in self._pairs) + self._host_s * 1e3
        steps = max(steps, 1)
        return {"allreduce_ms": ms / steps, "bytes": 4 * self._floats / steps, "allreduces_per_step": self._reduces / steps,
                "early_launches": self.stats["early_launches"], "late_only": self.stats["late_only"], "world": self.world,
                "clock": "HIP events on the compute stream around each blocking all-reduce and around the wait for the early one (its exposed part)" if self._pairs else "host clock around each all-reduce (CPU group)"}

    def unpack(self) -> None:
        """Averaged gradients back into the parameters' .grad tensors."""
        if self._n == 0 or self.world == 1:
            return
        _gather(self._flat, self._views, [p.grad for p in self._live], scatter=True)

    def __call__(self) -> int:
        """All-reduce (mean) every gradient that exists; returns the number of floats exchanged (at the end of backward)."""
        if self._key is not None and self.world > 1:  # remember how many early gradients this phase produces
            self._expected[self._key] = sum(1 for p in self.early if p.grad is not None)
        done = self._finish_early()
        if not done:
            self.stats["late_only"] += 1
        n = self.pack(skip=done)
        self.reduce()
        self.unpack()
        self._key = None
        return n + sum(p.grad.numel() for p in self._early_live if id(p) in done)
