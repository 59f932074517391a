"""NeRFSystem.training_step replayed from captured HIP graphs.

An eager step issues ~370 kernel launches through ctypes, torch's allocator and the autograd engine: 16-17 ms of host
time for ~19 ms of device time (round 1), i.e. any kernel improvement beyond ~2 ms per step would have been invisible.
Here the device work of a step -- pose refine -> rays -> render_rays coarse + fine -> TransientNet -> loss -> backward ->
both Adam updates (models/nerf_system.py:150-228) -- is captured ONCE per shape signature into a HIP graph and replayed:
~0.2 ms of host time per step (copy the batch into the graph's static input tensors, one `upnerf_set_scalars` launch with
the step's scalars, one graph launch), the host bookkeeping (step counts, LR schedulers, progress) runs beside it.

Shape signature = (rays in the batch, schedule phase, n_s): the fine-sample split n_s = round(sched * N_importance)
(models/rendering.py:283) is the only shape that moves during training, and it takes 129 values over the whole candidate
schedule -- a new graph every ~1 900 steps in that window, none outside it.  The first step with a new signature runs
eagerly (it also warms caches and workspaces), the second one is captured, every later one is replayed.  All graphs share
one memory pool (they never run concurrently), and only the most recent few are kept.

Per-step scalars (band weights, schedule multiplier, Adam step sizes) are read by the kernels from device memory
(step_scalars.py), so a replayed step is bitwise identical to the eager step it replaces -- tests/test_graph_step.py.

Data parallel (world_size > 1): the gradient all-reduce stays OUTSIDE the graphs -- graph 1 = forward + backward + pack of
the flat gradient buffer, RCCL all-reduce issued eagerly, graph 2 = unpack + Adam -- so nothing depends on collective
capture support in the communication library."""
from __future__ import annotations

import warnings
from collections import OrderedDict
from typing import Dict, Optional

import os
import time

import torch
import torch.distributed  # noqa: F401 (is_initialized / get_backend below)

from .rendering import band_weights
from .step_scalars import StepScalars

__all__ = ["GraphedTrainingStep"]

_BATCH_KEYS = ("ray_infos", "directions", "c2w", "feats", "img_idx", "rgbs", "inv_depths")


class _Entry:
    __slots__ = ("g1", "g2", "scalars", "loss", "loss_d", "done", "grads", "replays", "n_sync")


class GraphedTrainingStep:
    def __init__(self, system, max_graphs: int = 3, eager_steps: int = 1):
        self.system = system
        p = next(system.parameters())
        if not p.is_cuda:
            raise RuntimeError("GraphedTrainingStep needs the system on the GPU")
        self.device = p.device
        self.stream = torch.cuda.Stream(self.device)
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs: "OrderedDict[tuple, _Entry]" = OrderedDict()
        self.static: Dict[int, Dict[str, torch.Tensor]] = {}
        self._static_flat: Dict[int, tuple] = {}
        self.seen: Dict[tuple, int] = {}
        self.max_graphs, self.eager_steps = max_graphs, eager_steps
        self.stats = {"eager": 0, "captures": 0, "replays": 0}
        nc, nf = system.nerf_coarse, getattr(system, "nerf_fine", None)
        if nf is not None and (nf.xyz_L, nf.dir_L, nf.c2f) != (nc.xyz_L, nc.dir_L, nc.c2f):
            raise ValueError("coarse and fine fields must share the encoding configuration")
        # An optimiser whose step() reads Python scalars bakes them into the captured launches: SGD would replay the learning
        # rate of the capture for ever (the LR scheduler silently ignored), torch's non-capturable Adam raises in the middle
        # of the capture.  Only optimisers that split into step_device / step_host (FlatAdam) or are capturable with the
        # learning rate on the device replay correctly; anything else runs eagerly (round-2 ADVICE).
        self.eager_reason = self._not_capturable()
        if self.eager_reason:
            warnings.warn(f"GraphedTrainingStep: {self.eager_reason}; the step is launched eagerly instead of replayed")

    def _not_capturable(self) -> Optional[str]:
        opts, _ = self.system._opts_scheds()
        for o in opts:
            if hasattr(o, "step_device"):
                continue
            for g in o.param_groups:
                if not g.get("capturable", False):
                    return f"{type(o).__name__} is not capturable"
                if not torch.is_tensor(g.get("lr")) and self._lr_moves(o):
                    return f"{type(o).__name__} keeps its learning rate as a Python number under an LR schedule"
        return None

    def _lr_moves(self, opt) -> bool:
        """Does an LR schedule change this optimiser's learning rate from step to step?"""
        for sch in self.system._opts_scheds()[1]:
            if getattr(sch, "optimizer", None) is not opt:
                continue
            name = type(sch).__name__
            if name == "ConstantLR" and getattr(sch, "factor", None) == 1.0:
                return False
            if name == "ExponentialLR" and getattr(sch, "gamma", None) == 1.0:
                return False
            return True
        return False

    # ---- signature ---------------------------------------------------------------------------------------------------
    def key(self, batch) -> tuple:
        s = self.system
        sm = s.get_schedule_mult(s._host_progress)
        phase = 0 if sm == 0 else (2 if sm == 1 else 1)
        n_s = round(sm * s.hparams["nerf.N_importance"]) if phase == 1 else 0  # rendering.py:283 (banker's rounding)
        return (int(batch["img_idx"].shape[0]), phase, n_s)

    def _providers(self):
        s, m = self.system, self.system.nerf_coarse

        def prog():
            return float(torch.tensor(m.host_progress, dtype=torch.float32))  # what render_rays evaluates

        return {"wk_xyz": lambda: band_weights(m.xyz_L, prog(), m.c2f),
                "wk_dir": lambda: band_weights(m.dir_L, prog(), m.c2f),
                "sched": lambda: [s.get_schedule_mult(s._host_progress)],
                "step": lambda: [float(s.global_step)]}  # Philox counter of the stratified-sampling draws

    # ---- capture -----------------------------------------------------------------------------------------------------
    def _static_batch(self, batch):
        R = int(batch["img_idx"].shape[0])
        st = self.static.get(R)
        if st is None:
            # the static inputs are views into ONE flat buffer (256-byte aligned pieces), so that a replay's batch arrives
            # by one pack launch instead of one copy per tensor
            keys = [k for k in _BATCH_KEYS if k in batch]
            words = {k: (batch[k].numel() * batch[k].element_size() + 3) // 4 for k in keys}
            offs, off = {}, 0
            for k in keys:
                offs[k] = off
                off += (words[k] + 63) // 64 * 64
            flat = torch.empty(max(off, 1), device=self.device, dtype=torch.float32)
            st = {}
            for k in keys:
                b = batch[k]
                if b.element_size() % 4 == 0 or (b.numel() * b.element_size()) % 4 == 0:
                    st[k] = flat[offs[k]:offs[k] + words[k]].view(b.dtype).view(b.shape)
                else:
                    st[k] = torch.empty_like(b)
            self.static[R] = st
            self._static_flat[R] = (flat, offs, words)
        return st

    def _stage_batch(self, batch, static):
        """batch -> static input buffers: one upnerf_pack launch when every tensor is a dense device tensor of whole 32-bit
        words (the usual case: the sampler's output), one copy per tensor otherwise."""
        from ._lib import PackDesc, check, lib, ptr, stream
        R = int(batch["img_idx"].shape[0])
        flat, offs, words = self._static_flat[R]
        ok = all(batch[k].is_cuda and batch[k].is_contiguous() and batch[k].shape == t.shape and batch[k].dtype == t.dtype
                 and t.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for k, t in static.items())
        if not ok or len(static) > 16:
            for k, t in static.items():
                t.copy_(batch[k], non_blocking=True)
            return
        descs = [PackDesc(batch[k].data_ptr(), 1, words[k], words[k], offs[k], words[k], 0) for k in static]
        arr = (PackDesc * len(descs))(*descs)
        check(lib.upnerf_pack(ptr(flat), arr, len(descs), 0, stream()), "upnerf_pack")

    def _capture(self, key, static) -> _Entry:
        s = self.system
        s._last_rays = None  # an autograd graph kept alive from an earlier step would pin its AccumulateGrad nodes
        if any(m.host_progress is None for m in s.models.values() if hasattr(m, "host_progress")):
            raise RuntimeError("graph capture needs the host mirror of NeRF.progress (use NeRFSystem.set_progress)")
        sync = s.grad_sync if (s.grad_sync is not None and s.grad_sync.world > 1) else None
        # With a process group alive its watchdog thread polls the events of every collective it still tracks.  Two things follow
        # (the second found on hardware in round 6, tools/rccl_probe.py --step: the N > 1 path had only ever run on gloo):
        #  - only THIS thread's calls may be held to the capture rules ("thread_local"; the work of the autograd thread lands
        #    in the capture through the stream);
        #  - on HIP that is not enough: a hipEventQuery by the watchdog while this thread captures comes back as
        #    hipErrorCapturedEvent ("operation not permitted on an event last recorded in a capturing stream") and the
        #    watchdog takes the process down -- every time when the capture follows an eager step's all-reduce within the
        #    watchdog's 100 ms period, which is exactly what a training run does.  So the device is drained and the watchdog
        #    given five of its periods to retire what it tracks before the capture begins (once per captured graph).
        pg_alive = torch.distributed.is_available() and torch.distributed.is_initialized()
        mode = "thread_local" if (sync is not None or pg_alive) else "global"
        if pg_alive and torch.distributed.get_backend() == "nccl":
            torch.cuda.synchronize(self.device)
            time.sleep(float(os.environ.get("UPNERF_CAPTURE_DRAIN_S", "0.5")))
        e = _Entry()
        e.scalars = StepScalars(self.device, self._providers())
        e.g1, e.g2, e.replays = torch.cuda.CUDAGraph(), None, 0
        with e.scalars:
            with torch.cuda.graph(e.g1, pool=self.pool, stream=self.stream, capture_error_mode=mode):
                e.loss, e.loss_d = s._step_backward(static)
                if sync is None:
                    e.done = s._step_update()
                    e.n_sync = 0
                else:
                    e.n_sync = sync.pack()
            if sync is not None:
                e.g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(e.g2, pool=self.pool, stream=self.stream, capture_error_mode=mode):
                    sync.unpack()
                    e.done = s._step_update()
        e.grads = [(p, p.grad) for p in s.parameters()]
        self.stats["captures"] += 1
        self.graphs[key] = e
        while len(self.graphs) > self.max_graphs:
            self.graphs.popitem(last=False)
        return e

    # ---- one training step -------------------------------------------------------------------------------------------
    def __call__(self, batch, batch_nb: int = 0):
        s = self.system
        key = self.key(batch)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            e = self.graphs.get(key)
            if self.eager_reason or (e is None and self.seen.get(key, 0) < self.eager_steps):
                self.seen[key] = self.seen.get(key, 0) + 1
                self.stats["eager"] += 1
                loss = s.training_step(batch, batch_nb)
            else:
                static = self._static_batch(batch)
                self._stage_batch(batch, static)
                if e is None:
                    e = self._capture(key, static)  # capture does not execute: the replay below performs this step
                self.graphs.move_to_end(key)
                e.scalars.upload()
                e.g1.replay()
                if e.g2 is not None:
                    s.grad_sync.reduce(e.n_sync)
                    e.g2.replay()
                for p, g in e.grads:  # the tensors THIS graph writes (another graph's capture may have replaced them)
                    p.grad = g
                s._step_host(e.loss, e.loss_d, e.done)
                e.replays += 1
                self.stats["replays"] += 1
                loss = e.loss
        cur.wait_stream(self.stream)
        return loss
