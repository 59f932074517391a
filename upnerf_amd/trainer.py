"""Training loop around NeRFSystem.training_step with the behaviour the reference configures on Lightning's Trainer
(train.py:43-79), without Lightning:

  * `max_steps` counts optimiser steps, i.e. TWO per iteration when poses are optimised (train.py:64-67 doubles it for
    that reason); the loop ends when `system.global_step` reaches it;
  * validation every `val_check_interval` of an epoch (a fraction in (0, 1], or an integer number of batches), over all
    validation batches, `val/psnr` = their mean (nerf_system.py:318-324);
  * ModelCheckpoint(save_last=True, monitor="val/psnr", mode="max", save_top_k=2): after every validation the checkpoint
    is written as `epoch=E-step=S.ckpt` if its metric is among the best k, the one that falls out is deleted, and
    `last.ckpt` always follows the newest state;
  * `fit(..., ckpt_path=...)` (or an existing `<dirpath>/last.ckpt`, train.py:34-39) resumes: weights, optimiser and
    scheduler state, global step, and the position inside the epoch -- the batches already consumed are skipped, so a
    resumed run sees the same batches as an uninterrupted one.

Batches come from any `batches(epoch[, start]) -> iterable of dicts` (GpuRaySampler.batches bound to its arguments); the loop
itself never touches the host side of a batch and never synchronises with the device outside validation."""
from __future__ import annotations

import inspect
import math
import os
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch

from .checkpoint import checkpoint_dict, load_checkpoint, read_checkpoint


class TopKCheckpoints:
    """Bookkeeping of ModelCheckpoint(save_top_k=k, save_last=True): which files exist and which one to drop."""

    def __init__(self, dirpath: str, k: int = 2, mode: str = "max"):
        if mode not in ("max", "min"):
            raise ValueError("mode must be 'max' or 'min'")
        self.dirpath, self.k, self.sign = dirpath, k, 1.0 if mode == "max" else -1.0
        self.best: Dict[str, float] = {}  # path -> monitored value

    @property
    def last_path(self) -> str:
        return os.path.join(self.dirpath, "last.ckpt")

    def best_path(self) -> Optional[str]:
        return max(self.best, key=lambda p: self.sign * self.best[p]) if self.best else None

    def update(self, metric: float, epoch: int, step: int, save: Callable[[str], None]) -> Optional[str]:
        """Called after a validation: saves `last.ckpt`, and the named checkpoint when `metric` enters the top k.
        Returns the path of the named checkpoint, or None when the metric did not make it."""
        os.makedirs(self.dirpath, exist_ok=True)
        path = None
        metric = float(metric)
        worst = min(self.best, key=lambda p: self.sign * self.best[p]) if self.best else None
        enters = self.k != 0 and not math.isnan(metric) and (
            len(self.best) < self.k or self.k < 0 or self.sign * metric > self.sign * self.best[worst])
        if enters:
            path = os.path.join(self.dirpath, f"epoch={epoch}-step={step}.ckpt")
            save(path)
            self.best[path] = metric
            if 0 < self.k < len(self.best):
                drop = min(self.best, key=lambda p: self.sign * self.best[p])
                del self.best[drop]
                if os.path.exists(drop) and drop != path:
                    os.unlink(drop)
        save(self.last_path)
        return path

    def state_dict(self) -> dict:
        return {"best": dict(self.best), "k": self.k, "sign": self.sign}

    def load_state_dict(self, sd: dict) -> None:
        self.best = {p: float(v) for p, v in sd.get("best", {}).items() if os.path.exists(p)}


def _save(system, path: str, extra: dict) -> None:
    d = checkpoint_dict(system, epoch=extra["upnerf_loops"]["epoch"])
    d.update(extra)
    tmp = path + ".tmp"
    torch.save(d, tmp)
    os.replace(tmp, path)


class Trainer:
    def __init__(self, max_steps: int, val_check_interval=0.25, dirpath: Optional[str] = None, save_top_k: int = 2,
                 monitor: str = "val/psnr", mode: str = "max", seed: int = 0, log: Optional[Callable[[dict], None]] = None,
                 write_checkpoints: bool = True, graph: bool = True):
        self.graph = graph  # replay the step from HIP graphs (graph_step.py; bitwise the eager step); False = eager launches
        self.max_steps, self.val_check_interval = int(max_steps), val_check_interval
        self.dirpath, self.monitor, self.seed = dirpath, monitor, seed
        self.write_checkpoints = write_checkpoints  # False on ranks > 0: they read last.ckpt but never write
        self.ckpts = TopKCheckpoints(dirpath, save_top_k, mode) if dirpath else None
        self.log = log
        self.epoch = 0
        self.batch_in_epoch = 0  # batches of the current epoch already trained on
        self.history: List[dict] = []  # one entry per validation

    # ---- helpers -----------------------------------------------------------------------------------------------------
    def _val_every(self, n_batches: int) -> int:
        v = self.val_check_interval
        if isinstance(v, float):
            if not 0.0 < v <= 1.0:
                raise ValueError("a fractional val_check_interval must be in (0, 1]")
            return max(1, int(n_batches * v))  # Lightning: int(num_training_batches * val_check_interval)
        return max(1, int(v))

    def validate(self, system, val_batches: Sequence[dict]) -> dict:
        outs = [system.validation_step(b, i) for i, b in enumerate(val_batches)]
        for o in outs:
            o.pop("results", None)  # full-image maps: not needed for the epoch summary
        res = system.validation_epoch_end(outs) or {}
        return {k: float(v) for k, v in res.items()}

    def _checkpoint(self, system, metrics: dict) -> None:
        if self.ckpts is None or not self.write_checkpoints:
            return
        # private resume state under names Lightning does not reserve (its own `loops` / `callbacks` entries have another
        # structure: a file with those keys in this shape would break trainer.fit(ckpt_path=...) of the reference)
        extra = {"upnerf_loops": {"epoch": self.epoch, "batch_in_epoch": self.batch_in_epoch, "seed": self.seed},
                 "upnerf_topk": None,
                 # the stratified-sampling draws of render_rays come from torch's generators (rank 0's; every rank of a
                 # data-parallel run seeds identically, trainer.setup_seed)
                 "upnerf_rng": {"torch": torch.get_rng_state(),
                                "cuda": torch.cuda.get_rng_state() if torch.cuda.is_available() else None}}

        def save(path):
            extra["upnerf_topk"] = self.ckpts.state_dict()
            _save(system, path, extra)

        self.ckpts.update(metrics.get(self.monitor, float("nan")), self.epoch, int(system.global_step), save)

    def resume(self, system, ckpt_path: str) -> None:
        ck = read_checkpoint(ckpt_path)
        load_checkpoint(system, ck, resume=True)
        # Resume state lives under upnerf_* keys (a Lightning checkpoint of the reference carries its own, differently
        # shaped `loops` / `callbacks`).  Files written by round 1 of this package kept it under the bare names: read those
        # when the new keys are absent and the old ones have OUR shape; anything else resumes at the start of the stored
        # epoch -- loudly, because batches already consumed will then be replayed.
        loops, topk, rng = ck.get("upnerf_loops"), ck.get("upnerf_topk"), ck.get("upnerf_rng")
        legacy = ck.get("loops")
        if loops is None and isinstance(legacy, dict) and "batch_in_epoch" in legacy:
            loops = legacy
            topk = topk if topk is not None else (ck.get("callbacks") if isinstance(ck.get("callbacks"), dict) else None)
            rng = rng if rng is not None else ck.get("rng_states")
        if loops is None:
            import warnings
            warnings.warn(f"{ckpt_path}: no resume position (upnerf_loops) in the checkpoint: resuming at the start of epoch "
                          f"{int(ck.get('epoch', 0))}; batches of that epoch already trained on will be seen again")
            loops = {}
        self.epoch = int(loops.get("epoch", ck.get("epoch", 0)))
        self.batch_in_epoch = int(loops.get("batch_in_epoch", 0))
        if self.ckpts is not None and isinstance(topk, dict) and "best" in topk:
            self.ckpts.load_state_dict(topk)
        rng = rng or {}
        if rng.get("torch") is not None:
            torch.set_rng_state(rng["torch"])
        if rng.get("cuda") is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(rng["cuda"])

    # ---- the loop ----------------------------------------------------------------------------------------------------
    def fit(self, system, train_batches: Callable[[int], Iterable[dict]], n_batches_per_epoch: int,
            val_batches: Sequence[dict] = (), ckpt_path: Optional[str] = None) -> "Trainer":
        if ckpt_path is None and self.ckpts is not None and os.path.isfile(self.ckpts.last_path):
            ckpt_path = self.ckpts.last_path  # train.py:37-40
        if ckpt_path is not None:
            self.resume(system, ckpt_path)
        every = self._val_every(n_batches_per_epoch)
        step = system.training_step
        if self.graph and getattr(system, "supports_graph_step", False) and next(system.parameters()).is_cuda:
            from .graph_step import GraphedTrainingStep
            step = GraphedTrainingStep(system)  # after resume(): the captured graphs hold the addresses of the Adam state
        self.step_fn = step
        while system.global_step < self.max_steps:
            skip = self.batch_in_epoch
            if skip and len(inspect.signature(train_batches).parameters) >= 2:
                it = enumerate(train_batches(self.epoch, skip), skip)  # provider fast-forwards: nothing gathered twice
            else:
                it = enumerate(train_batches(self.epoch))
            for i, batch in it:
                if i < skip:
                    continue  # consumed before the checkpoint this run resumed from
                step(batch, i)
                self.batch_in_epoch = i + 1
                done = system.global_step >= self.max_steps
                if (self.batch_in_epoch % every == 0 or done) and len(val_batches):
                    metrics = self.validate(system, val_batches)
                    metrics.update(epoch=self.epoch, step=int(system.global_step))
                    self.history.append(metrics)
                    if self.log is not None:
                        self.log(metrics)
                    self._checkpoint(system, metrics)
                if done:
                    return self
            self.epoch += 1
            self.batch_in_epoch = 0
        return self


def setup_seed(seed: int) -> None:
    import random
    import numpy as np
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def fit_from_config(hparams: dict, train_dataset, val_dataset=None, device="cuda", rank: int = 0, world_size: int = 1,
                    log: Optional[Callable[[dict], None]] = None):
    """What train.py:23-79 does with a parsed configuration, on this package: seed, `<out_dir>/<scene>/<exp>` with
    `config.yaml` and `ckpts/`, resume from `resume_ckpt` or an existing `ckpts/last.ckpt`, optimiser-step budget
    doubled under pose optimisation, validation every `val.log_interval` of an epoch over the validation split.

    `train_dataset` carries the reference's ray buffers (GpuRaySampler.from_dataset), `N_images_train` and `white_back`;
    `val_dataset[i]` is one validation image as the reference's val `__getitem__` returns it (a dict of tensors).
    With world_size > 1 call it once per rank after torch.distributed is initialised (parallel.py)."""
    from .config import save_yaml
    from .nerf_system import NeRFSystem
    from .ray_sampler import GpuRaySampler
    if not hparams["pose.optimize"] and hparams.get("pose.c2f") is not None:
        raise AssertionError("if you don't optimize poses, pose.c2f must be None")
    setup_seed(hparams["seed"])
    save_dir = os.path.join(hparams.get("out_dir", "./outputs"), str(hparams.get("scene_name", "scene")),
                            str(hparams.get("exp_name", "exp")))
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
    system = NeRFSystem(hparams, train_dataset, val_dataset)
    system.setup()
    system.to(device)
    if world_size > 1:
        system.enable_data_parallel()
    sampler = GpuRaySampler.from_dataset(train_dataset, device)
    bs = int(hparams["train.batch_size"])
    n_batches = sampler.n_batches(bs, world_size)
    batches = lambda epoch, start=0: sampler.batches(bs, seed=int(hparams["seed"]), epoch=epoch, rank=rank,
                                                     world_size=world_size, start=start)
    val = []
    for i in range(len(val_dataset) if val_dataset is not None else 0):
        item = val_dataset[i]
        val.append({k: (v.to(device)[None] if torch.is_tensor(v) else v) for k, v in item.items()})
    max_steps = int(hparams["max_steps"]) * (2 if hparams["pose.optimize"] else 1)  # train.py:64-67
    trainer = Trainer(max_steps, hparams.get("val.log_interval", 0.25), dirpath=os.path.join(save_dir, "ckpts"),
                      seed=int(hparams["seed"]), log=log, write_checkpoints=rank == 0,
                      graph=bool(hparams.get("hip.graph", True)))
    if rank == 0:
        save_yaml(hparams, os.path.join(save_dir, "config.yaml"))
    trainer.fit(system, batches, n_batches, val, ckpt_path=hparams.get("resume_ckpt"))
    return system, trainer
