"""Checkpoint I/O in the layout the reference trains with.

The reference saves through Lightning's ModelCheckpoint (train.py:43-49), resumes with `trainer.fit(ckpt_path=...)`
(train.py:34-39, 79) and reads the files back with `torch.load(...)["state_dict"]` (eval.py:79-80,
nerf_system_optmize.py:257-264, utils/__init__.py:4-26).  The file is a pickled dict

    {"epoch", "global_step", "pytorch-lightning_version", "state_dict", "optimizer_states", "lr_schedulers",
     "hyper_parameters"}

with `state_dict` keyed by the module tree (`nerf_coarse.xyz_encoding_1.0.weight`, `se3_refine.weight`, ...; pinned by
tests/golden/state_keys.json) and `optimizer_states` in torch.optim's per-parameter shape (FlatAdam.state_dict emits
that shape and accepts it back).  Interchange with the reference: its `torch.load(...)["state_dict"]` consumers (eval,
test-time optimisation, `load_ckpt`) read a file written here as they read their own, and a file written by the reference
loads here (weights, optimiser moments, schedulers, step counters).  Resuming one of OUR files with Lightning's
`trainer.fit(ckpt_path=...)` is not claimed: Lightning additionally wants its own `loops` / `callbacks` entries, which this
package does not emit (trainer.py keeps its private resume state under `upnerf_*` keys that Lightning ignores).

Everything is written from CPU copies, so a checkpoint saved on an MI355X opens anywhere."""
from __future__ import annotations

import os
import tempfile
from typing import Iterable, Mapping

import torch

LAYOUT_VERSION = "1.9.0"  # the Lightning release the reference pins (requirements.txt); informational only


def _cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().to("cpu", copy=True)
    if isinstance(obj, Mapping):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def checkpoint_dict(system, epoch: int = 0) -> dict:
    """The checkpoint of `system` (a NeRFSystem after setup()) as a plain dict of CPU tensors."""
    opts, scheds = _as_list(system.optimizers()), _as_list(system.lr_schedulers())
    return {
        "epoch": int(epoch),
        "global_step": int(system.global_step),
        "pytorch-lightning_version": LAYOUT_VERSION,
        "state_dict": _cpu(system.state_dict()),
        "optimizer_states": [_cpu(o.state_dict()) for o in opts],
        "lr_schedulers": [_cpu(s.state_dict()) for s in scheds],
        "hyper_parameters": dict(system.hparams),
    }


def save_checkpoint(system, path: str, epoch: int = 0) -> str:
    """Write `path` atomically (temp file in the same directory, then rename): an interrupted save never leaves a
    truncated last.ckpt for the next run to resume from."""
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    fd, tmp = tempfile.mkstemp(prefix=".ckpt_", dir=d)
    try:
        with os.fdopen(fd, "wb") as f:
            torch.save(checkpoint_dict(system, epoch), f)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise
    return path


def read_checkpoint(path_or_dict) -> dict:
    if isinstance(path_or_dict, Mapping):
        return dict(path_or_dict)
    ckpt = torch.load(path_or_dict, map_location="cpu", weights_only=False)
    if not isinstance(ckpt, Mapping):
        raise ValueError(f"{path_or_dict}: not a checkpoint dict")
    return dict(ckpt)


def load_checkpoint(system, path_or_dict, resume: bool = True, strict: bool = True) -> dict:
    """Restore `system` from a checkpoint.  resume=True also restores optimiser moments / step counts, scheduler state
    and global_step (what `trainer.fit(ckpt_path=...)` does); resume=False loads the weights only (eval / TTO use).
    The schedule position (`NeRF.progress`) travels inside state_dict; its host mirror is refreshed here."""
    ckpt = read_checkpoint(path_or_dict)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt  # a bare state_dict is accepted like utils/__init__.py:7
    missing, unexpected = system.load_state_dict(sd, strict=strict)
    if hasattr(system, "nerf_coarse") and hasattr(system, "set_progress"):
        system.set_progress(float(system.nerf_coarse.progress.data))  # one device read, at load time only
    if resume and "optimizer_states" in ckpt:
        opts, scheds = _as_list(system.optimizers()), _as_list(system.lr_schedulers())
        if len(opts) != len(ckpt["optimizer_states"]):
            raise ValueError(f"checkpoint holds {len(ckpt['optimizer_states'])} optimisers, the system configures "
                             f"{len(opts)} (pose.optimize differs?)")
        for o, s in zip(opts, ckpt["optimizer_states"]):
            o.load_state_dict(s)
        for sc, s in zip(scheds, ckpt.get("lr_schedulers", [])):
            sc.load_state_dict(s)
            for g, lr in zip(sc.optimizer.param_groups, s.get("_last_lr", [])):
                g["lr"] = lr
        system.global_step = int(ckpt.get("global_step", 0))
    return {"missing": list(missing), "unexpected": list(unexpected), "global_step": int(ckpt.get("global_step", 0)),
            "epoch": int(ckpt.get("epoch", 0))}


def extract_model_state_dict(ckpt_path, model_name: str = "model", prefixes_to_ignore: Iterable[str] = ()) -> dict:
    """Sub-state-dict of one module of the tree (`model_name.` stripped), as utils/__init__.py:4-19 returns it."""
    ckpt = read_checkpoint(ckpt_path)
    sd = ckpt.get("state_dict", ckpt)
    head = model_name + "."
    skip = tuple(prefixes_to_ignore)
    return {k[len(head):]: v for k, v in sd.items()
            if k.startswith(head) and not (skip and k[len(head):].startswith(skip))}


def load_ckpt(model, ckpt_path, model_name: str = "model", prefixes_to_ignore: Iterable[str] = ()) -> None:
    """Overlay the checkpoint's tensors for `model_name` on `model` (entries absent from the file keep their current
    values; utils/__init__.py:22-26)."""
    merged = dict(model.state_dict())
    merged.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore))
    model.load_state_dict(merged)
