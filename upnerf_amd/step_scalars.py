"""Per-step scalars in device memory, for training steps replayed from a captured HIP graph.

A captured graph freezes every kernel argument.  The handful of host scalars that change from one optimisation step to
the next -- BARF band weights (models/nerf.py:137-143), the schedule multiplier of the loss terms (losses.py:21-64), the
learning rates and Adam bias corrections (utils/optim.py:20-45) -- are therefore read by the kernels from ONE small device
buffer when a capture is active (`upnerf_field_fwd_args.wk_xyz_dev`, `upnerf_ray_aux(wk_dir_dev)`, `upnerf_loss_args.
sched_dev`, `upnerf_adam_dyn`), and `upnerf_set_scalars` refreshes that buffer in front of every replay (values travel as
kernel arguments: no host staging buffer, no synchronisation).

Outside a capture `current()` is None and every call site passes its scalars by value, exactly as before; both routes
feed the kernels the same fp32 values, so an eager step and a replayed step are bitwise identical."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from ._lib import MAX_SCALARS, check, lib, stream

_CURRENT: Optional["StepScalars"] = None


def current() -> Optional["StepScalars"]:
    """The scalar table of the capture in progress, or None (eager execution)."""
    return _CURRENT


class StepScalars:
    def __init__(self, device, providers: Optional[Dict[str, Callable[[], Sequence[float]]]] = None):
        self.buf = torch.zeros(MAX_SCALARS, device=device, dtype=torch.float32)
        self.providers = dict(providers or {})   # name -> host function returning the current values
        self.named: Dict[str, Tuple[int, int]] = {}
        self.slots: List[Tuple[int, int, Callable[[], Sequence[float]]]] = []
        self.used = 0

    # ---- slot allocation (during capture) ----------------------------------------------------------------------------
    def _alloc(self, n: int, fn) -> int:
        if self.used + n > MAX_SCALARS:
            raise RuntimeError(f"more than {MAX_SCALARS} per-step scalars in one captured step")
        off = self.used
        self.used += n
        self.slots.append((off, n, fn))
        return off

    def ptr_named(self, name: str, n: int) -> int:
        """Device pointer of the `n` floats the provider `name` yields (allocated on first use)."""
        if name not in self.named:
            if name not in self.providers:
                raise KeyError(f"no provider for the per-step scalar '{name}'")
            self.named[name] = (self._alloc(n, self.providers[name]), n)
        off, k = self.named[name]
        if k != n:
            raise ValueError(f"per-step scalar '{name}' requested with {n} floats, registered with {k}")
        return self.buf.data_ptr() + 4 * off

    def ptr_fn(self, n: int, fn: Callable[[], Sequence[float]]) -> int:
        """Device pointer of `n` floats refreshed from `fn()` before every replay."""
        return self.buf.data_ptr() + 4 * self._alloc(n, fn)

    # ---- capture scope -----------------------------------------------------------------------------------------------
    def __enter__(self):
        global _CURRENT
        if _CURRENT is not None:
            raise RuntimeError("nested per-step scalar scopes")
        _CURRENT = self
        return self

    def __exit__(self, *exc):
        global _CURRENT
        _CURRENT = None
        return False

    # ---- before every replay -----------------------------------------------------------------------------------------
    def values(self) -> List[float]:
        vals = [0.0] * self.used
        for off, n, fn in self.slots:
            v = list(fn())
            if len(v) != n:
                raise ValueError("per-step scalar provider returned the wrong number of values")
            vals[off:off + n] = [float(x) for x in v]
        return vals

    def upload(self):
        """One launch on the current stream: buf[:used] = the providers' current values."""
        if self.used == 0:
            return
        vals = self.values()
        arr = (C.c_float * len(vals))(*vals)
        check(lib.upnerf_set_scalars(self.buf.data_ptr(), len(vals), arr, stream()), "upnerf_set_scalars")
