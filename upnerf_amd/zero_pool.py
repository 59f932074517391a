"""One zero fill per training step for the buffers that must start at zero.

A step asks for about a dozen zeroed fp32 buffers (packed parameter images and their gradients, the running maxima of the
weight-gradient scales, the loss kernel's gradient arena): a 5 us fill launch each.  Inside `with zero_pool.step(device):`
the requests are served as 256-byte aligned slices of ONE arena zeroed by one launch.  The arena is sized from the request
sequence the previous step made; a request that does not match that sequence (another schedule phase, another shape) and
every request outside a step fall back to `torch.zeros`, so a caller never sees anything but zeros.  Under graph capture the
arena is one allocation of the graph's pool like any other tensor of the step (graph_step.py runs one eager step per shape
signature before it captures, which is where the capture's plan comes from).

Lifetime (r5 ADVICE).  The slices are views that may outlive the step: the flat parameter gradient of packing._PackFn.backward is
one of them and autograd's AccumulateGrad may adopt it as a parameter's .grad.  Every step allocates a FRESH arena (`_S.arena` is
dropped when the scope closes; nothing is ever handed out twice), so a retained view can never be overwritten by a later step --
its cost is memory: one kept .grad pins the whole arena of its step (packed images, dP, the loss arena: ~20 MB at the headline
shape) until the next zero_grad(set_to_none=True).  tests/test_hip_parity.py::test_gradients_of_consecutive_eager_steps_do_not_alias
holds the no-aliasing half of this."""
from __future__ import annotations

import contextlib
import os

import torch

__all__ = ["step", "zeros"]

_PAD = 64  # floats: every slice starts on a 256-byte boundary
ENABLED = os.environ.get("UPNERF_ZERO_POOL", "1") != "0"  # (0: a fill per request, for A/B runs)


class _State:
    plan = None     # sizes (in floats) requested during the last complete step
    rec = None      # sizes requested so far in this step (None outside a step)
    arena = None    # this step's zeroed buffer, while the requests follow the plan
    cursor = 0
    i = 0
    device = None


_S = _State()


def _padded(n: int) -> int:
    return (n + _PAD - 1) // _PAD * _PAD


@contextlib.contextmanager
def step(device):
    """Scope of one training step: the first request allocates the arena the previous step's requests ask for."""
    if _S.rec is not None:  # nested scope (a system stepping inside another one): the outer one keeps the pool
        yield
        return
    _S.rec, _S.arena, _S.cursor, _S.i, _S.device = [], None, 0, 0, torch.device(device)
    try:
        yield
        _S.plan = (_S.device, tuple(_S.rec))
    finally:
        _S.rec, _S.arena = None, None


def _zeroed(n: int, device) -> torch.Tensor:
    """n zeroed floats by the library's own fill (upnerf_zero) on the GPU, torch.zeros elsewhere."""
    dev = torch.device(device)
    if dev.type != "cuda" or n <= 0:
        return torch.zeros(max(n, 0), device=dev, dtype=torch.float32)
    from ._lib import check, lib, stream
    t = torch.empty(n, device=dev, dtype=torch.float32)
    check(lib.upnerf_zero(t.data_ptr(), n, stream()), "upnerf_zero")
    return t


def zeros(n: int, device) -> torch.Tensor:
    """n zeroed floats (1-D, fp32)."""
    n = int(n)
    if _S.rec is None or n <= 0 or not ENABLED:
        return torch.zeros(max(n, 0), device=device, dtype=torch.float32)
    first = not _S.rec
    _S.rec.append(n)
    if first and _S.plan is not None and _S.plan[0] == torch.device(device) == _S.device:
        _S.arena = _zeroed(sum(_padded(k) for k in _S.plan[1]), device)
    if _S.arena is not None:
        sizes = _S.plan[1]
        if _S.i < len(sizes) and sizes[_S.i] == n and torch.device(device) == _S.device:
            out = _S.arena[_S.cursor:_S.cursor + n]
            _S.cursor += _padded(n)
            _S.i += 1
            return out
        _S.arena = None  # the step left the plan: plain fills from here on, the next step follows the new sequence
    return torch.zeros(n, device=device, dtype=torch.float32)
