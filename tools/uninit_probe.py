#!/usr/bin/env python3
"""Locator for reads of uninitialised device memory on the training step: every torch.empty / empty_like / new_empty issued
from the package is filled with a chosen bit pattern, one call site at a time, and the step's gradients are compared with
the all-zero-fill run.  A site whose fill value changes the result is read before it is written."""
import os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_trainer import _scene, _system  # noqa: E402
from test_sampler import _sampler  # noqa: E402

bufs, I = _scene(); smp = _sampler(bufs)
bs = list(smp.batches(128, seed=3, epoch=0))
PROGRESS = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
real_empty, real_empty_like = torch.empty, torch.empty_like
SITES, TARGET, FILL = {}, [None], [0.0]


def site():
    for f in reversed(traceback.extract_stack(limit=8)[:-2]):
        if "upnerf_amd" in f.filename:
            return f"{os.path.basename(f.filename)}:{f.lineno}"
    return None


def fill(t):
    s = site()
    if s is None or not t.is_cuda or t.numel() == 0:
        return t
    SITES[s] = SITES.get(s, 0) + 1
    v = FILL[0] if (TARGET[0] is None or TARGET[0] == s) else 0.0
    if t.dtype.is_floating_point:
        t.fill_(v)
    else:
        t.fill_(0 if v == 0.0 else 0x3F3F3F3F if t.dtype in (torch.int32, torch.int64) else 63)
    return t


torch.empty = lambda *a, **k: fill(real_empty(*a, **k))
torch.empty_like = lambda *a, **k: fill(real_empty_like(*a, **k))


def run():
    torch.manual_seed(11)
    s = _system(I)
    s.set_progress(PROGRESS)
    loss, _, _ = s.compute_loss(bs[0])
    for o in s.optimizers():
        o.zero_grad()
    s.manual_backward(loss)
    torch.cuda.synchronize()
    return float(loss), torch.cat([p.grad.flatten() for p in s.parameters() if p.grad is not None]).clone()


l0, g0 = run()
l0b, g0b = run()
print("zero fill twice identical:", l0 == l0b and torch.equal(g0, g0b), "sites:", len(SITES))
for val in (3.0e38, float("nan")):
    FILL[0] = val
    for s in sorted(SITES):
        TARGET[0] = s
        l, g = run()
        if l != l0 or not torch.equal(g, g0):
            bad = g != g0
            print(f"fill {val}: site {s} is read before written: loss {l0} -> {l}, {int(bad.sum())} gradient entries differ, "
                  f"nan {int(torch.isnan(g).sum())}", flush=True)
from upnerf_amd import ops  # noqa: E402
TARGET[0], FILL[0] = "none", 0.0
for key, t in ops._WS.items():
    for val in (3.0e38, float("nan")):
        t.fill_(val)
        l, g = run()
        if l != l0 or not torch.equal(g, g0):
            print(f"workspace {key} ({t.numel()} floats) filled with {val}: {int((g != g0).sum())} gradient entries differ, nan {int(torch.isnan(g).sum())}", flush=True)
        else:
            print(f"workspace {key} filled with {val}: no effect")
print("done")
