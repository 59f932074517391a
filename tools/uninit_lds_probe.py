#!/usr/bin/env python3
"""Locator for kernels whose results depend on what earlier kernels left in LDS: an eager training step is run with a kernel
in front of chosen C-ABI calls that fills the LDS of every CU with a bit pattern (tools/repro/poison_cu.hip); results that
move with the pattern come from a kernel that reads LDS it has not written.
    python tools/uninit_lds_probe.py [progress]"""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_trainer import _scene, _system  # noqa: E402
from test_sampler import _sampler  # noqa: E402
from upnerf_amd import _lib  # noqa: E402

P = ctypes.CDLL(os.path.join(ROOT, "tools", "repro", "libpoison_cu.so"))
P.poison_lds.argtypes = [ctypes.c_uint32, ctypes.c_void_p]
bufs, I = _scene(); smp = _sampler(bufs)
bs = list(smp.batches(128, seed=3, epoch=0))
PROGRESS = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
NAMES = [n for n in dir(_lib.lib) if n.startswith("upnerf_")] or []
TARGET, PATTERN, SEEN = [None], [0], {}


class Spy:
    def __init__(self, lib):
        object.__setattr__(self, "_l", lib)

    def __getattr__(self, name):
        f = getattr(self._l, name)
        if not name.startswith("upnerf_") or name in ("upnerf_abi_version",):
            return f

        def call(*a, **k):
            SEEN[name] = SEEN.get(name, 0) + 1
            if TARGET[0] in ("all", name):
                P.poison_lds(PATTERN[0], torch.cuda.current_stream().cuda_stream)
            return f(*a, **k)
        return call


spy = Spy(_lib.lib)
for name, m in list(sys.modules.items()):
    if name.startswith("upnerf_amd") and getattr(m, "lib", None) is _lib.lib:
        m.lib = spy


def run():
    torch.manual_seed(11)
    s = _system(I)
    s.set_progress(PROGRESS)
    loss, ld, res = s.compute_loss(bs[0])
    for o in s.optimizers():
        o.zero_grad()
    s.manual_backward(loss)
    torch.cuda.synchronize()
    out = {"res." + k: v.detach().clone() for k, v in res.items() if torch.is_tensor(v)}
    out["grads"] = torch.cat([p.grad.flatten() for p in s.parameters() if p.grad is not None]).clone()
    return out


c, c2 = run(), run()
print("clean twice identical:", all(torch.equal(c[k], c2[k]) for k in c), "| C-ABI entry points used:", len(SEEN))
for pat in (0x7F7F7F7F, 0x00000000, 0x7FC00000):
    PATTERN[0], TARGET[0] = pat, "all"
    p = run()
    bad = [k for k in c if not torch.equal(c[k], p[k])]
    print(f"LDS pattern {pat:#010x} in front of every call: differs in {bad[:6]}", flush=True)
    if bad:
        for name in sorted(SEEN):
            TARGET[0] = name
            q = run()
            b2 = [k for k in c if not torch.equal(c[k], q[k])]
            if b2:
                print(f"    in front of {name} only: {b2[:5]} nan={bool(torch.isnan(q['grads']).any())}", flush=True)
print("done")
