#!/usr/bin/env python3
"""ATen operators executed by one eager training step, by the line of the package that called them (TorchDispatchMode +
Python stack; autograd-engine calls show up as "(backward)"): where the small-launch tail of the step comes from."""
import os, sys, collections, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

VIEWS = ("view", "reshape", "expand", "slice", "select", "t.", "transpose", "permute", "detach", "alias", "unsqueeze", "squeeze",
         "as_strided", "_unsafe_view", "split", "unbind", "narrow", "is_", "size", "stride", "lift_fresh", "empty", "_local_scalar")
cnt = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            site = "(backward / torch internals)"
            for f in reversed(traceback.extract_stack(limit=14)[:-1]):
                if "upnerf_amd" in f.filename:
                    site = f"{os.path.basename(f.filename)}:{f.lineno}"
                    break
            cnt[(site, name.replace("aten.", ""))] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda", 0)
s = bench.build_system(dev, 0.3)
bs = bench.make_batches(dev, 3, 100)
for i in range(2):
    s.training_step(bs[i], i)
torch.cuda.synchronize()
with Log():
    s.training_step(bs[2], 2)
torch.cuda.synchronize()
print("non-view ATen operators in the step:", sum(cnt.values()))
bysite = collections.Counter()
for (site, op), n in cnt.items():
    bysite[site] += n
for site, n in bysite.most_common(40):
    print(f"{n:4d}  {site:34s}", ", ".join(f"{op} x{m}" for (s2, op), m in sorted(cnt.items(), key=lambda kv: -kv[1]) if s2 == site)[:150])
