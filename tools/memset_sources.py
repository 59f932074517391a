#!/usr/bin/env python3
"""Who issues hipMemsetAsync / hipMemcpyAsync in one eager training step (a memset captured into a HIP graph lost its order
once): torch profiler with Python stacks, one line per call site."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda", 0)
s = bench.build_system(dev, 0.3)
bs = bench.make_batches(dev, 3, 100)
for i in range(2):
    s.training_step(bs[i], i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    s.training_step(bs[2], 2)
    torch.cuda.synchronize()
evs = list(prof.events())
cnt = collections.Counter()
for ev in evs:
    if ev.name in ("hipMemsetAsync", "hipMemcpyAsync", "hipMemcpyWithStream", "hipMemset", "hipMemcpy"):
        # innermost aten op that contains this runtime call
        best = None
        for p in evs:
            if p.name.startswith("aten::") and p.time_range.start <= ev.time_range.start and p.time_range.end >= ev.time_range.end:
                if best is None or p.time_range.start >= best.time_range.start:
                    best = p
        st = [x for x in ((best.stack if best is not None else ev.stack) or []) if "upnerf_amd" in x][:1]
        cnt[(ev.name, best.name if best is not None else "(library call)", st[0] if st else "")] += 1
for (api, op, where), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n:3d} x {api:18s} {op:28s} {where}")
