import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.3)
batches = bench.make_batches(dev, 2, 100)
for i in range(4): sysm.training_step(batches[i % 2], i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    for i in range(5): sysm.training_step(batches[i % 2], i)
torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.key[:60]:60s} {e.count / 5:7.1f}/step  cpu {e.self_cpu_time_total / 5 / 1e3:7.3f} ms/step")
