import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
print("host threads available:", os.cpu_count())
for n in (16, 32, 64, 128):
    torch.set_num_threads(n)
    v, sec = bench.cpu_baseline(0.3, 763, rays=768, iters=2)
    print(f"threads {n:4d}: {v:7.1f} rays/s ({sec:.1f} s per iteration)", flush=True)
