"""Host time of each of the first training steps through GraphedTrainingStep (eager, capture, replays), with and without a
device synchronisation between steps: where one-off costs of a freshly instantiated graph land.  GPU box, repo root."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from upnerf_amd.graph_step import GraphedTrainingStep
dev = torch.device("cuda", 0)
for sync in (True, False):
    s = bench.build_system(dev, 0.3)
    b = bench.make_batches(dev, 4, 100)
    step = GraphedTrainingStep(s)
    out = []
    for i in range(14):
        if sync or i < 2:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(b[i % 4], i)
        out.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    print("sync between steps" if sync else "no sync after the capture", " ".join(f"{x:.2f}" for x in out), step.stats)
    del step, s
# the bench's own order of calls at --warmup 2 --steps 4 (step indices restart in every loop, four batches)
s = bench.build_system(dev, 0.3)
b = bench.make_batches(dev, 4, 100)
step = GraphedTrainingStep(s)
seq = [0, 1] + [0, 1] + [0, 1, 2, 3] + [0, 1, 2, 3]
out = []
for n, i in enumerate(seq):
    if n in (4, 8):
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(b[i % 4], i)
    out.append(1e3 * (time.perf_counter() - t0))
torch.cuda.synchronize()
print("bench order", " ".join(f"{x:.2f}" for x in out), step.stats)
