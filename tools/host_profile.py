"""Host-side (Python / launch) cost of one training step: cProfile over N steps without device syncs inside."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.3)
batches = bench.make_batches(dev, 4, 100)
for i in range(5):
    sysm.training_step(batches[i % 4], i)
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for i in range(N):
    sysm.training_step(batches[i % 4], i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time {1e3 * (t1 - t0) / N:.2f} ms/step, wall incl. drain {1e3 * (t2 - t0) / N:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    sysm.training_step(batches[i % 4], i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
