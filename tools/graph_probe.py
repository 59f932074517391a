"""Feasibility probe: capture one whole training step (forward, backward, both Adam updates) in a HIP graph through
torch.cuda.graph and replay it; prints eager vs replay time per step and whether a replay reproduces the eager step.
Scalars that change per step (band weights, learning rates, bias corrections) are baked in here -- the probe answers
"does capture work with ctypes-launched kernels + autograd, and what does a replay cost"."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
sysm = bench.build_system(dev, float(sys.argv[1]) if len(sys.argv) > 1 else 0.3)
batches = bench.make_batches(dev, 4, 100)
static = {k: v.clone() for k, v in batches[0].items()}

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(5):
        sysm.training_step(static, i)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()

N = 30
t0 = time.perf_counter()
for i in range(N):
    for k in static:
        static[k].copy_(batches[i % 4][k])
    sysm.training_step(static, i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"eager : host {1e3 * (t1 - t0) / N:.2f} ms  wall {1e3 * (t2 - t0) / N:.2f} ms/step", flush=True)

g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    loss = sysm.training_step(static, 0)
torch.cuda.synchronize()
print(f"capture took {time.perf_counter() - t0:.2f} s; pool {torch.cuda.memory_reserved() / 2**30:.1f} GiB reserved", flush=True)
for i in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    for k in static:
        static[k].copy_(batches[i % 4][k])
    g.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"replay: host {1e3 * (t1 - t0) / N:.2f} ms  wall {1e3 * (t2 - t0) / N:.2f} ms/step  loss {float(loss):.6f}", flush=True)
