"""Soak run: N training steps across the schedule phases (progress sweeps 0.08 -> 0.55), checking finiteness, memory
stability and step time.  python tools/soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.08)
batches = bench.make_batches(dev, 8, 500)
hp = sysm.hparams
t0 = time.perf_counter()
peak0 = None
for i in range(steps):
    prog = 0.08 + (0.55 - 0.08) * i / steps
    sysm.global_step = int(round(prog * 2 * hp["max_steps"]))
    sysm.set_progress(prog)
    loss = sysm.training_step(batches[i % len(batches)], i)
    if i % 50 == 49 or i == steps - 1:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50 * 1e3
        t0 = time.perf_counter()
        ok = all(torch.isfinite(p).all() for p in sysm.parameters())
        print(f"step {i + 1:4d} progress {prog:.3f} sched {sysm.get_schedule_mult(prog):.3f} loss {float(loss):.5f} "
              f"{dt:6.2f} ms/step  mem {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved "
              f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB  finite={ok}")
        assert ok and torch.isfinite(loss)
