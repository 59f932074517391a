"""Soak run: N training steps across the schedule phases (progress sweeps 0.08 -> 0.55), checking finiteness, memory
stability and step time.  python tools/soak.py [steps] [--graph]
--graph: through GraphedTrainingStep; the sweep changes the fine-sample split every few steps, so graphs are captured and
evicted all the time (the shared pool must not grow)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(args[0]) if args else 300
dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.08)
batches = bench.make_batches(dev, 8, 500)
hp = sysm.hparams
step = sysm.training_step
if "--graph" in sys.argv:
    from upnerf_amd.graph_step import GraphedTrainingStep
    step = GraphedTrainingStep(sysm)
t0 = time.perf_counter()
peak0 = None
for i in range(steps):
    prog = 0.08 + (0.55 - 0.08) * i / steps
    sysm.global_step = int(round(prog * 2 * hp["max_steps"]))
    sysm.set_progress(prog)
    loss = step(batches[i % len(batches)], i)
    if i % 50 == 49 or i == steps - 1:
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50 * 1e3
        t0 = time.perf_counter()
        ok = all(torch.isfinite(p).all() for p in sysm.parameters())
        print(f"step {i + 1:4d} progress {prog:.3f} sched {sysm.get_schedule_mult(prog):.3f} loss {float(loss):.5f} "
              f"{dt:6.2f} ms/step  mem {torch.cuda.memory_allocated() / 2**30:.2f} GiB reserved "
              f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB  finite={ok}" + (f"  {step.stats}" if hasattr(step, "stats") else ""))
        assert ok and torch.isfinite(loss)
