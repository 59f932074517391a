"""A/B of an environment switch on the training step inside ONE process image per run, alternating runs:
    python tools/ab_env.py UPNERF_FUSED_BLEND 0 1 [rounds]
prints host issue time and wall time per step (no kernel timing, 60 steps) for each value."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
var, vals, rounds = sys.argv[1], sys.argv[2:4], int(sys.argv[4]) if len(sys.argv) > 4 else 3
child = r'''
import sys, time, torch
sys.path.insert(0, %r)
import bench
dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.3)
b = bench.make_batches(dev, 4, 100)
for i in range(10): sysm.training_step(b[i %% 4], i)
torch.cuda.synchronize()
N = 60
t0 = time.perf_counter()
for i in range(N): sysm.training_step(b[i %% 4], i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host %%.2f ms  wall %%.2f ms/step" %% (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N))
''' % ROOT
for r in range(rounds):
    for v in vals:
        env = dict(os.environ, **{var: v})
        out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
        print(f"{var}={v}:", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
