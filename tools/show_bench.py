#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line: headline + per-kernel table."""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{d['value']:.0f} {d['unit']}  {d['ms_per_step']:.2f} ms/step  n_gpus={d['n_gpus']}")
ks = d.get("kernels", {})
tot = 0.0
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    tot += v["ms_per_step"]
    print(f"  {k:16s} {v['ms_per_step']:7.2f} ms/step  {v['launches_per_step']:5.1f} launches  avg {v['avg_ms']:.3f} ms"
          f"  {v.get('tflops_algorithmic', 0):6.1f} TF(alg)")
print(f"  timed kernels total {tot:.2f} ms/step")
if "roofline" in d:
    print("  roofline:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["roofline"].items() if k != "note"})
if "cpu_baseline" in d:
    print("  cpu_baseline:", d["cpu_baseline"])
