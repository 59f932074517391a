#!/bin/bash
# A/B of the 64-row / 128-row tilings of the field kernels on the training step (alternating runs, one process each)
out=gpurun_out/ab_tile; mkdir -p $out
for r in $(seq 1 ${1:-2}); do
  for t in 64 128; do
    UPNERF_FIELD_TILE=$t python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline ${2:-} 2>/dev/null | tail -1 > $out/b_${t}_$r.json
    python - $out/b_${t}_$r.json $t <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d.get('kernels',{})
print('tile', sys.argv[2], round(d['value']), 'rays/s', f"{d['ms_per_step']:.2f} ms", ' '.join(f"{n}={v['avg_ms']:.3f}" for n,v in k.items()))
PY
  done
done
