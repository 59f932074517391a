#!/bin/bash
# A/B of two builds of the library on the training step: per-kernel HIP-event times of the field kernels, alternating runs.
#   bash tools/ab_field.sh upnerf_amd/libupnerf_hip_base.so upnerf_amd/libupnerf_hip.so [rounds]
for r in $(seq 1 ${3:-3}); do
  for lib in "$1" "$2"; do
    UPNERF_LIB=$PWD/$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
print('$lib', round(d['value']), 'rays/s', ' '.join(f\"{n}={v['ms_per_step']:.3f}\" for n,v in k.items() if n.startswith('field')))"
  done
done
