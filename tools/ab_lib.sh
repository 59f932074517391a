#!/bin/bash
# A/B of two builds of the library on the training step with the full per-class kernel timing, alternating runs; prints the
# classes named on the command line.   bash tools/ab_lib.sh base.so new.so "composite_fwd composite_bwd" [rounds]
for r in $(seq 1 ${4:-3}); do
  for lib in "$1" "$2"; do
    UPNERF_LIB=$PWD/$lib python bench.py --steps 30 --warmup 8 --no-cpu-baseline --kernel-timing all 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
print('$lib', round(d['value']), 'rays/s', ' '.join(f\"{n}={k[n]['ms_per_step']:.3f}\" for n in '$3'.split() if n in k))"
  done
done
