#!/bin/bash
# A/B of library variants x tile sizes: kernel times of the field kernels (eager HIP events) and step rate
#   bash tools/r3_ab_libs.sh "lib1 lib2 ..." "tiles" [rounds]     (lib names are suffixes: "_ah2"; "-" = the shipped library)
out=gpurun_out/ab_libs; mkdir -p $out
for r in $(seq 1 ${3:-1}); do
  for lib in $1; do for t in $2; do
    sfx=$lib; [ "$lib" = "-" ] && sfx=""
    UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$sfx.so UPNERF_FIELD_TILE=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 > $out/b${sfx}_${t}_$r.json
    python - $out/b${sfx}_${t}_$r.json "$lib" $t <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=d.get('kernels',{})
print(f"lib '{sys.argv[2]}' tile {sys.argv[3]}: {round(d['value'])} rays/s {d['ms_per_step']:.2f} ms", ' '.join(f"{n}={v['avg_ms']:.3f}" for n,v in k.items()))
PY
  done; done
done
