#!/bin/bash
out=gpurun_out/r6rr; mkdir -p $out
UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_dmaall.so python tools/rr_check.py --bwd 2>&1 | tail -1
for r in 1 2 3; do for lib in _dmabuiltin "" _dmaall; do
UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$lib.so timeout 300 python bench.py --config trevi --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); print('lib[$lib]', round(d['value']), round(d['ms_per_step'],2), ' '.join(f'{n}={v[\"avg_ms\"]:.3f}' for n,v in k.items()))"
done; done | tee $out/ab_trevi2.txt
