#!/bin/bash
out=gpurun_out/r6p; mkdir -p $out
timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "wgrad or weight_grad or chained or slab or joined_head" 2>&1 | tail -2
bash tools/r3_ab_libs.sh "_nopin -" "64" 3 2>&1 | tee $out/ab_pin.txt
for r in 1 2 3; do for lib in _nopin ""; do
UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$lib.so timeout 300 python bench.py --config trevi --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('trevi lib[$lib]', round(d['value']), round(d['ms_per_step'],2))"
done; done | tee $out/ab_pin_trevi.txt
