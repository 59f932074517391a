# scratch driver for this round's GPU calls (edited per call)
set -u
mkdir -p gpurun_out/c10
(timeout 600 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "slice" 2>&1 | grep -v "^$" | tail -40) > gpurun_out/c10/slice.log 2>&1
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/c10/pytest.log 2>&1
for r in 1 2; do for lib in "" _ap1 _ap3; do UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$lib.so python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('brandenburg graph $lib', round(d['value']), round(d['ms_per_step'],3), ' '.join(f\"{n}={v['avg_ms']:.3f}\" for n,v in k.items()))" >> gpurun_out/c10/ab.log; done; done
cat gpurun_out/c10/slice.log; tail -6 gpurun_out/c10/pytest.log; cat gpurun_out/c10/ab.log
