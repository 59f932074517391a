# scratch driver for this round's GPU calls (edited per call)
set -u
mkdir -p gpurun_out/c12
(timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_midsize.py -m gpu -q -k "riding or wgrad or chain or slab or f16 or fp16 or trevi" 2>&1 | tail -12) > gpurun_out/c12/pytest.log 2>&1
for r in 1 2; do for ride in 0 1; do UPNERF_VEC_RIDE=$ride python bench.py --config trevi --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('trevi graph ride=$ride', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/c12/ab.log; done; done
tail -6 gpurun_out/c12/pytest.log; cat gpurun_out/c12/ab.log
