#!/bin/bash
out=gpurun_out/r6g; mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_graph_step.py tests/test_trainer.py tests/test_checkpoint.py -q -x -m gpu > $out/pytest.log 2>&1; tail -15 $out/pytest.log
for i in 1 2; do
UPNERF_FUSE_RESAMPLE=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('unfused', round(d['value']), d['ms_per_step'])"
timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused  ', round(d['value']), d['ms_per_step'])"
done
