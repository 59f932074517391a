set -u
mkdir -p gpurun_out/c15
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > gpurun_out/c15/pytest.log 2>&1
for r in 1 2; do python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; print('brandenburg graph', round(d['value']), round(d['ms_per_step'],3), ' '.join(f\"{n}={v['avg_ms']:.3f}\" for n,v in k.items()))" >> gpurun_out/c15/ab.log; done
tail -5 gpurun_out/c15/pytest.log; cat gpurun_out/c15/ab.log
