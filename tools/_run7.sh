cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/t7.txt
for r in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'])" >> gpurun_out/t7.txt
done
