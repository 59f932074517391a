cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu -k "graph or trainer or optim or adam or tto or bench" 2>&1 | tail -5 > gpurun_out/t13.txt
for r in 1 2 3; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/t13.txt
done
