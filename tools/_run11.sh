cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu -k "tile_partial or joined_head" 2>&1 | tail -6 > gpurun_out/t11.txt
for r in; do for f in 0 1; do
  UPNERF_JOIN_HEADS=$f timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('join=$f', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/t11.txt
done; done
