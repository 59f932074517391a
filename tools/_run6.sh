cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu -k "stage_by_stage or ragged" 2>&1 | tail -6 > gpurun_out/t6.txt
for r in 1 2; do for tp in 0 1; do
  UPNERF_VEC_FOLD=$tp timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fold=$tp', round(d['value']), d['ms_per_step'])" >> gpurun_out/t6.txt
done; done
