cd /root/repo
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "tile_partial or ragged or stage_by_stage" 2>&1 | tail -8 > gpurun_out/t2.txt
for r in 1 2; do for tp in 0 1; do
  UPNERF_TILE_PARTIALS=$tp timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tp=$tp', round(d['value']), d['ms_per_step'])" >> gpurun_out/t2.txt
done; done
