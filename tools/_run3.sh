cd /root/repo
echo skip-tests > gpurun_out/t3.txt
for r in 1 2 3; do for tp in 0 1; do
  UPNERF_VEC_FOLD=$tp timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fold=$tp', round(d['value']), d['ms_per_step'], {k:round(v['avg_ms'],4) for k,v in d.get('kernels',{}).items() if 'wgrad16' in k and '256x256' in k})" >> gpurun_out/t3.txt
done; done
