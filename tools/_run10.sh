cd /root/repo
timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "fused_transient" 2>&1 | tail -12 > gpurun_out/t10.txt
rm -f gpurun_out/t8.txt
for r in 1 2 3; do for f in 0 1; do
  UPNERF_TRANSIENT_FUSED=$f timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/t10.txt
done; done
