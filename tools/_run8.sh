cd /root/repo
rm -f gpurun_out/t8.txt
for r in 1 2 3; do for lib in ntall ntall2; do
  UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_$lib.so timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/t8.txt
done; done
for lib in base ntall2; do
  UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_$lib.so timeout 300 python bench.py --config trevi --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-configs34 --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('trevi $lib', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/t8.txt
done
