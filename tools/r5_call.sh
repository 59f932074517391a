# scratch driver for this round's GPU calls (edited per call)
set -u
mkdir -p gpurun_out/c4
(timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -m gpu -x -q -k "wgrad or chain or slab or tto or stage_by_stage or option" 2>&1 | tail -15) > gpurun_out/c4/pytest.log 2>&1
bash tools/ab_lib.sh upnerf_amd/libupnerf_hip_base.so upnerf_amd/libupnerf_hip.so "wgrad_256x256 wgrad_256x64 wgrad_128x128 field_fwd field_bwd" 2 > gpurun_out/c4/ab.log 2>&1
for lib in _base ""; do UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$lib.so python bench.py --config trevi --steps 10 --warmup 3 --no-extras --no-cpu-baseline --kernel-timing all 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
print('trevi $lib', round(d['value']), 'rays/s', ' '.join(f\"{n}={v['ms_per_step']:.3f}\" for n,v in k.items() if v['ms_per_step']>0.15))" >> gpurun_out/c4/ab.log; done
for lib in _base ""; do UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$lib.so python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('brandenburg graph $lib', round(d['value']), d['ms_per_step'])" >> gpurun_out/c4/ab.log; done
tail -5 gpurun_out/c4/pytest.log; cat gpurun_out/c4/ab.log
