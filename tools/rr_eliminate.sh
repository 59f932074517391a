#!/bin/bash
# Elimination table + in-kernel stamps of the register-resident fp16 kernels.  Two halves:
#   bash tools/rr_eliminate.sh build          (anywhere hipcc runs: builds the variant libraries next to the shipped one)
#   bash tools/rr_eliminate.sh run OUT.txt    (GPU box, repo root: times the field kernels of every variant on the Trevi step)
# Every variant except `full` and `stamps` computes WRONG results by construction (-DUPNERF_EXPERIMENT): what is left of the
# launch time when one ingredient is compiled out bounds what that ingredient costs in the full kernel.
set -u
VARIANTS="noepi:-DRR_EXP_NOEPI nomma:-DRR_EXP_NOMMA nostore:-DRR_EXP_NOSTORE nodma:-DRR_EXP_NODMA nobarrier:-DRR_EXP_NOBARRIER"
if [ "$1" = build ]; then
  for v in $VARIANTS; do make -C upnerf_amd/csrc variant NAME=x${v%%:*} EXP=${v##*:} > /dev/null 2>&1 || echo "build of $v failed"; done
  make -C upnerf_amd/csrc stamps > /dev/null 2>&1 || echo "stamps build failed"
  ls -la upnerf_amd/libupnerf_hip_x*.so upnerf_amd/libupnerf_hip_stamps.so
  exit 0
fi
OUT=${2:-gpurun_out/rr_eliminate.txt}
{
  echo "# field-kernel launch times (HIP events, bench.py --config trevi, 8192 rays: 1 M coarse + 2 M fine samples) by variant"
  for v in full:- $VARIANTS; do
    n=${v%%:*}
    lib=upnerf_amd/libupnerf_hip_x$n.so; [ $n = full ] && lib=upnerf_amd/libupnerf_hip.so
    [ -f $lib ] || { echo "$n: no library"; continue; }
    UPNERF_LIB=$PWD/$lib timeout 300 python bench.py --config trevi --no-extras --no-cpu-baseline --no-configs34 --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print(f\"$n ({'${v##*:}'}): fwd {k['field_fwd']['avg_ms']:.3f} ms ({k['field_fwd']['tflops_algorithmic']:.0f} TF alg)  bwd {k['field_bwd']['avg_ms']:.3f} ms ({k['field_bwd']['tflops_algorithmic']:.0f} TF alg)\")"
  done
  echo
  echo "# in-kernel stamps (s_memtime; -DUPNERF_STAMPS build: slower than the shipped kernels by the stamps themselves)"
  UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_stamps.so timeout 300 python tools/stamps_rr16.py 2>&1 | grep -v amdgpu.ids
} > $OUT
cat $OUT
