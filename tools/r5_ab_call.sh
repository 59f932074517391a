#!/bin/bash
# round 5: parity of the generic GEMM tests, then base-vs-new library step rate alternating on one box
#   bash tools/r5_ab_call.sh "<pytest -k expression>" [rounds]
out=gpurun_out/ab5; mkdir -p $out
timeout 1200 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "$1" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
one() {
  UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$2.so timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 > $out/b_$1.json
  python - $out/b_$1.json $1 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), 'rays/s', round(d['ms_per_step'],3), 'ms')
PY
}
for r in $(seq 1 ${2:-3}); do one old$r _base; one new$r ""; done
EXTRA="--config trevi"; for r in 1 2; do one told$r _base; one tnew$r ""; done
