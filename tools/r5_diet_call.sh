#!/bin/bash
# round 5, launch diet: parity of the touched pieces, then old-vs-new step rate alternating on one box, then the launch count
out=gpurun_out/diet; mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_graph_step.py -q -x -m gpu > $out/pytest.log 2>&1; tail -4 $out/pytest.log
one() {  # label, knobs on / off
  UPNERF_ZERO_POOL=$2 UPNERF_JOIN_RAYS=$2 UPNERF_EMBED_PREFETCH=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline ${EXTRA:-} 2>/dev/null | tail -1 > $out/b_$1.json
  python - $out/b_$1.json $1 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), 'rays/s', round(d['ms_per_step'],3), 'ms', d.get('launches_per_step'))
PY
}
for r in 1 2 3; do one old$r 0; one new$r 1; done
EXTRA="--config trevi" ; for r in 1 2; do one told$r 0; one tnew$r 1; done
R=$(pwd); O=$(realpath -m $out)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o k -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing > "$O/prof.log" 2>&1
cd "$R"
python tools/step_sequence.py "$O/prof" > "$O/step_sequence.txt" 2>&1
python tools/prof_summary.py "$O/prof" 67 60 > "$O/kernel_summary.txt" 2>&1
rm -f "$O/prof/"*kernel_trace.csv
head -1 $O/step_sequence.txt
