#!/bin/bash
# round 6: GPU parity run (full -m gpu suite; the widened-gate report lands in gpurun_out/parity_widened.txt) + one short bench line
out=gpurun_out/r6a; mkdir -p $out
timeout 1500 python -m pytest tests -q -m gpu -rA > $out/pytest.log 2>&1; tail -5 $out/pytest.log
grep -h "^\[masks\]\|^\[widened\]" $out/pytest.log | sort | uniq > $out/masks_widened.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>$out/bench.err | tail -1 > $out/bench_short.json
python tools/show_bench.py $out/bench_short.json 2>/dev/null | head -20
