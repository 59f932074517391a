#!/bin/bash
# round 6: full GPU suite on the final sources, then everything under profiles/r06_* (tools/final_profile.sh)
out=gpurun_out/r6final; mkdir -p $out
timeout 2400 python -m pytest tests -q -m gpu -rA --durations=8 > $out/pytest.log 2>&1; tail -4 $out/pytest.log
grep -h "^\[masks\]" $out/pytest.log | sort | uniq > $out/masks.txt
timeout 600 python __graft_entry__.py smoke > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 600 python tools/rr_check.py --bwd > $out/rr_check.txt 2>&1; tail -1 $out/rr_check.txt
timeout 600 python tools/rr_sweep.py > $out/rr_sweep.txt 2>&1; tail -1 $out/rr_sweep.txt
timeout 2400 bash tools/final_profile.sh gpurun_out/final_r06 r06 > $out/final_profile.log 2>&1; tail -3 $out/final_profile.log
python tools/bytes_table.py gpurun_out/final_r06/pmc.json > gpurun_out/final_r06/bytes_table.md 2>&1
timeout 900 make -C upnerf_amd/csrc stamps > /dev/null 2>&1
timeout 300 python tools/stamps_field16.py > gpurun_out/final_r06/stamps_field16.txt 2>/dev/null
timeout 300 python tools/stamps_rr16.py > gpurun_out/final_r06/stamps_rr16.txt 2>/dev/null
python tools/show_bench.py gpurun_out/final_r06/bench.json 2>/dev/null | head -6
