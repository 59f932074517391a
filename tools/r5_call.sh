# driver of the round's final GPU call: everything under profiles/r05_* comes from this run (tools/final_profile.sh); the stamps
# build and the overlap probe (field kernels unchanged since) are from the earlier call of the round
set -u
bash tools/final_profile.sh gpurun_out/final_r05 > gpurun_out/final_r05.log 2>&1
tail -3 gpurun_out/final_r05.log; python tools/show_bench.py gpurun_out/final_r05/bench.json 2>/dev/null | head -8
