# driver of the round's final GPU call: everything under profiles/r05_* comes from this run (tools/final_profile.sh) + the stamps build
set -u
bash tools/final_profile.sh gpurun_out/final_r05 > gpurun_out/final_r05.log 2>&1
UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_stamps.so python tools/stamps_field16.py > gpurun_out/final_r05/stamps_field16.txt 2>&1
python tools/overlap_probe.py > gpurun_out/final_r05/overlap_probe.txt 2>&1
tail -3 gpurun_out/final_r05.log; python tools/show_bench.py gpurun_out/final_r05/bench.json 2>/dev/null | head -8
