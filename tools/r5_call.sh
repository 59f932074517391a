# driver of the round's final GPU call: everything under profiles/r05_* comes from this run -- tools/final_profile.sh, then the three
# stamps builds of the field kernels (make -C upnerf_amd/csrc stamps [EXP=-DUPNERF_STAMPS_HEADS | -DUPNERF_STAMPS_BHEADS], copied to
# libupnerf_hip_stamps_{plain,h,b}.so).  The overlap probe (weight-gradient kernels unchanged since) is from the earlier call.
set -u
mkdir -p gpurun_out/final_r05
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/final_r05/pytest.log 2>&1; tail -3 gpurun_out/final_r05/pytest.log
bash tools/final_profile.sh gpurun_out/final_r05 > gpurun_out/final_r05.log 2>&1
{ UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_stamps_plain.so python tools/stamps_field16.py 2>&1 | grep -v amdgpu.ids
  UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_stamps_h.so python tools/stamps_field16.py --heads 2>&1 | grep -A12 "head stage"
  UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_stamps_b.so python tools/stamps_field16.py --bheads 2>&1 | grep -A12 "head stage"; } > gpurun_out/final_r05/stamps_field16.txt
tail -3 gpurun_out/final_r05.log; python tools/show_bench.py gpurun_out/final_r05/bench.json 2>/dev/null | head -8
