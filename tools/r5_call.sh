# scratch driver of the round's gpurun calls (rewritten per call)
mkdir -p gpurun_out/heads
timeout 2400 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_hip_fullsize.py -q -x -m gpu > gpurun_out/heads/pytest.log 2>&1; tail -4 gpurun_out/heads/pytest.log
UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_stamps.so python tools/stamps_field16.py --heads 2>&1 | tail -10
bash tools/r3_ab_libs.sh "_base - _base - _base -" 64 1 2>&1 | tail -6
