# scratch driver of the round's gpurun calls (rewritten per call)
mkdir -p gpurun_out/prol
timeout 2400 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_hip_fullsize.py -q -x -m gpu > gpurun_out/prol/pytest.log 2>&1; tail -3 gpurun_out/prol/pytest.log
bash tools/r3_ab_libs.sh "_base - _base - _base -" 64 1 2>&1 | tail -6
