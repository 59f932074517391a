# scratch driver of the round's gpurun calls (rewritten per call)
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests/test_hip_fullsize.py tests/test_hip_midsize.py tests/test_hip_parity.py -q -m gpu > gpurun_out/full/pytest.log 2>&1; tail -25 gpurun_out/full/pytest.log
