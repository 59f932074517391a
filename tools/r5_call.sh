# scratch driver of the round's gpurun calls (rewritten per call)
mkdir -p gpurun_out/nf2
timeout 900 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k "feature_less" > gpurun_out/nf2/pytest.log 2>&1; tail -25 gpurun_out/nf2/pytest.log
