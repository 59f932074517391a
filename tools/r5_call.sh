# scratch driver of the round's gpurun calls (rewritten per call)
mkdir -p gpurun_out/rrp
timeout 2400 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py -q -x -m gpu -k "rr or f16 or fp16 or frag" > gpurun_out/rrp/pytest.log 2>&1; tail -3 gpurun_out/rrp/pytest.log
timeout 600 python tools/rr_check.py > gpurun_out/rrp/rr_check.log 2>&1; tail -1 gpurun_out/rrp/rr_check.log
EXTRA="--config trevi" bash tools/r3_ab_libs.sh "_base - _base - _base - _base -" 64 1 2>&1 | tail -8
