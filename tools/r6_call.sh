#!/bin/bash
out=gpurun_out/r6j; mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_hip_midsize.py tests/test_hip_fullsize.py -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
bash tools/r3_ab_libs.sh "_nojoin -" "64" 3 2>&1 | tee $out/ab_join.txt
