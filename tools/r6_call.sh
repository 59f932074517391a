#!/bin/bash
out=gpurun_out/r6s; mkdir -p $out
bash tools/r3_ab_libs.sh "- _stl1 _stl2" "64" 3 2>&1 | tee $out/ab_stl.txt
UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip_stl2.so timeout 900 python -m pytest tests/test_hip_kernels.py -q -x -m gpu -k "stage_by_stage or ragged or golden" 2>&1 | tail -3
