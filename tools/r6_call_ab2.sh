cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_hip_midsize.py -m gpu -x -q 2>&1 | tail -3
bash tools/r6_ab_call.sh "_base -" "transient|embed_bwd_grouped|ray_aux"
timeout 600 bash tools/r3_ab_libs.sh "_base -" "64" 3
