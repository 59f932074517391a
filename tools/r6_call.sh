#!/bin/bash
# round-6 GPU driver (rewritten per call; the last content is the final profile run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lin
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_trainer.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/lin/tests.txt
bash tools/r3_ab_libs.sh "- _oldlin" "64" 3 > gpurun_out/lin/ab.txt 2>&1
for lib in "" _oldlin; do
  export UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$lib.so
  rocprofv3 --kernel-trace --stats -d gpurun_out/lin/prof$lib -o p -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/lin/bench$lib.log 2>&1
  f=$(find gpurun_out/lin/prof$lib -name '*kernel_stats.csv' | head -1)
  grep -E "linear_kernel|matvec|Name" $f | cut -c1-220 > gpurun_out/lin/stats$lib.txt
done
cat gpurun_out/lin/tests.txt gpurun_out/lin/ab.txt gpurun_out/lin/stats.txt gpurun_out/lin/stats_oldlin.txt
