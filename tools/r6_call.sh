#!/bin/bash
# round-6 GPU driver (rewritten per call; the last content is the final profile run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cr; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "parameter_gradients_match_the_oracle" --durations=3 2>&1 | tail -12 > $O/fulltest.txt
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "composit" 2>&1 | tail -3 > $O/tests.txt
for lib in _cr1 "" _cr8; do
  export UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$lib.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$lib -o p -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench$lib.log 2>&1
  f=$(find $O/prof$lib -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && grep -E "composite" "$f" | cut -c1-200 > $O/stats$lib.txt
  tail -1 $O/bench$lib.log | cut -c1-120 >> $O/stats$lib.txt
  rm -rf $O/prof$lib
done
unset UPNERF_LIB
timeout 600 bash tools/r3_ab_libs.sh "_cr1 - _cr8" "64" 2 > $O/ab.txt 2>&1
cat $O/fulltest.txt $O/tests.txt $O/stats_cr1.txt $O/stats.txt $O/stats_cr8.txt $O/ab.txt
