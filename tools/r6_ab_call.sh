#!/bin/bash
# A/B of library variants by kernel time (rocprofv3 stats of a short bench run per variant): bash tools/r6_ab_call.sh "<suffixes>" <kernel regex>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/abk; mkdir -p $O; : > $O/stats.txt
for lib in ${1:-"-"}; do
  sfx=$lib; [ "$lib" = "-" ] && sfx=""
  export UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$sfx.so
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$sfx -o p -- python3 bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench$sfx.log 2>&1
  f=$(find $O/prof$sfx -name '*kernel_stats.csv' | head -1)
  echo "== lib '$lib'" >> $O/stats.txt
  [ -n "$f" ] && grep -E "${2:-transient}" "$f" | cut -c1-200 >> $O/stats.txt
  rm -rf $O/prof$sfx
done
cat $O/stats.txt
