#!/bin/bash
out=gpurun_out/r6w; mkdir -p $out
for v in _wpsleep; do
echo "== lib$v"; UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$v.so python tools/bench_wgrad_planes.py 786432 2>&1 | grep "us per launch"
UPNERF_LIB=$PWD/upnerf_amd/libupnerf_hip$v.so python tools/bench_wgrad_planes.py 262144 2>&1 | grep "us per launch"
done | tee $out/planes_eliminate.txt
