#!/bin/bash
out=gpurun_out/r6e; mkdir -p $out
timeout 300 tools/repro/pair_trunk_probe > $out/pair_trunk_probe.txt 2>&1; cat $out/pair_trunk_probe.txt
