#!/bin/bash
# A/B of HIP runtime switches on the graph-replayed training step: bash tools/ab_graph_env.sh "VAR=val" "VAR2=val" ...
# ("-" = no switch); prints rays/s, ms per step, host issue per step for each, two rounds, alternating.
for r in 1 2; do
  for kv in "$@"; do
    if [ "$kv" = "-" ]; then E=""; else E="$kv"; fi
    env $E python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$kv', round(d['value']), round(d['ms_per_step'], 2), round(d['host_issue_ms_per_step'], 2))" || echo "$kv FAILED"
  done
done
