#!/bin/bash
# Engine clock / power while the training step runs (is the MFMA peak in MI355X_MICROARCH.md reachable at this clock?).
#   bash tools/clocks_under_load.sh [steps]
python bench.py --steps ${1:-400} --warmup 10 --no-cpu-baseline > gpurun_out/clk_bench.json 2>/dev/null &
pid=$!
sleep 25
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | tr -s ' ' | head -6
  echo ---
  sleep 1
done
wait $pid
tail -c 400 gpurun_out/clk_bench.json
