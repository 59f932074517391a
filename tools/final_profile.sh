#!/bin/bash
# Everything the round's profiles/ entries come from, in one GPU call (from the repo root on the GPU box):
#   bash tools/final_profile.sh gpurun_out/final r06      (second argument: the round tag tools/copy_profiles.sh will file the
#                                                          results under -- roofline.traffic_source cites profiles/<tag>_pmc.json)
# kernel trace + stats for both configs, the three PMC passes for both, pmc_current.json from the Brandenburg PMC run (so
# that the bench line that follows carries roofline.traffic measured on the sources it runs), both full bench lines.
set -u
OUT=$(realpath -m "${1:-gpurun_out/final}"); mkdir -p "$OUT"
TAG=${2:-r06}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o k -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing > "$OUT/prof.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_trevi" -o k -- python3 "$R/bench.py" --config trevi --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing > "$OUT/prof_trevi.log" 2>&1
cd "$R"
bash tools/pmc_collect.sh "$OUT/pmc" --no-extras > "$OUT/pmc.log" 2>&1
bash tools/pmc_collect.sh "$OUT/pmc_trevi" --no-extras --config trevi > "$OUT/pmc_trevi.log" 2>&1
python tools/pmc_current.py "$OUT/pmc.json" --field f16x3 --config brandenburg --progress 0.3 --name ${TAG}_pmc.json > "$OUT/pmc_current.log" 2>&1
cp profiles/pmc_current.json "$OUT/pmc_current.json"
python tools/pmc_current.py "$OUT/pmc_trevi.json" --field f16 --config trevi --progress 0.3 --out pmc_current_trevi.json --name ${TAG}_trevi_pmc.json > "$OUT/pmc_current_trevi.log" 2>&1
cp profiles/pmc_current_trevi.json "$OUT/pmc_current_trevi.json"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --config trevi > "$OUT/bench_trevi.json" 2> "$OUT/bench_trevi.err"
python tools/prof_summary.py "$OUT/prof" 67 45 > "$OUT/kernel_summary.txt" 2>&1
python tools/prof_summary.py "$OUT/prof_trevi" 67 45 > "$OUT/kernel_summary_trevi.txt" 2>&1
python tools/step_sequence.py "$OUT/prof" > "$OUT/step_sequence.txt" 2>&1
python tools/step_sequence.py "$OUT/prof_trevi" > "$OUT/step_sequence_trevi.txt" 2>&1
rm -f "$OUT/prof/"*kernel_trace.csv "$OUT/prof_trevi/"*kernel_trace.csv
ls "$OUT"
