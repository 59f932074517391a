#!/bin/bash
# Three rocprofv3 PMC passes over a short bench run (SQ/GRBM counters; FETCH_SIZE; WRITE_SIZE do not fit one pass).
# usage (on the GPU box, from the repo root): bash tools/pmc_collect.sh gpurun_out/pmc_X [extra bench args]
set -u
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
  --output-format csv -d "$OUT/a" -o a -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > "$OUT.a.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/b" -o b -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > "$OUT.b.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/c" -o c -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > "$OUT.c.log" 2>&1
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT.json" "$OUT.md" "$OUT/a" "$OUT/b" "$OUT/c"
# the raw per-dispatch CSVs are large: keep only the summaries
rm -rf "$OUT/a" "$OUT/b" "$OUT/c"
