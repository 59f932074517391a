#!/bin/bash
# Instruction mix and stall counters of the field kernels (two rocprofv3 PMC passes over a short bench run).
# usage (GPU box, repo root): bash tools/pmc_field.sh OUTDIR [extra bench args]   (--config trevi: the 256-sample register-resident kernels)
set -u
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES \
  --output-format csv -d "$OUT/a" -o a -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras "$@" > "$OUT.a.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/b" -o b -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extras "$@" > "$OUT.b.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "field16" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in acc:
    print(k[:70])
    for c in sorted(acc[k]): print(f"   {c:34s} {acc[k][c] / len(n[k][c]):14.4g}")
PY
rm -rf "$OUT/a" "$OUT/b"
