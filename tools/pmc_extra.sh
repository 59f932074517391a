#!/bin/bash
# Extra diagnostic PMC passes (vector-memory FIFO back-pressure, MFMA/VALU co-execution).  usage: bash tools/pmc_extra.sh OUTDIR
set -u
OUT=$(realpath -m "$1"); shift
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES \
  --output-format csv -d "$OUT/a" -o a -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > "$OUT.a.log" 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES \
  --output-format csv -d "$OUT/b" -o b -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > "$OUT.b.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "field16" not in k and "wgrad_f16x3_kernel<4, 4>" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in acc:
    print(k[:60])
    for c in sorted(acc[k]): print(f"   {c:34s} {acc[k][c] / len(n[k][c]):14.4g}")
PY
rm -rf "$OUT/a" "$OUT/b"
