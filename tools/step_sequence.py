"""Kernel launch sequence of ONE replayed training step from a rocprofv3 --kernel-trace CSV (consecutive repeats folded):
    python tools/step_sequence.py DIR [marker-substring]
The step is cut at the marker kernel (default: the set_scalars kernel that opens every replayed step)."""
import csv, glob, re, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "set_scalars_kernel"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = starts[-3], starts[-2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:95]
seq, tot = [], 0.0
for r in rows[a:b]:
    n, dur = short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    if seq and seq[-1][0] == n:
        seq[-1][1] += 1; seq[-1][2] += dur
    else:
        seq.append([n, 1, dur])
print(f"{b - a} launches, {tot / 1e3:.3f} ms of kernel time, wall {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6:.3f} ms")
for n, c, dur in seq:
    print(f"{c:3d} x {dur / c:8.1f} us  {n}")
