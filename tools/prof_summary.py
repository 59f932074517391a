"""Print a per-kernel summary of a rocprofv3 --kernel-trace --stats CSV directory: python tools/prof_summary.py DIR STEPS"""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.3f} ms/step over {steps} steps ({f})")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step "
          f"{float(r['AverageNs']) / 1e3:9.1f} us avg")
