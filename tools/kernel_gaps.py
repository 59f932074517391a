import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 2000 kernels (the exchange leg's timed steps)
rows = rows[-1500:]
prev_end = None
out = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev_end is not None:
        gap = (s - prev_end) / 1e3
        if gap > 8:
            out.append((gap, prev_name[:50], r['Kernel_Name'][:60]))
    prev_end, prev_name = max(e, prev_end or 0), r['Kernel_Name']
from collections import Counter, defaultdict
d = defaultdict(list)
for g, a, b in out:
    d[(a, b)].append(g)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{len(v):3d} x {sum(v)/len(v):7.1f} us  after {k[0]}  before {k[1]}")
# ATen kernels (only the exchange leg has any inside its timed steps): per launch and per step
from collections import defaultdict as _dd
agg = _dd(list)
for r in rows:
    n = r['Kernel_Name']
    if 'at::native' in n or 'rccl' in n.lower() or 'nccl' in n.lower() or 'copyBuffer' in n:
        agg[n[:110]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
nsteps = sum(1 for r in rows if 'loss_finish_kernel' in r['Kernel_Name'])
print('steps in window', nsteps)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{len(v)/max(nsteps,1):5.1f}/step x {sum(v)/len(v):7.1f} us  {k}")
tot = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) for r in rows) / 1e3 / max(nsteps, 1)
print(f"kernel time per step in the window: {tot:.1f} us")
