import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
from golden_util import Case, orc, rel_err
from test_hip_parity import build_system, oracle_grads
name = sys.argv[1] if len(sys.argv) > 1 else "small_tto"
c = Case(name)
sysm = build_system(c)
batch = {k: v.cuda() for k, v in c.batch().items()}
keep = {}
loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list], keep=keep)
loss.backward()
exp = c.expected_results()
print("fwd rel errs:", {k: f"{rel_err(res[k].detach().cpu().numpy().reshape(v.shape), v):.1e}" for k, v in exp.items()})
g32, ok32 = oracle_grads(c, torch.float32)
print("z_coarse max diff", float((keep["z_coarse"].cpu() - ok32["z_coarse"]).abs().max()))
dz = (keep["z_fine"].cpu() - ok32["z_fine"]).abs()
print("z_fine max diff", float(dz.max()), "mean", float(dz.mean()), "n>1e-6", int((dz > 1e-6).sum()), "of", dz.numel())
gz32, _ = oracle_grads(c, torch.float32, z_fine=keep["z_fine"].cpu())
gz64, _ = oracle_grads(c, torch.float64, z_fine=keep["z_fine"].cpu())
got = {n: p.grad for n, p in sysm.named_parameters()}
rows = []
for n, r in g32.items():
    if r is None or n.endswith("progress") or got.get(n) is None: continue
    g = got[n].detach().cpu().double()
    f = lambda a: float((g - a.double()).abs().max() / max(float(a.abs().max()), 1e-30))
    rows.append((max(f(r), f(gz32[n])), n, f(r), f(gz32[n]), f(gz64[n]), float((gz32[n].double()-gz64[n]).abs().max()/max(float(gz64[n].abs().max()),1e-30))))
rows.sort(reverse=True)
print("name | vs oracle32 | vs oracle32@gpu-z | vs oracle64@gpu-z | oracle32-vs-64@gpu-z")
for r in rows[:14]:
    print(f"{r[1]:45s} {r[2]:.1e} {r[3]:.1e} {r[4]:.1e} {r[5]:.1e}")
