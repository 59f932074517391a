import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/oracle", ROOT + "/tests"):
    sys.path.insert(0, p)
import numpy as np, torch
from golden_util import Case, orc
from test_hip_parity import build_system
c = Case("cfg2_phase0")
sysm = build_system(c)
batch = {k: v.cuda() for k, v in c.batch().items()}
loss, loss_d, res = sysm.compute_loss(batch, u_list=[u.clone() for u in c.u_list])
sysm._last_rays.retain_grad()
loss.backward()
g = sysm.se3_refine.weight.grad.cpu().numpy()
e = c.g["grad_se3_refine.weight"].reshape(g.shape)
np.set_printoptions(precision=4, suppress=False, linewidth=200)
print("idx", batch["img_idx"].cpu().numpy())
print("gpu\n", g)
print("gold\n", e)
# CPU pose backward fed with the GPU's ray gradients
gr = sysm._last_rays.grad.cpu()
st = c.state()
b = c.batch()
rows = st["se3_refine"][b["img_idx"]]
o, d = orc.get_rays(b["directions"], orc.compose_pair(orc.se3_exp(rows), b["c2w"]))
((o * gr[:, :3]).sum() + (d * gr[:, 3:6]).sum()).backward()
print("cpu pose-bwd of gpu ray grads\n", st["se3_refine"].grad.numpy())
