"""Run-to-run reproducibility of one field pass (forward + backward) on 65 536 samples: lists the backward buffers that
differ between identical runs, where, and compares against the fp32-kernel result.  UPNERF_LIB selects a diagnostic build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from upnerf_amd import synth, rendering as rd
from upnerf_amd.nerf import NeRF
from test_hip_kernels import gen
R, S = 1024, 64
kw = dict(D=8, W=256, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
model = NeRF("coarse", c2f=None, **kw); model.load_state_dict(synth.nerf_state("coarse", seed=3, **kw)); model = model.cuda()
pk = model.packer
o = (gen((R, 3), 70) * 0.3).cuda(); d = torch.nn.functional.normalize(gen((R, 3), 71), dim=-1).cuda()
z = (torch.sort(gen((R, S), 72).abs() * 3 + 0.1, dim=-1).values).cuda()
c_rows, a_rows = gen((R, 16), 73).cuda(), gen((R, 48), 74).cuda()
cfg = rd._PassCfg(pk, 1, True, True, [1.0] * 10, [1.0] * 4)
runs = []
for it in range(3):
    leaves = [t.clone().requires_grad_(True) for t in (o, d, c_rows, a_rows, model.packed().detach())]
    outs = rd._FieldPass.apply(leaves[0], leaves[1], z, leaves[2], leaves[3], leaves[4], cfg)
    sink = {}; rd._DEBUG_SINK = sink
    sum((t * gen(tuple(t.shape), 80 + i).cuda()).sum() for i, t in enumerate(outs) if t.numel()).backward()
    rd._DEBUG_SINK = None
    runs.append({k: v.clone() for k, v in sink.items() if v is not None})
a, b = runs[0], runs[1]
for k in a:
    if not torch.equal(a[k], b[k]):
        x, y = a[k], b[k]
        diff = (x != y)
        print(k, tuple(x.shape), "differing elements", int(diff.sum()), "max rel", float((x - y).abs().max() / x.abs().max()))
        if k == "gz_e":
            rows = diff.any(1).nonzero().flatten()
            cols = diff.any(0).nonzero().flatten()
            print("   rows:", rows[:40].tolist(), "... n =", rows.numel(), " tiles:", sorted(set((rows // 64).tolist()))[:20])
            print("   cols:", cols[:64].tolist(), "n =", cols.numel())
            r0 = int(rows[0]); print("   row", r0, "cols differing", diff[r0].nonzero().flatten()[:64].tolist())
            print("   vals", x[r0, diff[r0]][:6].tolist(), y[r0, diff[r0]][:6].tolist())

# ---- where do the wrong values come from?  reference = fp32-kernel result (deterministic)
rd.FIELD_MODE = "f32"
leaves = [t.clone().requires_grad_(True) for t in (o, d, c_rows, a_rows, model.packed().detach())]
outs = rd._FieldPass.apply(leaves[0], leaves[1], z, leaves[2], leaves[3], leaves[4], cfg)
sink = {}; rd._DEBUG_SINK = sink
sum((t * gen(tuple(t.shape), 80 + i).cuda()).sum() for i, t in enumerate(outs) if t.numel()).backward()
rd._DEBUG_SINK = None
for name in ("gz_e", "gz_h"):
    ref = sink[name].reshape(-1, 256)
    for ri, run in enumerate(runs[:2]):
        x = run[name].reshape(-1, 256)
        bad = ((x - ref).abs() > 1e-4 * ref.abs().max()).nonzero()
        print(name, "run", ri, "bad elements", bad.shape[0])
        for (r_, c_) in bad[:3].tolist():
            v = float(x[r_, c_])
            # search the same tile (64 rows) of the reference for this value
            t0 = (r_ // 64) * 64
            blk = ref[t0:t0 + 64]
            dist = (blk - v).abs()
            mn = dist.min(); pos = (dist == mn).nonzero()[0].tolist()
            print(f"   bad at row {r_} (tile row {r_ % 64}) col {c_}: got {v:.6g}, ref {float(ref[r_, c_]):.6g}; nearest ref value in tile at row {pos[0]} col {pos[1]} (|d|={float(mn):.2e})")
