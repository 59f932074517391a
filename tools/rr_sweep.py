"""Register-resident fp16 kernels against the 64-sample tile kernels over field shapes the goldens do not have (depth, skip
position / none, no appearance or candidate embedding): outputs and every gradient.  GPU box, repo root: python tools/rr_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upnerf_amd import rendering as rd
from upnerf_amd.nerf import NeRF


def gen(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def l2(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


bad = 0
for tag, kw, mode, uc, ur in (("D=8 skip 4", dict(D=8, skips=[4]), 1, True, True), ("D=4 skip 2", dict(D=4, skips=[2]), 1, True, True),
                              ("D=6 no skip", dict(D=6, skips=[]), 1, True, True), ("D=2 skip 1", dict(D=2, skips=[1]), 1, True, True),
                              ("no appearance", dict(appearance_dim=0), 1, True, True), ("no candidate", dict(candidate_dim=0), 3, False, True),
                              ("no candidate, no colour yet", dict(candidate_dim=0), 3, False, False), ("D=1", dict(D=1, skips=[]), 1, True, True)):
    base = dict(D=8, W=256, skips=[4], feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
    base.update(kw)
    torch.manual_seed(5)
    try:
        model = NeRF("coarse", c2f=None, **base).cuda()
    except Exception as e:
        print(f"{tag:30s} construction refused: {type(e).__name__}: {e}")
        continue
    with torch.no_grad():
        for p in model.parameters():
            if p.dim() == 2:
                p.mul_(1.3)
    pk = model.packer
    R, S = 29, 70
    o = (gen((R, 3), 70) * 0.3).cuda(); d = torch.nn.functional.normalize(gen((R, 3), 71), dim=-1).cuda()
    z = (torch.sort(gen((R, S), 72).abs() * 3 + 0.1, dim=-1).values).cuda()
    c_rows = gen((R, 16), 73).cuda() if base["candidate_dim"] else None
    a_rows = gen((R, 48), 74).cuda() if base["appearance_dim"] else None
    res = {}
    try:
        for kt, rr in (("tile", 0), ("rr", 1)):
            rd.FIELD_MODE, rd.FIELD_RR = "f16", rr
            cfg = rd._PassCfg(pk, mode, uc, ur, [1.0] * 10, [1.0] * 4)
            leaves = [None if t is None else t.clone().requires_grad_(True) for t in (o, d, c_rows, a_rows, model.packed().detach())]
            outs = rd._FieldPass.apply(leaves[0], leaves[1], z, leaves[2], leaves[3], leaves[4], cfg)
            sum((t * gen(tuple(t.shape), 80 + i).cuda()).sum() for i, t in enumerate(outs) if t.numel() and t.requires_grad).backward()
            torch.cuda.synchronize()
            res[kt] = ([t.detach().cpu() for t in outs], [None if t is None or t.grad is None else t.grad.cpu() for t in leaves])
    except Exception as e:
        print(f"{tag:30s} {kt} raised {type(e).__name__}: {str(e)[:100]}")
        continue
    eo = max(l2(a, b) for a, b in zip(res["rr"][0], res["tile"][0]) if a.numel())
    eg = max(l2(a, b) for a, b in zip(res["rr"][1], res["tile"][1]) if a is not None and float(b.abs().max()) > 0)
    flag = "" if eo < 2e-3 and eg < 3e-2 else "   <-- BAD"
    bad += bool(flag)
    print(f"{tag:30s} outputs L2 {eo:.2e}  gradients L2 {eg:.2e}{flag}")
print("BAD" if bad else "OK", bad)
