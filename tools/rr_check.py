"""Register-resident fp16 kernels (csrc/field16rr.hip) against the 64-sample tile-in-LDS kernels of the same fp16 mode and against
the fp32-MFMA kernels: every stored tensor of the forward pass (and, with --bwd, of the backward pass).  GPU box, repo root:
    python tools/rr_check.py [--bwd] [--R 37 --S 70] [--time]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=37); ap.add_argument("--S", type=int, default=70)
ap.add_argument("--mode", type=int, default=1); ap.add_argument("--bwd", action="store_true"); ap.add_argument("--time", action="store_true")
args = ap.parse_args()
from upnerf_amd import rendering as rd, synth
from upnerf_amd.nerf import NeRF


def gen(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


R, S, mode = args.R, args.S, args.mode
use_cand, use_rgb = mode in (0, 1), mode in (1, 2)
kw = dict(D=8, W=256, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
model = NeRF("coarse", c2f=None, **kw)
model.load_state_dict(synth.nerf_state("coarse", seed=3, **kw))
model = model.cuda()
pk = model.packer
o = (gen((R, 3), 70) * 0.3).cuda(); d = torch.nn.functional.normalize(gen((R, 3), 71), dim=-1).cuda()
z = (torch.sort(gen((R, S), 72).abs() * 3 + 0.1, dim=-1).values).cuda()
c_rows, a_rows = gen((R, 16), 73).cuda(), gen((R, 48), 74).cuda()
cfg = rd._PassCfg(pk, mode, use_cand, use_rgb, [1.0] * 10, [1.0] * 4)
res = {}
for tag, fm, rr in (("f32", "f32", 0), ("tile", "f16", 0), ("rr", "f16", 1)):
    rd.FIELD_MODE, rd.FIELD_RR = fm, rr
    leaves = [t.clone().requires_grad_(True) for t in (o, d, c_rows, a_rows, model.packed().detach())]
    outs = rd._FieldPass.apply(leaves[0], leaves[1], z, leaves[2], leaves[3], leaves[4], cfg)
    torch.cuda.synchronize()
    node = next(t.grad_fn for t in outs if t.grad_fn is not None)
    sv = node.saved
    M = R * S
    got = {k: sv[k].float().cpu() for k in ("x0", "e", "g1", "g2", "r1", "sigma_s", "sigma_c", "rgb") if sv.get(k) is not None}
    for k, k16, kexp in (("e", "e16", "eexp"), ("g2", "g2_16", "g2exp"), ("r1", "r1_16", "r1exp"), ("g1", "g1_16", "g1exp")):
        if sv.get(k) is None and sv.get(k16) is not None:
            got[k] = rd.dequant16(sv[k16][None], sv[kexp][None], frag=True)[0, :M].cpu()
    if sv.get("h16") is not None:
        got["h"] = rd.dequant16(sv["h16"], sv["hexp"], frag=bool(rr))[:, :M].cpu()
        got["h_last"] = sv["h"][0].cpu() if sv.get("h") is not None else got["h"][-1]
    else:
        got["h"] = sv["h"].cpu(); got["h_last"] = sv["h"][-1].cpu()
    got["outs"] = [t.detach().cpu() for t in outs]
    if args.bwd:
        sink = {}
        rd._DEBUG_SINK = sink
        sum((t * gen(tuple(t.shape), 80 + i).cuda()).sum() for i, t in enumerate(outs) if t.numel()).backward()
        rd._DEBUG_SINK = None
        torch.cuda.synchronize()
        got["sink"] = {k: v.float().cpu() for k, v in sink.items() if v is not None}
        got["grads"] = [t.grad.cpu() if t.grad is not None else None for t in leaves]
    res[tag] = got


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))


def l2(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


ref = res["f32"]
bad = 0
for k in ("x0", "sigma_s", "sigma_c", "rgb", "e", "g1", "g2", "r1", "h_last"):
    if k in ref:
        et, er = rel(res["tile"][k], ref[k]), rel(res["rr"][k], ref[k])
        flag = "" if er < max(1e-2, 3 * et) else "   <-- BAD"
        bad += bool(flag)
        print(f"{k:10s} tile {et:.2e}  rr {er:.2e}{flag}")
for l in range(8):
    et, er = rel(res["tile"]["h"][l], ref["h"][l]), rel(res["rr"]["h"][l], ref["h"][l])
    flag = "" if er < max(1e-2, 3 * et) else "   <-- BAD"
    bad += bool(flag)
    print(f"h{l:<9d} tile {et:.2e}  rr {er:.2e}{flag}")
for i, t in enumerate(ref["outs"]):
    if t.numel():
        et, er = rel(res["tile"]["outs"][i], t), rel(res["rr"]["outs"][i], t)
        flag = "" if er < max(1e-2, 3 * et) else "   <-- BAD"
        bad += bool(flag)
        print(f"out{i:<7d} tile {et:.2e}  rr {er:.2e}{flag}")
if args.bwd:
    for k in ref["sink"]:
        a, b, c = ref["sink"][k], res["tile"]["sink"][k], res["rr"]["sink"][k]
        if a.dim() == 3 and a.shape[0] == 8:
            for l in range(8):
                et, er = l2(b[l], a[l]), l2(c[l][: a.shape[1]], a[l])
                flag = "" if er < max(6e-2, 3 * et) else "   <-- BAD"
                bad += bool(flag)
                print(f"{k}[{l}]   L2: tile {et:.2e}  rr {er:.2e}{flag}")
        else:
            et, er = l2(b, a), l2(c[: a.shape[0]], a)
            flag = "" if er < max(6e-2, 3 * et) else "   <-- BAD"
            bad += bool(flag)
            print(f"{k:10s} L2: tile {et:.2e}  rr {er:.2e}{flag}")
    for i, g in enumerate(ref["grads"]):
        if g is not None:
            et, er = l2(res["tile"]["grads"][i], g), l2(res["rr"]["grads"][i], g)
            flag = "" if er < max(6e-2, 3 * et) else "   <-- BAD"
            bad += bool(flag)
            print(f"grad{i:<6d} L2: tile {et:.2e}  rr {er:.2e}{flag}")
print("BAD" if bad else "OK", bad)
if args.time:
    R2, S2 = 8192, 192
    o = (gen((R2, 3), 70) * 0.3).cuda(); d = torch.nn.functional.normalize(gen((R2, 3), 71), dim=-1).cuda()
    z = (torch.sort(gen((R2, S2), 72).abs() * 3 + 0.1, dim=-1).values).cuda()
    c_rows, a_rows = gen((R2, 16), 73).cuda(), gen((R2, 48), 74).cuda()
    for tag, rr in (("tile", 0), ("rr", 1), ("tile", 0), ("rr", 1)):
        rd.FIELD_MODE, rd.FIELD_RR = "f16", rr
        from upnerf_amd.ops import TIMER
        leaves = [t.clone().requires_grad_(True) for t in (o, d, c_rows, a_rows, model.packed().detach())]
        for it in range(3):
            outs = rd._FieldPass.apply(leaves[0], leaves[1], z, leaves[2], leaves[3], leaves[4], cfg)
            if args.bwd:
                sum(t.sum() for t in outs if t.numel()).backward()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = time.time(); ev[0].record()
        for it in range(5):
            outs = rd._FieldPass.apply(leaves[0], leaves[1], z, leaves[2], leaves[3], leaves[4], cfg)
            if args.bwd:
                sum(t.sum() for t in outs if t.numel()).backward()
        ev[1].record(); torch.cuda.synchronize()
        print(f"{tag}: {ev[0].elapsed_time(ev[1]) / 5:.3f} ms per pass (R {R2} S {S2}, fwd{'+bwd' if args.bwd else ''} incl. compositing / weight gradients)")
if "h_last" in res["rr"] and bad:
    a, b = res["rr"]["h_last"], ref["h_last"]
    d = (a - b).abs()
    print("h_last debug: rr nonzero frac", float((a != 0).float().mean()), "ref nonzero frac", float((b != 0).float().mean()))
    rows = d.max(dim=1).values
    print("rows with error > 1e-2 of max:", int((rows > 1e-2 * float(b.abs().max())).sum()), "of", rows.numel(), "first bad rows", torch.nonzero(rows > 1e-2 * float(b.abs().max()))[:8].flatten().tolist())
    cols = d.max(dim=0).values
    print("bad cols", torch.nonzero(cols > 1e-2 * float(b.abs().max()))[:16].flatten().tolist())
    print("ratio sample", (a[0, :8] / b[0, :8].clamp_min(1e-9)).tolist())
