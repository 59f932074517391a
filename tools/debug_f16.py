import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from upnerf_amd import synth, rendering as rd
from upnerf_amd.nerf import NeRF
from test_hip_kernels import gen, cpu

R, S, mode, use_cand, use_rgb = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (7, 40, 1, 1, 1)
kw = dict(D=8, W=256, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
model = NeRF("coarse", c2f=None, **kw)
model.load_state_dict(synth.nerf_state("coarse", seed=3, **kw))
model = model.cuda()
pk = model.packer
o = (gen((R, 3), 70) * 0.3).cuda()
d = torch.nn.functional.normalize(gen((R, 3), 71), dim=-1).cuda()
z = (torch.sort(gen((R, S), 72).abs() * 3 + 0.1, dim=-1).values).cuda()
scale = torch.logspace(-3, 2, R).reshape(R, 1)
if os.environ.get("NOSCALE"): scale = torch.ones(R, 1)
c_rows, a_rows = (gen((R, 16), 73) * scale).cuda(), (gen((R, 48), 74) * scale.flip(0)).cuda()
cfg = rd._PassCfg(pk, mode, bool(use_cand), bool(use_rgb), [1.0] * 10, [1.0] * 4)
res = {}
for fm in ("f32", "f16x3"):
    rd.FIELD_MODE = fm
    leaves = [t.clone().requires_grad_(True) for t in (o, d, c_rows, a_rows, model.packed().detach())]
    outs = rd._FieldPass.apply(leaves[0], leaves[1], z, leaves[2], leaves[3], leaves[4], cfg)
    sv = next(t.grad_fn for t in outs if t.grad_fn is not None).saved
    sink = {}
    rd._DEBUG_SINK = sink
    sum((t * gen(tuple(t.shape), 80 + i).cuda()).sum() for i, t in enumerate(outs) if t.numel()).backward()
    rd._DEBUG_SINK = None
    res[fm] = dict(sv={k: cpu(v) for k, v in sv.items() if torch.is_tensor(v)}, sink={k: cpu(v) for k, v in sink.items() if v is not None})
a, b = res["f32"], res["f16x3"]
M = R * S
for l in range(8):
    x, y = a["sink"]["gz_h"][l], b["sink"]["gz_h"][l]
    rowerr = (x - y).norm(dim=1) / (x.norm(dim=1) + 1e-30)
    print("gz_h", l, "rel", float((x - y).norm() / x.norm()), "worst rows", rowerr.topk(5).indices.tolist(), [f"{v:.2e}" for v in rowerr.topk(5).values.tolist()],
          "norm", float(x.norm()))
    hx, hy = a["sv"]["h"][l], b["sv"]["h"][l]
    flips = ((hx > 0) != (hy > 0)).sum().item()
    print("   h rel", float((hx - hy).norm() / hx.norm()), "sign flips", flips)
x, y = a["sink"]["gz_e"], b["sink"]["gz_e"]
print("gz_e rel", float((x - y).norm() / x.norm()))
x, y = a["sink"]["dxyz"], b["sink"]["dxyz"]
rowerr = (x - y).norm(dim=1) / (x.norm(dim=1) + 1e-30)
print("dxyz rel", float((x - y).norm() / x.norm()), rowerr.topk(5).indices.tolist())
