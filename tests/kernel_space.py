"""Torch (CPU) restatement of what the fused HIP kernels compute FROM THE PACKED BUFFERS, stage by stage, with every
intermediate exposed.  Test infrastructure only: it lets the GPU tests say which stage of a fused kernel is wrong
(x0, h_l, e, g1, g2, r1, sigma, rgb, the per-ray sums, and -- through autograd -- every pre-activation gradient),
and it lets a CPU test prove that packing + folding + composite-then-project reproduce the oracle."""
import torch
import torch.nn.functional as F

X0, AUXK, CK = 64, 80, 16


def mat(P, off, n, k):
    return P[off:off + n * k].view(n, k)


def posenc_w(x, L, wk):
    freq = (2 ** torch.arange(L, dtype=torch.float32)).to(x.dtype) * torch.pi
    arg = x[..., None] * freq
    enc = torch.stack([arg.sin(), arg.cos()], dim=-2) * torch.as_tensor(wk, dtype=x.dtype)
    return torch.cat([x, enc.reshape(*x.shape[:-1], -1)], -1)


def ray_aux(rays_d, a_rows, wk_dir):
    R = rays_d.shape[0]
    pe = posenc_w(rays_d, 4, wk_dir)
    a = a_rows if a_rows is not None else torch.zeros(R, 48, dtype=rays_d.dtype)
    return torch.cat([pe, a, torch.zeros(R, AUXK - 75, dtype=rays_d.dtype)], 1)


def field(P, pk, rays_o, rays_d, z, c_rows, aux, wk_xyz, use_cand, use_rgb, keep_pre=False):
    """Returns dict with x0 [M,64], h [D][M,W], e, g1, g2, r1, sigma_s, sigma_c, rgb and (keep_pre) the
    pre-activation tensors with retain_grad() set."""
    L, W, W2, D = pk.L, pk.W, pk.W2, pk.D
    R, S = z.shape
    M = R * S
    xyz = (rays_o[:, None, :] + rays_d[:, None, :] * z[..., None]).reshape(M, 3)
    x0 = F.pad(posenc_w(xyz, 10, wk_xyz), (0, 1))
    out = {"x0": x0, "h": [], "pre_h": []}
    ray = torch.arange(M) // S

    def keep(t):
        if keep_pre and t.requires_grad:
            t.retain_grad()
        return t

    h = x0
    for l in range(D):
        k = X0 if l == 0 else (X0 + W if l == pk.skip else W)
        w, b = mat(P, L.w[l], W, k), P[L.b[l]:L.b[l] + W]
        inp = torch.cat([x0, h], 1) if l == pk.skip else h
        pre = keep(inp @ w.t() + b)
        h = torch.relu(pre)
        out["pre_h"].append(pre)
        out["h"].append(h)
    pre_s = keep(h @ P[L.wsig:L.wsig + W] + P[L.bsig])
    out["pre_sig_s"], out["sigma_s"] = pre_s, F.softplus(pre_s)
    e = keep(h @ mat(P, L.we, W, W).t() + P[L.be:L.be + W])
    out["e"] = e
    if use_cand:
        wc1 = mat(P, L.wc1, W2, W + CK)
        pre_g1 = keep(torch.cat([e, c_rows[ray]], 1) @ wc1.t() + P[L.bc1:L.bc1 + W2])
        g1 = torch.relu(pre_g1)
        pre_g2 = keep(g1 @ mat(P, L.wc2, W2, W2).t() + P[L.bc2:L.bc2 + W2])
        g2 = torch.relu(pre_g2)
        pre_c = keep(g2 @ P[L.wcsig:L.wcsig + W2] + P[L.bcsig])
        out.update(pre_g1=pre_g1, g1=g1, pre_g2=pre_g2, g2=g2, pre_sig_c=pre_c, sigma_c=F.softplus(pre_c))
    if use_rgb:
        wr1 = mat(P, L.wr1, W2, W + AUXK)
        pre_r1 = keep(torch.cat([e, aux[ray]], 1) @ wr1.t() + P[L.br1:L.br1 + W2])
        r1 = torch.relu(pre_r1)
        pre_rgb = keep(r1 @ mat(P, L.wr2, 4, W2)[:3].t() + P[L.br2:L.br2 + 3])
        out.update(pre_r1=pre_r1, r1=r1, pre_rgb=pre_rgb, rgb=torch.sigmoid(pre_rgb))
    return out


def excl_cumprod(x):
    return torch.cumprod(torch.cat([torch.ones_like(x[:, :1]), x], -1)[:, :-1], -1)


def composite(f, z, mode, has_rgb, W):
    """Per-ray sums in trunk-width space + per-sample weights, exactly the outputs of upnerf_composite_fwd."""
    R, S = z.shape
    sig_s = f["sigma_s"].view(R, S)
    delta = torch.cat([z[:, 1:] - z[:, :-1], 1e2 * torch.ones_like(z[:, :1])], -1)
    a_s = 1 - torch.exp(-delta * sig_s)
    o = {}
    e = f["e"].view(R, S, -1)
    if mode <= 1:
        sig_c = f["sigma_c"].view(R, S)
        a_c = 1 - torch.exp(-delta * sig_c)
        a_all = 1 - torch.exp(-delta * (sig_s + sig_c))
        T = excl_cumprod(1 - a_all)
        o["w_all"], o["w_sj"], o["w_cj"] = a_all * T, a_s * T, a_c * T
        o["c_depth"] = (o["w_all"] * z).sum(1)
        o["t_weight"] = o["w_cj"].sum(1)
        o["E_s"] = (o["w_sj"][..., None] * e).sum(1)
        o["G_c"] = (o["w_cj"][..., None] * f["g2"].view(R, S, -1)).sum(1)
        o["sum_sfeat"] = o["w_sj"].sum(1)
    o["w_s"] = a_s * excl_cumprod(1 - a_s)
    o["s_depth"] = (o["w_s"] * z).sum(1)
    if mode == 3:
        o["E_s"] = (o["w_s"][..., None] * e).sum(1)
        o["sum_sfeat"] = o["w_s"].sum(1)
    if has_rgb:
        o["rgb_map"] = (o["w_s"][..., None] * f["rgb"].view(R, S, 3)).sum(1)
    return o
