"""render_rays on MI355X: same signature, result keys and schedule semantics as the reference's
models/rendering.py:53-314, executed by the hand-written HIP kernels of libupnerf_hip.so.

Call graph of one pass (coarse or fine), all on the caller's current HIP stream:
    upnerf_ray_aux -> upnerf_field_fwd -> upnerf_composite_fwd -> (per-ray feature projection, upnerf_linear)
and of its backward:
    upnerf_composite_bwd -> upnerf_field_bwd -> upnerf_wgrad x layers (+ upnerf_vec_wgrad for the 1/3-wide heads)
    -> upnerf_ray_sum / upnerf_ray_geom_bwd for the per-ray inputs.
Between the passes: upnerf_sample_pdf + upnerf_sort_rows (no gradient: the reference detaches the weights,
rendering.py:271-303, and z is a constant w.r.t. every trainable).

There is no fallback: unsupported configurations raise."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch

from . import _lib, step_scalars, zero_pool
from ._lib import (CompositeBwdArgs, CompositeFwdArgs, FieldBwdArgs, FieldFwdArgs, AUXK, CK, RR_PART_STRIDE, TILE_PART_STRIDE, X0, check, lib,
                   ptr, stream)
from .ops import (TIMER, WgradChain, embed_rows, hip_linear, linear_kn_view, linear_raw, nsplit_for, vec_wgrad_frag16_into, vec_wgrad_into, wgrad_f16p_into,
                  wgrad_f16x3_into, wgrad_into, workspace)

__all__ = ["render_rays", "sample_pdf", "band_weights", "join_rays", "retain_ray_gradient", "ray_gradient"]

# Arithmetic of the field contractions (all HIP kernels of libupnerf_hip.so; there is no non-HIP path):
#   "f16x3"  3-term fp16 hi/lo split on the f16 matrix cores, fp32-level accuracy (csrc/field16.hip; needs W = 256 and
#            >= 32 samples per ray, other shapes use the fp32 kernels) -- the default, BASELINE.json configs[1]
#   "f32"    fp32 MFMA (csrc/field.hip) everywhere
#   "f16"    fp16 weights and activations, ONE MFMA per product, fp32 accumulate; encoding, density / colour outputs,
#            compositing, loss and optimiser stay fp32 (BASELINE.json configs[3]; same shape limits as f16x3, and no
#            silent fallback: an unsupported shape raises)
FIELD_MODE = __import__("os").environ.get("UPNERF_FIELD_MODE", "f16x3")  # env: diagnostic tools only
_MODES = ("f16x3", "f32", "f16")


def _field16_ok(pk, S: int) -> bool:
    if FIELD_MODE not in _MODES:
        raise ValueError(f"unknown FIELD_MODE {FIELD_MODE!r} (one of {_MODES})")
    ok = pk.W == 256 and S >= 32
    if FIELD_MODE == "f16" and not ok:
        raise ValueError("FIELD_MODE 'f16' needs W = 256 and at least 32 samples per ray")
    return FIELD_MODE != "f32" and ok


# Storage of the trunk activations / their gradients between the field kernels and the weight-gradient kernels in the f16x3
# mode (the f16 mode always uses "f16"):
#   "f32"  (default) fp32 tensors; weight gradients contracted with the 3-term split: fp32-accurate end to end
#   "f16"  fp16 tiles + per-tile exponents, half the HBM bytes; the weight gradients dW = gz^T h are contracted from the
#          fp16-ROUNDED operands with one MFMA per product.  Forward pass and data-gradient chain unchanged (bitwise the
#          default's outputs); a weight gradient carries ~3e-4 of unbiased rounding noise -- the golden gradient gates (1e-3)
#          stay green.  Measured 17.4 vs 20.0 ms per step; bench.py reports it as the extra object `wgrad_f16`, never as `value`.
#   (Also measured and dropped: lossless hi + lo fp16 PAIRS -- the same bytes as fp32 -- gave the field kernels and the
#   weight-gradient kernel nothing: 2.69 / 2.65 / 0.25 ms against 2.69 / 2.66 / 0.22.  It is the bytes, not the store pattern.)
WGRAD_STORE = __import__("os").environ.get("UPNERF_WGRAD_STORE", "f32")
# Per-tile partial sums of the vector heads and per-ray sums from the backward field kernel (upnerf_field_bwd_args.tile_part);
# 0 = the separate upnerf_vec_wgrad / upnerf_ray_sum launches (always used with the fp32-MFMA kernels).
TILE_PARTIALS = int(__import__("os").environ.get("UPNERF_TILE_PARTIALS", "1"))
# Slab reductions of the f16x3 weight gradients inside the next weight-gradient launch (upnerf_wgrad_f16x3_chain).
WGRAD_CHAIN = int(__import__("os").environ.get("UPNERF_WGRAD_CHAIN", "1"))
# Colour and candidate heads: [gz_r1 | gz_g1] stored as one tensor, one weight-gradient launch against e for both first layers.
# The shared density head's weight gradient inside the final layer's weight-gradient launch (same B operand); 0 = upnerf_vec_wgrad.
VEC_RIDE = int(__import__("os").environ.get("UPNERF_VEC_RIDE", "1"))
JOIN_RAYS = int(__import__("os").environ.get("UPNERF_JOIN_RAYS", "1"))  # (0: render_rays slices the [R][8] rows back apart, for A/B runs)
if WGRAD_STORE == "f24" and not WGRAD_CHAIN:  # (r4 ADVICE: the 24-bit operands have no un-chained entry point)
    raise RuntimeError("UPNERF_WGRAD_STORE=f24 needs UPNERF_WGRAD_CHAIN=1 (upnerf_wgrad_f24p_chain is the only kernel that reads the hi + lo8 operands)")
JOIN_HEADS = int(__import__("os").environ.get("UPNERF_JOIN_HEADS", "1"))
HMASK_SCALE = int(__import__("os").environ.get("UPNERF_HMASK_SCALE", "1"))  # experiment builds with more threads per tile
# fp16 mode: the register-resident kernels (csrc/field16rr.hip; include/upnerf_hip.h tile_rows = 256) -- weights staged once per
# 256 samples in an LDS ring, trunk activations / gradients stored as operand fragments.  0 = the 64-sample tile-in-LDS kernels.
FIELD_RR = int(__import__("os").environ.get("UPNERF_FIELD_RR", "1"))
RR_TILE = 256


def _rr_ok(S: int) -> bool:
    return bool(FIELD_RR) and FIELD_MODE == "f16" and S >= 32


def _planes() -> int:
    return 1 if FIELD_MODE == "f16" else 2


_LINSPACE: Dict[tuple, torch.Tensor] = {}
_DEBUG_SINK: Optional[dict] = None  # tests set this to a dict to receive the backward's intermediate buffers


def _linspace01(n: int, device) -> torch.Tensor:
    """torch.linspace(0, 1, n) evaluated on the host (the reference's values on its CPU path) and cached."""
    key = (n, str(device))
    t = _LINSPACE.get(key)
    if t is None:
        t = torch.linspace(0, 1, n, dtype=torch.float32).to(device)
        _LINSPACE[key] = t
    return t


def band_weights(L: int, progress: float, c2f) -> List[float]:
    """BARF coarse-to-fine weights w_k of the positional encoding (models/nerf.py:137-143), evaluated in fp32 on
    the host exactly as the reference evaluates them on its tensors."""
    if c2f is None:
        return [1.0] * L
    start, end = c2f
    alpha = (torch.tensor(float(progress), dtype=torch.float32) - start) / (end - start) * L
    k = torch.arange(L, dtype=torch.float32)
    w = (1 - ((alpha - k).clamp(min=0, max=1) * torch.pi).cos()) / 2
    return [float(x) for x in w]


def _empty(*shape, device):
    return torch.empty(*shape, device=device, dtype=torch.float32)


class _PassCfg:
    """Static (non-tensor) description of one field pass."""

    def __init__(self, packer, mode: int, use_cand: bool, use_rgb: bool, wk_xyz, wk_dir, rgb_joint: bool = False):
        self.packer, self.mode, self.use_cand, self.use_rgb = packer, mode, use_cand, use_rgb
        self.rgb_joint = bool(rgb_joint and mode <= 1 and use_rgb)  # encode_feat = False: the shared colour under the joint weights too
        self.wk_xyz, self.wk_dir = wk_xyz, wk_dir
        # ctx.needs_input_grad reports requires_grad of the inputs even under torch.no_grad(); whether a backward pass
        # can follow is decided where the pass is configured (Function.forward itself always runs with grad disabled)
        self.grad = torch.is_grad_enabled()


class _FieldPass(torch.autograd.Function):
    """Field evaluation + compositing of all samples of a ray batch.

    Returns per-ray sums in the trunk-width space (projected to the 384-d feature map by the caller) and the
    per-sample weights; see include/upnerf_hip.h:upnerf_composite_fwd_args for the meaning of every output."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, z, c_rows, a_rows, P, cfg: _PassCfg):
        ctx.set_materialize_grads(False)
        pk, L = cfg.packer, cfg.packer.L
        W, W2, D = pk.W, pk.W2, pk.D
        R, S = z.shape
        M = R * S
        dev = z.device
        st = stream()
        rays_o, rays_d, z = rays_o.detach().contiguous(), rays_d.detach().contiguous(), z.detach().contiguous()
        P = P.detach().contiguous()
        use16 = _field16_ok(pk, S)
        P16 = PT16 = wexp = wnorm = None
        rr = use16 and _rr_ok(S)  # register-resident fp16 kernels: per-sample tensors padded to whole 256-sample tiles
        Mp = (M + RR_TILE - 1) // RR_TILE * RR_TILE if rr else M
        if use16:  # matrices as scaled fp16 (hi, lo) fragments, forward and transposed sets in one pass
            P16, PT16, wexp, wnorm = pk.frag16_hip(P, perm=rr)
            PF = P  # the kernel reads only the vectors from it
        else:
            PF = pk.frag_hip(P)  # what the kernels read: matrices in MFMA fragment order
        c_rows = c_rows.detach().contiguous() if (c_rows is not None and cfg.use_cand) else None
        a_rows_c = a_rows.detach().contiguous() if a_rows is not None else None
        joint, want_feat = cfg.mode <= 1, cfg.mode != 2

        # per-step scalars: by value when run eagerly, from the device table of the capture in progress otherwise
        dyn = step_scalars.current() if cfg.grad else None
        aux = None
        if cfg.use_rgb:
            aux = _empty(R, AUXK, device=dev)
            wk = (C.c_float * 4)(*cfg.wk_dir)
            check(lib.upnerf_ray_aux(R, ptr(rays_d), ptr(a_rows_c), wk, dyn.ptr_named("wk_dir", 4) if dyn else None,
                                     ptr(aux), st), "upnerf_ray_aux")

        sigma_s = _empty(M, device=dev)
        sigma_c = _empty(M, device=dev) if cfg.use_cand else None
        rgb = _empty(M, 3, device=dev) if cfg.use_rgb else None
        # No gradient wanted (validation / test renders under torch.no_grad()): the kernels skip every store that only
        # the backward pass reads -- 8 of the 11 KB per sample -- and keep what compositing needs (e, g2) plus x0.
        train = cfg.grad and any(ctx.needs_input_grad)
        x0 = _empty(Mp, X0, device=dev)[:M]
        # f16 mode: trunk activations are STORED as fp16 (the kernel's LDS plane per 64-sample tile + the tile's exponent):
        # half the bytes written here and read back by the weight-gradient kernels; fp32 only for the last layer (its
        # consumers are the density-head and final-layer weight gradients)
        store16 = train and use16 and (FIELD_MODE == "f16" or WGRAD_STORE in ("f16", "f24"))
        # "f24" (f16x3 mode only): beside h16 a byte tensor with the residual of every element -- hi + lo to 2^-20 of the tile's
        # maximum in 3 bytes instead of 4 (an option like "f16", reported beside `value`)
        store24 = store16 and FIELD_MODE != "f16" and WGRAD_STORE == "f24" and not rr
        ntile = Mp // 32 if rr else (M + 63) // 64  # exponent tables: one entry per 64 rows whatever the kernel's tile (rr: per 32)
        h16 = torch.empty(D, Mp, W, device=dev, dtype=torch.float16) if store16 else None  # (rr: fragment order, see dequant16)
        hexp = torch.empty(D, ntile, device=dev, dtype=torch.int32) if store16 else None
        h_lo8 = torch.empty(D, M, W, device=dev, dtype=torch.uint8) if store24 else None
        # (rr: no fp32 copy of the last layer either -- the density head's and the final layer's weight gradients read its fragments)
        h = (None if rr else _empty(1, M, W, device=dev) if store16 else _empty(D, M, W, device=dev)) if train else None
        # rr: e leaves as fp16 operand fragments only (compositing and the joined heads' weight gradient read those) unless a
        # single head is on in training, whose separate weight gradient wants fp32 rows
        e_frag = bool(rr and (train or want_feat) and (not train or _join_ok(cfg, W, use16, rr, tile_ok=True)))
        e = _empty(Mp, W, device=dev)[:M] if ((train or want_feat) and not e_frag) else None
        e16 = torch.empty(Mp, W, device=dev, dtype=torch.float16) if e_frag else None
        eexp = torch.empty(Mp // 32, device=dev, dtype=torch.int32) if e_frag else None
        hmask = (torch.empty((D + 3) * Mp * 4 if rr else (D + 1) * ((M + 127) // 128) * 512 * HMASK_SCALE, device=dev,
                             dtype=torch.int64) if train else None)  # 64 bits per lane and tile, either tiling (rr: 128 per lane)
        # running max|.| of the stored tensors (scales of the f16x3 weight gradients): slots [0, 16) filled by this pass, [16, 32)
        # by the backward kernel -- one zero fill and, later, one exponent launch for both
        mx32 = zero_pool.zeros(32, dev) if train else None
        amax = mx32[:16] if train else None
        g1 = _empty(Mp, W2, device=dev)[:M] if (cfg.use_cand and train and not e_frag) else None
        g1_16 = torch.empty(Mp, W2, device=dev, dtype=torch.float16) if (cfg.use_cand and train and e_frag) else None
        g1exp = torch.empty(Mp // 32, device=dev, dtype=torch.int32) if g1_16 is not None else None
        # rr with e as fragments: g2 and r1 leave the same way (compositing / the 128-wide output layers' weight gradients read
        # them; the backward kernel works from the sign bits)
        g2 = _empty(Mp, W2, device=dev)[:M] if (cfg.use_cand and (train or joint) and not e_frag) else None
        r1 = _empty(Mp, W2, device=dev)[:M] if (cfg.use_rgb and train and not e_frag) else None
        g2_16 = torch.empty(Mp, W2, device=dev, dtype=torch.float16) if (cfg.use_cand and (train or joint) and e_frag) else None
        g2exp = torch.empty(Mp // 32, device=dev, dtype=torch.int32) if g2_16 is not None else None
        r1_16 = torch.empty(Mp, W2, device=dev, dtype=torch.float16) if (cfg.use_rgb and train and e_frag) else None
        r1exp = torch.empty(Mp // 32, device=dev, dtype=torch.int32) if r1_16 is not None else None
        tile = RR_TILE if rr else 64  # samples per workgroup (include/upnerf_hip.h: tile_rows); the backward pass gets the same
        x0f = None
        fa = FieldFwdArgs(R=R, S=S, use_cand=int(cfg.use_cand), use_rgb=int(cfg.use_rgb), rays_o=ptr(rays_o),
                          rays_d=ptr(rays_d), z=ptr(z), c_rows=ptr(c_rows), aux=ptr(aux),
                          wk_xyz=(C.c_float * 10)(*cfg.wk_xyz), P=ptr(PF), sigma_s=ptr(sigma_s), sigma_c=ptr(sigma_c),
                          rgb=ptr(rgb), x0=ptr(x0), h=ptr(h), hmask=ptr(hmask), amax=ptr(amax), e=ptr(e), g1=ptr(g1), g2=ptr(g2), r1=ptr(r1),
                          P16=ptr(P16), wexp=ptr(wexp), wk_xyz_dev=dyn.ptr_named("wk_xyz", 10) if dyn else None,
                          planes=_planes(), tile_rows=tile, wnorm=ptr(wnorm), h16=ptr(h16), hexp=ptr(hexp),
                          h_last_only=int(store16), x0f=ptr(x0f), e16=ptr(e16), eexp=ptr(eexp),
                          g2_16=ptr(g2_16), g2exp=ptr(g2exp), r1_16=ptr(r1_16), r1exp=ptr(r1exp), g1_16=ptr(g1_16), g1exp=ptr(g1exp), h_lo8=ptr(h_lo8),
                          rows_capacity=Mp if rr else 0)  # (rr: every per-sample tensor above was allocated with Mp rows)
        fwd_fn = lib.upnerf_field_fwd_f16x3 if use16 else lib.upnerf_field_fwd
        check(TIMER.run("field_fwd", lambda: fwd_fn(C.byref(L), C.byref(fa), st), units=M), "upnerf_field_fwd")

        w_all = _empty(R, S, device=dev) if joint else None
        w_sj = _empty(R, S, device=dev) if joint else None
        w_cj = _empty(R, S, device=dev) if joint else None
        w_s = _empty(R, S, device=dev)
        E_s = _empty(R, W, device=dev) if want_feat else None
        G_c = _empty(R, W2, device=dev) if joint else None
        sum_sfeat = _empty(R, device=dev) if want_feat else None
        t_weight = _empty(R, device=dev) if joint else None
        c_depth = _empty(R, device=dev) if joint else None
        s_depth = _empty(R, device=dev)
        rgb_map = _empty(R, 3, device=dev) if cfg.use_rgb else None
        rgbj_map = _empty(R, 3, device=dev) if cfg.rgb_joint else None
        ca = CompositeFwdArgs(R=R, S=S, W=W, mode=cfg.mode, z=ptr(z), sigma_s=ptr(sigma_s), sigma_c=ptr(sigma_c),
                              rgb=ptr(rgb), has_rgb=int(cfg.use_rgb), e=ptr(e), g2=ptr(g2), w_all=ptr(w_all),
                              w_sj=ptr(w_sj), w_cj=ptr(w_cj), w_s=ptr(w_s), E_s=ptr(E_s), G_c=ptr(G_c),
                              sum_sfeat=ptr(sum_sfeat), t_weight=ptr(t_weight), c_depth=ptr(c_depth),
                              s_depth=ptr(s_depth), rgb_map=ptr(rgb_map), e16=ptr(e16), eexp=ptr(eexp), g2_16=ptr(g2_16), g2exp=ptr(g2exp),
                              rgb_joint_map=ptr(rgbj_map))
        check(TIMER.run("composite_fwd", lambda: lib.upnerf_composite_fwd(C.byref(ca), st), units=M),
              "upnerf_composite_fwd")

        ctx.cfg, ctx.dims, ctx.planes, ctx.tile_rows = cfg, (R, S), _planes(), tile
        ctx.rr, ctx.Mp = rr, Mp
        ctx.has_a = a_rows is not None
        ctx.saved = dict(rays_o=rays_o, rays_d=rays_d, z=z, c_rows=c_rows, aux=aux, P=P, sigma_s=sigma_s,
                         sigma_c=sigma_c, rgb=rgb, x0=x0, h=h, h16=h16, hexp=hexp, h_lo8=h_lo8, hmask=hmask, amax=amax, mx32=mx32, e=e, e16=e16, eexp=eexp, g2_16=g2_16, g2exp=g2exp, r1_16=r1_16, r1exp=r1exp, g1_16=g1_16, g1exp=g1exp, g1=g1, g2=g2, r1=r1, PT16=PT16, wexp=wexp, x0f=x0f,
                         w_all=w_all, w_sj=w_sj,
                         w_cj=w_cj, w_s=w_s, wnorm=wnorm)
        z0 = torch.zeros(0, device=dev)
        outs = (E_s, G_c, sum_sfeat, t_weight, c_depth, s_depth, rgb_map, w_all, w_s, rgbj_map)
        return tuple(o if o is not None else z0 for o in outs)

    @staticmethod
    def backward(ctx, gE, gG, gsf, gtw, gcd, gsd, grm, gwall, gws, grj):
        cfg, (R, S), sv = ctx.cfg, ctx.dims, ctx.saved
        pk, L = cfg.packer, cfg.packer.L
        W, W2, D = pk.W, pk.W2, pk.D
        M = R * S
        dev = sv["z"].device
        st = stream()
        joint, want_feat = cfg.mode <= 1, cfg.mode != 2

        def g(t, ok=True):
            return t.contiguous() if (ok and t is not None and t.numel() > 0) else None

        gE, gsf = g(gE, want_feat), g(gsf, want_feat)
        gG, gtw, gcd, gwall = g(gG, joint), g(gtw, joint), g(gcd, joint), g(gwall, joint)
        gsd, gws, grm, grj = g(gsd), g(gws), g(grm, cfg.use_rgb), g(grj, cfg.rgb_joint)

        d_sigma_s = _empty(M, device=dev)
        d_sigma_c = _empty(M, device=dev) if joint else None
        d_rgb = _empty(M, 3, device=dev) if cfg.use_rgb else None
        cb = CompositeBwdArgs(R=R, S=S, W=W, mode=cfg.mode, has_rgb=int(cfg.use_rgb), z=ptr(sv["z"]),
                              sigma_s=ptr(sv["sigma_s"]), sigma_c=ptr(sv["sigma_c"]), rgb=ptr(sv["rgb"]),
                              e=ptr(sv["e"]), g2=ptr(sv["g2"]), w_all=ptr(sv["w_all"]), w_sj=ptr(sv["w_sj"]),
                              w_cj=ptr(sv["w_cj"]), w_s=ptr(sv["w_s"]), g_E_s=ptr(gE), g_G_c=ptr(gG),
                              g_sum_sfeat=ptr(gsf), g_t_weight=ptr(gtw), g_c_depth=ptr(gcd), g_s_depth=ptr(gsd),
                              g_rgb_map=ptr(grm), g_w_all=ptr(gwall), g_w_s=ptr(gws), d_sigma_s=ptr(d_sigma_s),
                              d_sigma_c=ptr(d_sigma_c), d_rgb=ptr(d_rgb), e16=ptr(sv.get("e16")), eexp=ptr(sv.get("eexp")),
                              g2_16=ptr(sv.get("g2_16")), g2exp=ptr(sv.get("g2exp")), g_rgb_joint_map=ptr(grj))
        check(TIMER.run("composite_bwd", lambda: lib.upnerf_composite_bwd(C.byref(cb), st), units=M),
              "upnerf_composite_bwd")

        need_dxyz = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        P = sv["P"]
        use16 = sv["PT16"] is not None
        PT = None if use16 else pk.frag_t_hip(P)
        store16 = sv.get("h16") is not None
        rr, Mp = ctx.rr, ctx.Mp  # register-resident fp16 kernels: per-sample tensors padded to whole 256-sample tiles
        gz_e = None if rr else _empty(M, W, device=dev)  # (rr: layer D of gz16, as fragments)
        gz_h = None if store16 else _empty(D, M, W, device=dev)
        gz16 = torch.empty(D + int(rr), Mp, W, device=dev, dtype=torch.float16) if store16 else None  # (rr: fragment order)
        gzexp = torch.empty(D + int(rr), sv["hexp"].shape[1], device=dev, dtype=torch.int32) if store16 else None
        store24 = sv.get("h_lo8") is not None
        gz_lo8 = torch.empty(D, M, W, device=dev, dtype=torch.uint8) if store24 else None
        # [gz_r1 | gz_g1] as ONE [M][W] tensor when both heads are on and the chained f16x3 weight gradients run: the two first
        # layers of the heads are both fed by e, so their weight gradients are one launch that reads e once (chain.wgrad2)
        joined = _join_ok(cfg, W, use16, rr, tile_ok=ctx.tile_rows == 64)  # (upnerf_ray_sum, the fallback, wants dense tensors)
        rg16 = joined and rr and sv.get("e16") is not None  # rr: [gz_r1 | gz_g1] as fp16 fragments only, against e's fragments
        gz_rg = _empty(Mp, W, device=dev)[:M] if (joined and not rg16) else None
        gz_rg16 = torch.empty(Mp, W, device=dev, dtype=torch.float16) if rg16 else None
        gzrgexp = torch.empty(Mp // 32, device=dev, dtype=torch.int32) if rg16 else None
        gz_g1 = (None if rg16 else gz_rg[:, W2:] if joined else _empty(Mp, W2, device=dev)[:M]) if cfg.use_cand else None
        g2f = cfg.use_cand and sv.get("g1_16") is not None  # rr: gz_g2 as fragments against g1's (candidate_encoding.2)
        gz_g2 = _empty(Mp, W2, device=dev)[:M] if (cfg.use_cand and not g2f) else None
        gz_g2_16 = torch.empty(Mp, W2, device=dev, dtype=torch.float16) if g2f else None
        gzg2exp = torch.empty(Mp // 32, device=dev, dtype=torch.int32) if g2f else None
        gz_r1 = (None if rg16 else gz_rg[:, :W2] if joined else _empty(Mp, W2, device=dev)[:M]) if cfg.use_rgb else None
        dpre_s = _empty(M, device=dev)
        dpre_c = _empty(M, device=dev) if cfg.use_cand else None
        dpre_rgb = _empty(M, 4, device=dev) if cfg.use_rgb else None
        dxyz = _empty(M, 3, device=dev) if need_dxyz else None
        gmax = sv["mx32"][16:]
        # 64-sample f16 kernels: per-tile partial sums of the 128-wide vector heads and of the per-ray sums, written by the
        # backward kernel (which holds those tiles anyway) and finished by three small launches -- instead of five kernels
        # that read M x 128 tensors again
        tile_part = (_empty((M + 63) // 64, TILE_PART_STRIDE, device=dev)
                     if (use16 and ctx.tile_rows == 64 and TILE_PARTIALS and (cfg.use_cand or cfg.use_rgb)) else None)
        # register-resident kernels: the per-ray sums only, per 32 samples (the 128-wide vector heads stay separate launches:
        # that kernel works from the sign bits of g2 / r1 and never holds their values)
        ray_part = (_empty(Mp // 32, RR_PART_STRIDE, device=dev)
                    if (rr and TILE_PARTIALS and (cfg.use_cand or cfg.use_rgb)) else None)
        w_feat = (sv["w_sj"] if joint else sv["w_s"]) if gE is not None else None
        fb = FieldBwdArgs(R=R, S=S, use_cand=int(cfg.use_cand), use_rgb=int(cfg.use_rgb), need_dxyz=int(need_dxyz),
                          PT=ptr(PT), P=ptr(P), d_sigma_s=ptr(d_sigma_s), d_sigma_c=ptr(d_sigma_c), d_rgb=ptr(d_rgb),
                          sigma_s=ptr(sv["sigma_s"]), sigma_c=ptr(sv["sigma_c"]), rgb=ptr(sv["rgb"]),
                          w_feat_s=ptr(w_feat), w_cj=ptr(sv["w_cj"]) if gG is not None else None, g_E_s=ptr(gE),
                          g_G_c=ptr(gG), x0=ptr(sv["x0"]), h=ptr(sv["h"]), g1=ptr(sv["g1"]), g2=ptr(sv["g2"]),
                          r1=ptr(sv["r1"]), hmask=ptr(sv["hmask"]), gmax=ptr(gmax), gz_h=ptr(gz_h), gz_e=ptr(gz_e),
                          gz_g1=(gz_rg.data_ptr() + 4 * W2) if gz_rg is not None else ptr(gz_g1), gz_g2=ptr(gz_g2),
                          gz_r1=ptr(gz_rg) if gz_rg is not None else ptr(gz_r1), gz_rg_ld=W if gz_rg is not None else 0,
                          gz_rg16=ptr(gz_rg16), gzrgexp=ptr(gzrgexp), gz_g2_16=ptr(gz_g2_16), gzg2exp=ptr(gzg2exp), gz_lo8=ptr(gz_lo8), dpre_sig_s=ptr(dpre_s), dpre_sig_c=ptr(dpre_c), dpre_rgb=ptr(dpre_rgb),
                          dxyz=ptr(dxyz), PT16=ptr(sv["PT16"]), wexp=ptr(sv["wexp"]), planes=ctx.planes, tile_rows=ctx.tile_rows, xs=ptr(sv.get("x0f")), gz16=ptr(gz16),
                          gzexp=ptr(gzexp), tile_part=ptr(ray_part if rr else tile_part), wnorm=ptr(sv.get("wnorm")),
                          rows_capacity=Mp if rr else 0)
        bwd_fn = lib.upnerf_field_bwd_f16x3 if use16 else lib.upnerf_field_bwd
        check(TIMER.run("field_bwd", lambda: bwd_fn(C.byref(L), C.byref(fb), st), units=M), "upnerf_field_bwd")

        if _DEBUG_SINK is not None:
            _DEBUG_SINK.update(d_sigma_s=d_sigma_s, d_sigma_c=d_sigma_c, d_rgb=d_rgb,
                               gz_h=gz_h if gz_h is not None else dequant16(gz16[:D], gzexp[:D], frag=rr)[:, :M],
                               gz_e=gz_e if gz_e is not None else dequant16(gz16[D:], gzexp[D:], frag=True)[0, :M],
                               gz_g1=dequant16(gz_rg16[None], gzrgexp[None], frag=True)[0, :M, W2:] if rg16 else gz_g1,
                               gz_r1=dequant16(gz_rg16[None], gzrgexp[None], frag=True)[0, :M, :W2] if rg16 else gz_r1,
                               gz_g2=gz_g2 if gz_g2 is not None else (dequant16(gz_g2_16[None], gzg2exp[None], frag=True)[0, :M] if g2f else None),
                               dpre_s=dpre_s, dpre_c=dpre_c, dpre_rgb=dpre_rgb, dxyz=dxyz)
        # ---- weight gradients, written straight into a buffer with P's layout
        dP = zero_pool.zeros(L.total, dev) if ctx.needs_input_grad[5] else None
        d_c_rows = d_a_rows = None
        if dP is not None:
            base = dP.data_ptr()
            at = lambda off: base + 4 * off
            h, x0 = sv["h"], sv["x0"]
            # power-of-two scales of the f16x3 contraction: 2^14 / max|.| per tensor, from the maxima the field kernels
            # tracked (device side, no host sync).  ea[i] pairs with gmax slot i, eb[i] with amax slot i.
            e32 = torch.empty(32, device=dev, dtype=torch.int32)
            check(lib.upnerf_scale_exponents(ptr(sv["mx32"]), 32, ptr(e32), st), "upnerf_scale_exponents")
            EA = lambda i: e32.data_ptr() + 4 * (16 + i)
            EB = lambda i: e32.data_ptr() + 4 * i
            ctx_keep = (e32,)

            # WGRAD_CHAIN: the slab reduction of every f16x3 weight gradient rides on the next one's launch (ops.WgradChain)
            chain = WgradChain(dev) if WGRAD_CHAIN else None

            def wg(gz, lda, N, Bt, ldb, K, off, ldo, boff, ia, ib, b_off=0, vhead=None):
                if chain is not None:
                    v, voff, vboff = vhead if vhead is not None else (None, None, None)
                    chain.wgrad(M, gz, lda, N, Bt, ldb, K, at(off), ldo, None if boff is None else at(boff), EA(ia), EB(ib),
                                b_off=b_off, planes=ctx.planes, v=v, dv_ptr=None if v is None else at(voff),
                                dbv_ptr=None if v is None else at(vboff))
                else:
                    wgrad_f16x3_into(M, gz, lda, N, Bt, ldb, K, at(off), ldo, None if boff is None else at(boff), dev,
                                     expo_a=EA(ia), expo_b=EB(ib), b_off=b_off, planes=ctx.planes)

            if store16:  # fp16-stored operands (1 KB per sample and layer instead of 2)
                h16, hexp = sv["h16"], sv["hexp"]

                def wgp(l, B, ldb, bexp, K, off, ldo, boff, ib):
                    if store24:  # hi + lo8 operands, three MFMAs per block (B: the previous layer's pair, or the fp32 encoding)
                        blo = sv["h_lo8"][l - 1] if bexp is not None else None
                        chain.wgrad_p24(M, gz16[l], gz_lo8[l], W, gzexp[l], W, B, blo, ldb, bexp, K, at(off), ldo,
                                        None if boff is None else at(boff), EA(l), EB(ib))
                    elif chain is not None:
                        chain.wgrad_p(M, gz16[l], W, gzexp[l], W, B, ldb, bexp, K, at(off), ldo, None if boff is None else at(boff),
                                      EA(l), EB(ib), frag=rr)
                    else:
                        wgrad_f16p_into(M, gz16[l], W, gzexp[l], W, B, ldb, bexp, K, at(off), ldo, None if boff is None else at(boff),
                                        dev, EA(l), EB(ib), frag=rr)

                for l in range(D):
                    if l == 0:
                        wgp(0, x0, X0, None, X0, L.w[0], X0, L.b[0], D + 4)
                    elif l == pk.skip:
                        wgp(l, x0, X0, None, X0, L.w[l], X0 + W, L.b[l], D + 4)
                        wgp(l, h16[l - 1], W, hexp[l - 1], W, L.w[l] + X0, X0 + W, None, l - 1)
                    else:
                        wgp(l, h16[l - 1], W, hexp[l - 1], W, L.w[l], W, L.b[l], l - 1)
            for l in (range(D) if not store16 else ()):
                gz = gz_h[l]
                if l == 0:
                    wg(gz, W, W, x0, X0, X0, L.w[0], X0, L.b[0], 0, D + 4)
                elif l == pk.skip:
                    wg(gz, W, W, x0, X0, X0, L.w[l], X0 + W, L.b[l], l, D + 4)
                    wg(gz, W, W, h[l - 1], W, W, L.w[l] + X0, X0 + W, None, l, l - 1)
                else:
                    wg(gz, W, W, h[l - 1], W, W, L.w[l], W, L.b[l], l, l - 1)
            if rr:  # both operands of the final layer's gradient, and the density head's, as fragments
                ride = VEC_RIDE and chain is not None  # the density head's gradient inside the final layer's launch (same B operand)
                if chain is not None:
                    chain.wgrad_p(M, gz16[D], W, gzexp[D], W, h16[D - 1], W, hexp[D - 1], W, at(L.we), W, at(L.be), EA(D), EB(D - 1), frag=True,
                                  v=dpre_s if ride else None, dv_ptr=at(L.wsig) if ride else None, dbv_ptr=at(L.bsig) if ride else None)
                else:
                    wgrad_f16p_into(M, gz16[D], W, gzexp[D], W, h16[D - 1], W, hexp[D - 1], W, at(L.we), W, at(L.be), dev, EA(D), EB(D - 1), frag=True)
                if not ride:
                    vec_wgrad_frag16_into(M, dpre_s, 1, 1, h16[D - 1], hexp[D - 1], W, at(L.wsig), at(L.bsig), dev)
            else:
                h_last = h[0] if store16 else h[D - 1]
                # the shared density head reads the B operand of the final layer's weight gradient: its gradient rides on that
                # launch (upnerf_wgrad_f16x3_chain_v: one read of h_last instead of two) where the launch is the f16x3 chain's
                ride = VEC_RIDE and chain is not None and ctx.planes == 2 and W == 256
                wg(gz_e, W, W, h_last, W, W, L.we, W, L.be, D, D - 1, vhead=(dpre_s, L.wsig, L.bsig) if ride else None)
                if not ride:
                    vec_wgrad_into(M, dpre_s, 1, 1, h_last, W, W, at(L.wsig), at(L.bsig), dev)
        rs_c = _empty(R, W2, device=dev) if cfg.use_cand else None
        rs_r = _empty(R, W2, device=dev) if cfg.use_rgb else None
        if tile_part is not None:
            want_w = dP is not None
            ws = workspace("tile_part", 128 * 520, dev) if want_w else None
            check(lib.upnerf_tile_part_finish(
                R, S, ptr(tile_part), ptr(rs_c), ptr(rs_r),
                at(L.wcsig) if (want_w and cfg.use_cand) else None, at(L.bcsig) if (want_w and cfg.use_cand) else None,
                at(L.wr2) if (want_w and cfg.use_rgb) else None, at(L.br2) if (want_w and cfg.use_rgb) else None,
                ptr(ws), st), "upnerf_tile_part_finish")
        if ray_part is not None:
            check(lib.upnerf_ray_part_finish(R, S, ptr(ray_part), ptr(rs_c), ptr(rs_r), st), "upnerf_ray_part_finish")
        if cfg.use_cand:
            if tile_part is None and ray_part is None:
                check(lib.upnerf_ray_sum(R, S, ptr(gz_g1), W2, ptr(rs_c), st), "upnerf_ray_sum")
            if dP is not None:
                if rg16:
                    chain.wgrad_p(M, gz_rg16, W, gzrgexp, W, sv["e16"], W, sv["eexp"], W, at(L.wr1), W + AUXK, at(L.br1), EA(D + 4), EB(D),
                                  frag=True, n2=W2, dW2_ptr=at(L.wc1), ldo2=W + CK, db2_ptr=at(L.bc1))
                elif joined:  # rows [0, W2) -> wr1 / br1 (colour head), rows [W2, W) -> wc1 / bc1 (candidate head)
                    chain.wgrad2(M, gz_rg, W, W, sv["e"], W, W, at(L.wr1), W + AUXK, at(L.br1), W2, at(L.wc1), W + CK, at(L.bc1),
                                 EA(D + 4), EB(D), planes=ctx.planes)
                else:
                    wg(gz_g1, W2, W2, sv["e"], W, W, L.wc1, W + CK, L.bc1, D + 1, D)
                wgrad_into(R, rs_c, W2, W2, sv["c_rows"], CK, CK, at(L.wc1 + W), W + CK, None, dev)
                if g2f:
                    chain.wgrad_p(M, gz_g2_16, W2, gzg2exp, W2, sv["g1_16"], W2, sv["g1exp"], W2, at(L.wc2), W2, at(L.bc2), EA(D + 2),
                                  EB(D + 1), frag=True)
                else:
                    wg(gz_g2, W2, W2, sv["g1"], W2, W2, L.wc2, W2, L.bc2, D + 2, D + 1)
                if sv.get("g2_16") is not None:
                    vec_wgrad_frag16_into(M, dpre_c, 1, 1, sv["g2_16"], sv["g2exp"], W2, at(L.wcsig), at(L.bcsig), dev)
                elif tile_part is None:
                    vec_wgrad_into(M, dpre_c, 1, 1, sv["g2"], W2, W2, at(L.wcsig), at(L.bcsig), dev)
            if ctx.needs_input_grad[3]:
                d_c_rows = linear_kn_view(rs_c, P, L.wc1 + W, W + CK, CK)  # rs . wc1[:, W:]
        if cfg.use_rgb:
            if tile_part is None and ray_part is None:
                check(lib.upnerf_ray_sum(R, S, ptr(gz_r1), W2, ptr(rs_r), st), "upnerf_ray_sum")
            if dP is not None:
                if not joined:
                    wg(gz_r1, W2, W2, sv["e"], W, W, L.wr1, W + AUXK, L.br1, D + 3, D)
                wgrad_into(R, rs_r, W2, W2, sv["aux"], AUXK, AUXK, at(L.wr1 + W), W + AUXK, None, dev)
                if sv.get("r1_16") is not None:
                    vec_wgrad_frag16_into(M, dpre_rgb, 4, 3, sv["r1_16"], sv["r1exp"], W2, at(L.wr2), at(L.br2), dev)
                elif tile_part is None:
                    vec_wgrad_into(M, dpre_rgb, 4, 3, sv["r1"], W2, W2, at(L.wr2), at(L.br2), dev)
            if ctx.has_a and ctx.needs_input_grad[4]:
                d_a_rows = linear_kn_view(rs_r, P, L.wr1 + W + 27, W + AUXK, 48)  # rs . wr1[:, W+27 : W+75]
        if dP is not None and chain is not None:
            chain.finish()
        d_o = d_d = None
        if need_dxyz:
            d_o, d_d = _empty(R, 3, device=dev), _empty(R, 3, device=dev)
            check(lib.upnerf_ray_geom_bwd(R, S, ptr(dxyz), ptr(sv["z"]), ptr(d_o), ptr(d_d), st), "upnerf_ray_geom_bwd")
        ctx.saved = None
        return d_o, d_d, None, d_c_rows, d_a_rows, dP, None


def _join_ok(cfg, W: int, use16: bool, rr: bool, tile_ok: bool) -> bool:
    """[gz_r1 | gz_g1] as one tensor with one weight-gradient launch against e (both heads on, chained f16 weight gradients,
    per-tile partial sums: the 64-sample tiling or the register-resident kernels)."""
    return bool(use16 and cfg.use_cand and cfg.use_rgb and JOIN_HEADS and WGRAD_CHAIN and W == 256 and (tile_ok or rr)
                and TILE_PARTIALS)


def dequant16(t16: torch.Tensor, texp: torch.Tensor, frag: bool = False) -> torch.Tensor:
    """fp32 view of an fp16-stored, tile-scaled tensor [D][M][W] with exponents [D][ceil(M/64)] (tests, debugging).
    frag: the register-resident kernels' layout (include/upnerf_hip.h, tile_rows = 256) -- [D][M/32][k-block 16][lane 64][8]
    with feature 16 s + 8 (j / 4) + 4 (lane / 32) + j % 4, row 32 tile + lane % 32, one exponent per 32 rows."""
    D, M, W = t16.shape
    if frag:
        t = t16.view(D, M // 32, W // 16, 2, 32, 2, 4)        # [D][tile][s][hh][li][j>>2][j&3]
        rows = t.permute(0, 1, 4, 2, 5, 3, 6).reshape(D, M, W)  # [D][tile][li][s][j>>2][hh][j&3]
        scale = torch.ldexp(torch.ones((), device=t16.device), -texp.float()).repeat_interleave(32, dim=1)[:, :M]
        return rows.float() * scale[:, :, None]
    scale = torch.ldexp(torch.ones((), device=t16.device), -texp.float()).repeat_interleave(64, dim=1)[:, :M]
    return t16.float() * scale[:, :, None]


def quant16_frag(rows: torch.Tensor, texp: torch.Tensor) -> torch.Tensor:
    """Inverse of dequant16(frag=True) for one layer (tests): rows [M][W] fp32 (M a multiple of 32), texp [M/32] -> fp16
    fragments [M][W] holding rows * 2^texp[tile]."""
    M, W = rows.shape
    x = (rows * torch.ldexp(torch.ones((), device=rows.device), texp.float()).repeat_interleave(32)[:, None]).to(torch.float16)
    return x.view(M // 32, 32, W // 16, 2, 2, 4).permute(0, 2, 4, 1, 3, 5).reshape(M, W).contiguous()  # [tile][s][hh][li][j>>2][j&3]


def sample_pdf(z_coarse: torch.Tensor, weights: torch.Tensor, n: int, det: bool, out: torch.Tensor, col0: int,
               u: Optional[torch.Tensor] = None):
    """Inverse-CDF resampling of one weight set (models/rendering.py:7-50) into out[:, col0:col0+n].

    z_coarse [R][S] (bins = interval mid-points), weights [R][S] (entries 1..S-2 are used, as the reference
    passes weights[:, 1:-1]).  det=True uses u = linspace(0,1,n); otherwise `u` ([R][n]) or fresh torch.rand."""
    R, S = z_coarse.shape
    if n == 0:
        return
    if det:
        uu, rows = _linspace01(n, z_coarse.device), 1
    else:
        uu = u if u is not None else torch.rand(R, n, device=z_coarse.device)
        uu, rows = uu.contiguous(), R
        if tuple(uu.shape) != (R, n):
            raise ValueError(f"sample_pdf: u has shape {tuple(uu.shape)}, expected {(R, n)}")
    stride = out.shape[1]
    check(lib.upnerf_sample_pdf(R, S, ptr(z_coarse), ptr(weights), ptr(uu), rows, n, out.data_ptr() + 4 * col0, stride,
                                stream()), "upnerf_sample_pdf")


_ZEROS = {}


def _zeros(rows, cols, device):
    """Cached [rows, cols] zero block (K padding of _project_feat; never written)."""
    key = (rows, cols, device)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(rows, cols, device=device)
    return _ZEROS[key]


class _ProjectFeat(torch.autograd.Function):
    """feat = [x_0 | x_1 | ... | 0] . [w_0 | w_1 | ... | 0]^T with the pieces gathered by ONE pack launch and every gradient
    handed back contiguous by ONE unpack launch (torch.cat + autograd's slices cost a copy launch per piece: the compositing
    backward wants dense [R][W] rows, AccumulateGrad clones a strided weight gradient)."""

    @staticmethod
    def forward(ctx, n, *tensors):
        from ._lib import PackDesc
        xs, ws = tensors[:n], tensors[n:]
        R, N, dev = xs[0].shape[0], ws[0].shape[0], xs[0].device
        cols = [x.shape[1] for x in xs]
        K = sum(cols)
        Kp = (K + 31) // 32 * 32
        st = stream()
        keep = [t.detach().contiguous() for t in tensors]
        buf = torch.empty((R + N) * Kp, device=dev)
        descs, k0 = [], 0
        for j, c in enumerate(cols):
            descs.append(PackDesc(ptr(keep[j]), R, c, c, k0, Kp, 0))
            descs.append(PackDesc(ptr(keep[n + j]), N, c, c, R * Kp + k0, Kp, 0))
            k0 += c
        if Kp != K:  # the padding columns come from a cached zero block: no fill launch
            descs.append(PackDesc(ptr(_zeros(R, Kp - K, dev)), R, Kp - K, Kp - K, K, Kp, 0))
            descs.append(PackDesc(ptr(_zeros(N, Kp - K, dev)), N, Kp - K, Kp - K, R * Kp + K, Kp, 0))
        arr = (PackDesc * len(descs))(*descs)
        check(lib.upnerf_pack(ptr(buf), arr, len(descs), 0, st), "upnerf_pack")
        X, Wc = buf[:R * Kp].view(R, Kp), buf[R * Kp:].view(N, Kp)
        y = _empty(R, N, device=dev)
        check(lib.upnerf_linear(R, N, Kp, ptr(X), Kp, ptr(Wc), Kp, None, ptr(y), N, 0, st), "upnerf_linear")
        ctx.n, ctx.cols, ctx.dims, ctx.buf = n, cols, (R, N, K, Kp), buf
        ctx.shapes = [tuple(t.shape) for t in tensors]
        return y

    @staticmethod
    def backward(ctx, gy):
        from ._lib import PackDesc
        from .ops import wgrad_blocks_into
        n, cols, (R, N, K, Kp), buf = ctx.n, ctx.cols, ctx.dims, ctx.buf
        dev = gy.device
        st = stream()
        gy = gy.contiguous()
        X, Wc = buf[:R * Kp].view(R, Kp), buf[R * Kp:].view(N, Kp)
        g = torch.empty((R + N) * Kp, device=dev)
        gX, gW = g[:R * Kp].view(R, Kp), g[R * Kp:].view(N, Kp)
        check(lib.upnerf_linear(R, Kp, N, ptr(gy), N, ptr(Wc), Kp, None, ptr(gX), Kp, 2, st), "upnerf_linear")  # gy . Wc
        wgrad_blocks_into(R, gy, N, N, X, Kp, Kp, ptr(gW), Kp, None, dev)                                        # gy^T . X
        outs = [torch.empty(sh, device=dev) for sh in ctx.shapes]
        descs, k0 = [], 0
        for j, c in enumerate(cols):
            if ctx.needs_input_grad[1 + j]:
                descs.append(PackDesc(ptr(outs[j]), R, c, c, k0, Kp, 0))
            if ctx.needs_input_grad[1 + n + j]:
                descs.append(PackDesc(ptr(outs[n + j]), N, c, c, R * Kp + k0, Kp, 0))
            k0 += c
        if descs:
            arr = (PackDesc * len(descs))(*descs)
            check(lib.upnerf_pack(ptr(g), arr, len(descs), 1, st), "upnerf_pack")
        return (None,) + tuple(o if ctx.needs_input_grad[1 + i] else None for i, o in enumerate(outs))


def _project_feat(model, E_s, sum_sfeat, G_c=None, t_weight=None):
    """feat map = W_f (sum w e) + b_f sum w  [+ W_cf (sum w_c g) + b_cf sum w_c]  (nerf.py:95,100 composited by
    rendering.py:166-177) as ONE product [E_s | sum w | G_c | sum w_c | 0] . [W_f | b_f | W_cf | b_cf | 0]^T: the bias terms
    ride along as one more input column each (K padded to the next multiple of 32 for the weight-gradient kernel)."""
    fs = model.feat_share_layer
    xs, ws = [E_s, sum_sfeat[:, None]], [fs.weight, fs.bias[:, None]]
    if G_c is not None:
        fc = model.feat_candidate_layer
        xs += [G_c, t_weight[:, None]]
        ws += [fc.weight, fc.bias[:, None]]
    return _ProjectFeat.apply(len(xs), *xs, *ws)


def _project_rgb(model, G_c, t_weight):
    """The candidate half of `c_rgb` (encode_feat = False, rendering.py:183-188): sum_i w_cj (W_rc g2_i + b_rc) =
    W_rc (sum w_cj g2) + b_rc sum w_cj -- the same composite-then-project algebra as _project_feat, three columns wide."""
    rc = model.rgb_candidate_layer
    # (eight output columns, five of them zero: the data-gradient GEMM of _ProjectFeat contracts over them in blocks of eight)
    w8 = torch.nn.functional.pad(rc.weight, (0, 0, 0, 5))
    b8 = torch.nn.functional.pad(rc.bias, (0, 5))
    return _ProjectFeat.apply(2, G_c, t_weight[:, None], w8, b8[:, None])[:, :3]


def join_rays(rays_o, rays_d, near_far):
    """The [R][8] rows render_rays takes (o | d | near far; models/nerf_system.py:166), remembering the tensors they were
    concatenated from: render_rays then reads origins and directions from those instead of slicing them back out of the rows
    (two copy launches forward; backward two zero fills, two copies, an add and two more copies to hand contiguous gradients
    to the pose kernel).  The values are the same; the gradient then reaches `rays_o` / `rays_d` directly and `rays.grad`
    stays empty -- `ray_gradient(rays)` returns it either way."""
    rays = torch.cat([rays_o, rays_d, near_far], 1)
    if JOIN_RAYS and rays_o.dtype == torch.float32 and rays_o.dim() == 2 and rays_o.shape[1] == 3 and rays_d.shape == rays_o.shape:
        # ... as long as nobody has written to the rows since (r5 ADVICE): render_rays takes the parts only while the tensor's
        # version counter still says so; an in-place edit of `rays` (rays[:, 3:6] = ..., a near / far rescale) sends it back to
        # slicing the rows, which is the drop-in contract of render_rays(rays=...)
        rays._upnerf_parts = (rays_o, rays_d, near_far, rays._version)
    return rays


def retain_ray_gradient(rays):
    """Ask for the gradient w.r.t. the rows of `rays` (call before backward; tests)."""
    for t in (rays,) + tuple(getattr(rays, "_upnerf_parts", ())[:2]):
        if t.requires_grad and not t.is_leaf:
            t.retain_grad()


def ray_gradient(rays):
    """d loss / d rays [R][8] after a backward preceded by retain_ray_gradient, or None."""
    parts = getattr(rays, "_upnerf_parts", None)
    if parts is not None and (parts[0].grad is not None or parts[1].grad is not None):
        z = lambda t, g: torch.zeros_like(t) if g is None else g
        g = torch.cat([z(parts[0], parts[0].grad), z(parts[1], parts[1].grad), torch.zeros_like(parts[2])], 1)
        return g if rays.grad is None else g + rays.grad
    return rays.grad


def render_rays(models, embeddings, rays, img_idx, sched_mult, N_samples=64, use_disp=False, perturb=0,
                N_importance=0, test_time=False, encode_feat=True, **kwargs):
    """Drop-in for the reference's render_rays (models/rendering.py:53-66): same positional/keyword arguments,
    unknown kwargs (sched_phase, white_back, validation) accepted and ignored, same result keys
    per schedule phase.  `test_time` is dead in the reference (Q1) and is ignored here too.

    Extension for training (NeRFSystem passes it): kwargs["rng"] = dict(seed, step, row0) draws the stratified-sampling
    uniforms from a Philox counter keyed by (seed, step, row0 + ray, draw, column) instead of torch.rand -- the same numbers
    whatever the number of ranks the global batch is split over.

    Extensions used by the parity tests: kwargs["u_list"] = explicit uniform draws consumed in the reference's RNG
    call order (coarse jitter [R,Nc], then the sample_pdf draws); kwargs["keep"] = dict that receives the sampled
    depths (z_coarse, z_fine); kwargs["z_fine"] = [R, N_samples + N_importance] sorted depths the fine pass is evaluated
    at INSTEAD of resampling (the resampling carries no gradient, rendering.py:271-306: with the reference's own depths
    from a golden every fine-pass output and gradient compares strictly, whatever an eps-mass bin of the inverse CDF did).

    Extension for the eval / test-time-optimisation callers (nerf_system_optmize.py:84-111 read `s_rgb_fine` only):
    kwargs["coarse_sigma_only"] = True evaluates the coarse field up to its density head only, without gradient (the
    resampling weights are detached in the reference too, rendering.py:271-306), and returns just `s_weights_coarse` /
    `s_depth_coarse` for it; the fine pass is unchanged, bit for bit."""
    if not rays.is_cuda:
        raise RuntimeError("upnerf_amd.render_rays runs on the GPU only (no CPU fallback)")
    draws = list(kwargs["u_list"]) if kwargs.get("u_list") is not None else None
    dev = rays.device
    R = rays.shape[0]
    st = stream()

    rng = kwargs.get("rng")  # dict(seed, step, row0[, stride]): keyed draws (upnerf_uniform_keyed); absent: torch.rand like the reference
    n_drawn = [0]

    def draw(n):
        if draws is not None:
            t = draws.pop(0).to(dev, torch.float32).contiguous()
            if tuple(t.shape) != (R, n):
                raise ValueError(f"u_list entry has shape {tuple(t.shape)}, expected {(R, n)}")
            return t
        if rng is None:
            return torch.rand(R, n, device=dev)
        # Philox keyed by (seed, step, global ray, draw, column): independent of how the batch is split over ranks; under
        # graph replay the step counter is read from the per-step scalar table
        t = _empty(R, n, device=dev)
        n_drawn[0] += 1
        if n == 0:
            return t
        dyn = step_scalars.current()
        check(lib.upnerf_uniform_keyed(R, n, int(rng["seed"]) & 0xFFFFFFFFFFFFFFFF, int(rng["step"]),
                                       dyn.ptr_named("step", 1) if dyn else None, int(rng["row0"]), int(rng.get("stride", 1)),
                                       n_drawn[0] - 1, ptr(t), st),
              "upnerf_uniform_keyed")
        return t

    parts = getattr(rays, "_upnerf_parts", None)
    if (parts is not None and parts[3] == rays._version and parts[0].shape == (R, 3) and parts[0].is_contiguous()
            and parts[1].is_contiguous()):
        rays_o, rays_d = parts[0], parts[1]  # join_rays: the tensors `rays` was concatenated from, no slice copies
        near_far = parts[2].detach().contiguous()
    else:
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()  # once, not once per field pass
        near_far = rays[:, 6:8].detach().contiguous()
    z = _empty(R, N_samples, device=dev)
    u0 = draw(N_samples) if perturb > 0 else None
    check(lib.upnerf_sample_coarse(R, N_samples, ptr(near_far), ptr(_linspace01(N_samples, dev)), ptr(u0),
                                   float(perturb), int(bool(use_disp)), ptr(z), st), "upnerf_sample_coarse")
    results = {}

    def inference(model, zz, sigma_only=False):
        typ = model.typ
        if sigma_only:  # nerf.py:90-91: trunk + density head, composited into resampling weights and a depth
            hp = getattr(model, "host_progress", None)
            progress = float(model.progress.data) if hp is None else float(torch.tensor(hp, dtype=torch.float32))
            with torch.no_grad():
                cfg = _PassCfg(model.packer, 2, False, False, band_weights(model.xyz_L, progress, model.c2f),
                               band_weights(model.dir_L, progress, model.c2f))
                outs = _FieldPass.apply(rays_o, rays_d, zz, None, None, model.packed(), cfg)
            results[f"s_weights_{typ}"], results[f"s_depth_{typ}"] = outs[8], outs[5]
            return
        a_rows = embed_rows(embeddings[f"{typ}_a"], img_idx, defer_grad=True) if model.encode_appearance else None
        c_rows = embed_rows(embeddings[f"{typ}_c"], img_idx, defer_grad=True) if model.encode_candidate else None
        # host mirror kept by NeRF.set_progress; code that writes model.progress.data directly (the reference's way)
        # leaves it None and pays a device read here
        hp = getattr(model, "host_progress", None)
        progress = float(model.progress.data) if hp is None else float(torch.tensor(hp, dtype=torch.float32))
        if bool(getattr(model, "encode_feat", True)) != bool(encode_feat):
            raise ValueError(f"render_rays(encode_feat={encode_feat}) with a {typ} field built with encode_feat={model.encode_feat}")
        if not encode_feat and sched_mult < 1 and not model.encode_candidate:
            # the reference has no such path either: `raise NotImplemented` (rendering.py:149-150)
            raise NotImplementedError("encode_feat=False needs the candidate head while sched_mult < 1 (models/rendering.py:149-150)")
        use_cand = bool(sched_mult < 1 and model.encode_candidate)
        use_rgb = bool(sched_mult > 0 or not encode_feat)  # without features the colour head always runs (nerf.py:110-117)
        mode = (1 if use_rgb else 0) if use_cand else (3 if sched_mult < 1 else 2)
        cfg = _PassCfg(model.packer, mode, use_cand, use_rgb, band_weights(model.xyz_L, progress, model.c2f),
                       band_weights(model.dir_L, progress, model.c2f), rgb_joint=not encode_feat)
        E_s, G_c, sum_sf, t_w, c_dep, s_dep, rgb_map, w_all, w_s, rgbj_map = _FieldPass.apply(
            rays_o, rays_d, zz, c_rows, a_rows, model.packed(), cfg)
        if sched_mult < 1:
            if not model.encode_candidate:  # rendering.py:134-150
                results[f"s_weights_{typ}"] = w_s
                results[f"feat_{typ}"] = _project_feat(model, E_s, sum_sf)
            else:  # rendering.py:151-182
                results[f"c_weights_{typ}"] = w_all
                results[f"c_depth_{typ}"] = c_dep
                if encode_feat:
                    results[f"feat_{typ}"] = _project_feat(model, E_s, sum_sf, G_c, t_w)
                else:  # rendering.py:177-190: c_rgb = sum w_sj s_rgb + sum w_cj c_rgb, the candidate half projected per ray
                    results[f"c_rgb_{typ}"] = rgbj_map + _project_rgb(model, G_c, t_w)
                results[f"t_weight_{typ}"] = t_w
        if sched_mult > 0:  # rendering.py:195-209
            results[f"s_weights_{typ}"] = w_s
            results[f"s_rgb_{typ}"] = rgb_map
        results[f"s_depth_{typ}"] = s_dep  # rendering.py:211-218

    keep = kwargs.get("keep")
    if keep is not None:
        keep["z_coarse"] = z
    coarse_sigma_only = bool(kwargs.get("coarse_sigma_only")) and N_importance > 0
    if coarse_sigma_only and models["nerf_fine"].encode_candidate and sched_mult < 1:
        raise ValueError("coarse_sigma_only needs a schedule phase that resamples from the shared weights only "
                         "(sched_mult == 1 or no candidate head): c_weights_coarse come from the candidate head")
    inference(models["nerf_coarse"], z, sigma_only=coarse_sigma_only)
    if N_importance > 0:
        model = models["nerf_fine"]
        det = perturb == 0
        S = N_samples + N_importance
        zf = _empty(R, S, device=dev)
        zf[:, :N_samples] = z
        z_inject = kwargs.get("z_fine")

        def resample(key, n, col0):
            if z_inject is not None:
                return
            sample_pdf(z, results[key].detach().contiguous(), n, det, zf, col0, None if det else draw(n))

        if model.encode_candidate:  # rendering.py:267-300
            if sched_mult == 0:
                resample("c_weights_coarse", N_importance, N_samples)
            elif 0 < sched_mult < 1:
                n_s = round(sched_mult * N_importance)  # Python banker's rounding, like the reference (Q6)
                resample("c_weights_coarse", N_importance - n_s, N_samples + n_s)
                resample("s_weights_coarse", n_s, N_samples)
            elif sched_mult == 1:
                resample("s_weights_coarse", N_importance, N_samples)
        else:  # rendering.py:300-307
            resample("s_weights_coarse", N_importance, N_samples)
        if z_inject is not None:
            if tuple(z_inject.shape) != (R, S):
                raise ValueError(f"z_fine has shape {tuple(z_inject.shape)}, expected {(R, S)}")
            zf.copy_(z_inject.detach().to(dev, torch.float32))
        else:
            check(lib.upnerf_sort_rows(R, S, ptr(zf), st), "upnerf_sort_rows")
        if keep is not None:
            keep["z_fine"] = zf
        inference(model, zf)
    return results
