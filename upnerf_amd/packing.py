"""Parameter packing for the fused field kernels.

The HIP kernels read every matrix of one NeRF from ONE flat fp32 buffer `P` (row-major [N][Kp], K padded to a
multiple of 8, see include/upnerf_hip.h:upnerf_layout) and the backward chain reads transposed copies from `PT`.
`pack()` builds P from the module's parameters with differentiable torch ops (pad / cat / one small matmul for the
folded colour layer), so the gradient the kernels write in P's layout flows back to parameters that keep the
reference's names and shapes (models/nerf.py:39-78) -- state_dicts stay interchangeable with the reference.

Folding: rgb_share_layer.0 consumes cat[s_feat, PE(dir), a] with s_feat = W_f e + b_f, so
  W_r1[:, :F] s_feat = (W_r1[:, :F] W_f) e + W_r1[:, :F] b_f
and the kernels only ever see the [W/2][W] product; autograd differentiates the product."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from . import zero_pool
from ._lib import AUXK, CK, MAX_D, X0, Frag16Desc, FragDesc, Layout


def _align(n, a=4):
    return (n + a - 1) // a * a


class _PackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packer, names, *tensors):
        from ._lib import PackDesc, check, lib, ptr, stream
        L, W, W2, Fd = packer.L, packer.W, packer.W2, packer.feat_dim
        p = {n: t.detach().contiguous() for n, t in zip(names, tensors)}
        dev = tensors[0].device
        if not packer.encode_feat:  # no feature layer to fold into the colour head: the pack launch is the whole job
            buf = zero_pool.zeros(L.total, dev)
            descs = packer._pack_descs(p, lambda n: p[n].data_ptr())
            arr = (PackDesc * len(descs))(*descs)
            check(lib.upnerf_pack(ptr(buf), arr, len(descs), 0, stream()), "upnerf_pack")
            ctx.packer, ctx.names, ctx.buf = packer, names, None
            ctx.shapes = [tuple(t.shape) for t in tensors]
            return buf
        buf = zero_pool.zeros(L.total + W2 * Fd, dev)  # P followed by a copy of W_r1[:, :F]
        st = stream()
        descs = packer._pack_descs(p, lambda n: p[n].data_ptr(), L.total)
        arr = (PackDesc * len(descs))(*descs)
        check(lib.upnerf_pack(ptr(buf), arr, len(descs), 0, st), "upnerf_pack")
        base = buf.data_ptr()
        wrF = base + 4 * L.total
        # folded colour layer: W_r1[:, :F] . W_feat -> wr1[:, :W];  W_r1[:, :F] . b_feat + b_r1 -> br1
        check(lib.upnerf_linear(W2, W, Fd, wrF, Fd, ptr(p["feat_share_layer.weight"]), W, None, base + 4 * L.wr1, W + AUXK, 2,
                                st), "upnerf_linear")
        # W_r1[:, :F] . b_feat + b_r1 -> br1: one small matrix-vector launch straight into P (r3 ADVICE: rocBLAS gemv made the
        # summation order -- and with it the bitwise replay == eager contract -- depend on the rocBLAS version, and created its
        # handle inside a capture)
        check(lib.upnerf_matvec(W2, Fd, wrF, Fd, ptr(p["feat_share_layer.bias"]), ptr(p["rgb_share_layer.0.bias"]), base + 4 * L.br1, 0,
                                st), "upnerf_matvec")
        ctx.packer, ctx.names, ctx.buf = packer, names, buf
        ctx.save_for_backward(p["feat_share_layer.weight"], p["feat_share_layer.bias"])
        ctx.shapes = [tuple(t.shape) for t in tensors]
        return buf[:L.total]

    @staticmethod
    def backward(ctx, dP):
        from ._lib import PackDesc, check, lib, ptr, stream
        from .ops import wgrad_blocks_into, wgrad_into
        packer, names, buf = ctx.packer, ctx.names, ctx.buf
        L, W, W2, Fd = packer.L, packer.W, packer.W2, packer.feat_dim
        dP = dP.contiguous()
        dev = dP.device
        st = stream()
        # one flat buffer for all parameter gradients
        sizes = [math.prod(s) for s in ctx.shapes]
        flat = zero_pool.zeros(sum(sizes), dev)
        grads, off = {}, 0
        for n, s_, k in zip(names, ctx.shapes, sizes):
            grads[n] = flat[off:off + k].view(s_)
            off += k
        descs = packer._pack_descs(grads, lambda n: grads[n].data_ptr())
        arr = (PackDesc * len(descs))(*descs)
        check(lib.upnerf_pack(ptr(dP), arr, len(descs), 1, st), "upnerf_pack")
        if not packer.encode_feat:
            return (None, None) + tuple(grads[n] for n in names)
        feat_w, feat_b = ctx.saved_tensors
        base = dP.data_ptr()
        gbr1 = dP[L.br1:L.br1 + W2]
        g_wr = grads["rgb_share_layer.0.weight"]          # [W2][F + 27 + A]
        in_rgb = g_wr.shape[1]
        # d W_r1[:, :F] = gfold . W_feat^T + gbr1 (x) b_feat
        check(lib.upnerf_linear(W2, Fd, W, base + 4 * L.wr1, W + AUXK, ptr(feat_w), W, None, ptr(g_wr), in_rgb, 0, st),
              "upnerf_linear")
        # (the second term, g_wr[:, :F] += gbr1 (x) b_feat, rides on the matrix-vector launch below: upnerf_matvec_rank1)
        # d W_feat = W_r1[:, :F]^T . gfold ; d b_feat = W_r1[:, :F]^T . gbr1 ; d b_r1 = gbr1
        wrF = buf[L.total:].view(W2, Fd)
        g_fw = grads["feat_share_layer.weight"]
        if Fd > 256:
            wgrad_blocks_into(W2, wrF, Fd, Fd, dP, W + AUXK, W, g_fw.data_ptr(), W, None, dev, b_off=L.wr1)
        else:
            wgrad_into(W2, wrF, Fd, Fd, dP, W + AUXK, W, g_fw.data_ptr(), W, None, dev, b_off=L.wr1)
        check(lib.upnerf_matvec_rank1(W2, Fd, ptr(wrF), Fd, ptr(gbr1), ptr(grads["feat_share_layer.bias"]), ptr(g_wr), in_rgb,
                                      ptr(feat_b), st), "upnerf_matvec_rank1")
        return (None, None) + tuple(grads[n] for n in names)


class NerfPacker:
    def __init__(self, W: int, D: int, skips, in_xyz: int, in_dir: int, feat_dim: int, appearance_dim: int,
                 candidate_dim: int, encode_feat: bool = True):
        # encode_feat = False (models/nerf.py:52-56, 110-123): the colour head reads xyz_encoding_final itself, so the
        # "folded" first colour matrix of the layout is W_r1[:, :W] as it stands -- same layout, no fold launches
        self.encode_feat = bool(encode_feat)
        if W not in (64, 256):
            raise ValueError(f"HIP field kernels support W in (64, 256), got {W}")
        if not 1 <= D <= MAX_D:
            raise ValueError(f"HIP field kernels support 1 <= D <= {MAX_D}, got {D}")
        if in_xyz != 63:
            raise ValueError("HIP field kernels are specialised for N_emb_xyz = 10 (63-wide encoding)")
        if in_dir != 27:
            raise ValueError("HIP field kernels are specialised for N_emb_dir = 4 (27-wide encoding)")
        if appearance_dim not in (0, 48) or candidate_dim not in (0, CK):
            raise ValueError("HIP field kernels support appearance_dim in (0, 48) and candidate_dim in (0, 16)")
        active = [s for s in skips if 0 < s < D]
        if len(active) > 1:
            raise ValueError("at most one active skip connection is supported")
        self.W, self.D, self.W2 = W, D, W // 2
        self.skip = active[0] if active else -1
        self.feat_dim, self.A, self.C, self.in_dir = feat_dim, appearance_dim, candidate_dim, in_dir
        self.has_cand = candidate_dim > 0
        L = Layout()
        L.W, L.D, L.skip = W, D, self.skip
        off = 0

        def take(n):
            nonlocal off
            o = off
            off += _align(n)
            return o

        W2 = self.W2
        for l in range(D):
            k = X0 if l == 0 else (X0 + W if l == self.skip else W)
            L.w[l] = take(W * k)
        for l in range(D):
            L.b[l] = take(W)
        L.we, L.be = take(W * W), take(W)
        L.wsig, L.bsig = take(W), take(4)
        L.wc1, L.bc1 = take(W2 * (W + CK)), take(W2)
        L.wc2, L.bc2 = take(W2 * W2), take(W2)
        L.wcsig, L.bcsig = take(W2), take(4)
        L.wr1, L.br1 = take(W2 * (W + AUXK)), take(W2)
        L.wr2, L.br2 = take(4 * W2), take(4)
        L.total = off
        off = 0
        for l in range(D):
            L.t_w[l] = take((X0 if l == 0 else W) * W)
        L.t_skipx = take(X0 * W)
        L.t_we = take(W * W)
        L.t_head = take(W * W)
        L.t_wc2 = take(W2 * W2)
        L.t_total = off
        self.L = L

    # ------------------------------------------------------------------ forward-form buffer
    def pack(self, p: Dict[str, torch.Tensor]) -> torch.Tensor:
        """p: name -> parameter (reference names, e.g. 'xyz_encoding_1.0.weight').  Returns P [L.total]."""
        W, W2, D, L = self.W, self.W2, self.D, self.L
        dev, dt = p["xyz_encoding_1.0.weight"].device, torch.float32
        pieces = []

        def put(t, n):
            t = t.reshape(-1)
            pad = _align(n) - t.numel()
            assert t.numel() == n and pad >= 0
            pieces.append(F.pad(t, (0, pad)) if pad else t)

        for l in range(D):
            w = p[f"xyz_encoding_{l + 1}.0.weight"]
            if l == 0:
                put(F.pad(w, (0, 1)), W * X0)
            elif l == self.skip:
                put(torch.cat([F.pad(w[:, :63], (0, 1)), w[:, 63:]], 1), W * (X0 + W))
            else:
                put(w, W * W)
        for l in range(D):
            put(p[f"xyz_encoding_{l + 1}.0.bias"], W)
        put(p["xyz_encoding_final.weight"], W * W)
        put(p["xyz_encoding_final.bias"], W)
        put(p["share_sigma.0.weight"], W)
        put(F.pad(p["share_sigma.0.bias"], (0, 3)), 4)
        if self.has_cand:
            put(p["candidate_encoding.0.weight"], W2 * (W + CK))
            put(p["candidate_encoding.0.bias"], W2)
            put(p["candidate_encoding.2.weight"], W2 * W2)
            put(p["candidate_encoding.2.bias"], W2)
            put(p["candidate_sigma.0.weight"], W2)
            put(F.pad(p["candidate_sigma.0.bias"], (0, 3)), 4)
        else:
            for n in (W2 * (W + CK), W2, W2 * W2, W2, W2, 4):
                put(torch.zeros(n, device=dev, dtype=dt), n)
        Fd = self.feat_dim if self.encode_feat else W
        wr = p["rgb_share_layer.0.weight"]  # [W2][F + 27 + A]  (encode_feat = False: [W2][W + 27 + A])
        fold = wr[:, :Fd] @ p["feat_share_layer.weight"] if self.encode_feat else wr[:, :W]  # [W2][W]
        aux_w = F.pad(wr[:, Fd:], (0, AUXK - (self.in_dir + self.A))) if self.A else \
            F.pad(wr[:, Fd:], (0, AUXK - self.in_dir))
        put(torch.cat([fold, aux_w], 1), W2 * (W + AUXK))
        put(wr[:, :Fd] @ p["feat_share_layer.bias"] + p["rgb_share_layer.0.bias"] if self.encode_feat
            else p["rgb_share_layer.0.bias"], W2)
        put(F.pad(p["rgb_share_layer.2.weight"], (0, 0, 0, 1)), 4 * W2)
        put(F.pad(p["rgb_share_layer.2.bias"], (0, 1)), 4)
        P = torch.cat(pieces)
        assert P.numel() == L.total
        return P

    # ------------------------------------------------------------------ HIP pack (product path on the GPU)
    def _pack_descs(self, p: Dict[str, torch.Tensor], base_of, scratch_off=None):
        """Copy descriptors of pack().  `base_of(name)` gives the device pointer a descriptor starts from (the parameter,
        or its gradient buffer for the backward); `scratch_off`: where the contiguous copy of W_r1[:, :F] goes (forward
        only; its gradient comes from the fold, not from a copy)."""
        from ._lib import PackDesc
        W, W2, D, L, Fd = self.W, self.W2, self.D, self.L, self.feat_dim
        out = []

        def add(name, col0, rows, cols, dst_off, dst_ld, acc=0):
            t = p[name]
            ld = t.shape[1] if t.dim() == 2 else t.numel()
            out.append(PackDesc(base_of(name) + 4 * col0, rows, cols, ld if t.dim() == 2 else cols, dst_off, dst_ld, acc))

        for l in range(D):
            n = f"xyz_encoding_{l + 1}.0"
            if l == 0:
                add(n + ".weight", 0, W, 63, L.w[0], X0)
            elif l == self.skip:
                add(n + ".weight", 0, W, 63, L.w[l], X0 + W)
                add(n + ".weight", 63, W, W, L.w[l] + X0, X0 + W)
            else:
                add(n + ".weight", 0, W, W, L.w[l], W)
            add(n + ".bias", 0, 1, W, L.b[l], W)
        add("xyz_encoding_final.weight", 0, W, W, L.we, W)
        add("xyz_encoding_final.bias", 0, 1, W, L.be, W)
        add("share_sigma.0.weight", 0, 1, W, L.wsig, W)
        add("share_sigma.0.bias", 0, 1, 1, L.bsig, 1)
        if self.has_cand:
            add("candidate_encoding.0.weight", 0, W2, W + CK, L.wc1, W + CK)
            add("candidate_encoding.0.bias", 0, 1, W2, L.bc1, W2)
            add("candidate_encoding.2.weight", 0, W2, W2, L.wc2, W2)
            add("candidate_encoding.2.bias", 0, 1, W2, L.bc2, W2)
            add("candidate_sigma.0.weight", 0, 1, W2, L.wcsig, W2)
            add("candidate_sigma.0.bias", 0, 1, 1, L.bcsig, 1)
        naux = self.in_dir + self.A
        if not self.encode_feat:  # W_r1 = [W | PE(dir) | appearance] columns go where the folded matrix and its side columns sit
            add("rgb_share_layer.0.weight", 0, W2, W + naux, L.wr1, W + AUXK)
            add("rgb_share_layer.0.bias", 0, 1, W2, L.br1, W2)
        else:
            add("rgb_share_layer.0.weight", Fd, W2, naux, L.wr1 + W, W + AUXK)   # [PE(dir) | appearance] columns
            if scratch_off is not None:
                add("rgb_share_layer.0.weight", 0, W2, Fd, scratch_off, Fd)       # contiguous copy of W_r1[:, :F]
            else:  # backward: d b_r1 = d br1 (forward, br1 = W_r1[:, :F] b_feat + b_r1 is the mat-vec's output, not a copy)
                add("rgb_share_layer.0.bias", 0, 1, W2, L.br1, W2)
        add("rgb_share_layer.2.weight", 0, 3, W2, L.wr2, W2)
        add("rgb_share_layer.2.bias", 0, 1, 3, L.br2, 3)
        return out

    PACK_NAMES_BASE = ["share_sigma.0.weight", "share_sigma.0.bias", "xyz_encoding_final.weight", "xyz_encoding_final.bias",
                       "feat_share_layer.weight", "feat_share_layer.bias", "rgb_share_layer.0.weight",
                       "rgb_share_layer.0.bias", "rgb_share_layer.2.weight", "rgb_share_layer.2.bias"]
    PACK_NAMES_CAND = ["candidate_encoding.0.weight", "candidate_encoding.0.bias", "candidate_encoding.2.weight",
                       "candidate_encoding.2.bias", "candidate_sigma.0.weight", "candidate_sigma.0.bias"]

    def pack_names(self):
        base = [n for n in self.PACK_NAMES_BASE if self.encode_feat or not n.startswith("feat_share_layer")]
        names = [f"xyz_encoding_{l + 1}.0.{k}" for l in range(self.D) for k in ("weight", "bias")] + base
        return names + (self.PACK_NAMES_CAND if self.has_cand else [])

    def pack_hip(self, p: Dict[str, torch.Tensor]) -> torch.Tensor:
        """pack() with HIP kernels and a hand-written backward: 4 launches instead of ~45 (and ~10 instead of ~50 in the
        backward).  Same layout; the folded colour matrix is produced by upnerf_linear (exact fp32 MFMA)."""
        names = self.pack_names()
        return _PackFn.apply(self, names, *[p[n] for n in names])

    # ------------------------------------------------------------------ one-launch HIP re-layout (product path)
    def _descs(self):
        """(forward descriptors, transposed descriptors) for upnerf_frag_copy, built once."""
        if getattr(self, "_desc_cache", None) is None:
            W, W2, D, L = self.W, self.W2, self.D, self.L
            fwd, bwd = [], []
            kp = lambda l: X0 if l == 0 else (X0 + W if l == self.skip else W)
            for l in range(D):
                fwd.append((L.w[l], kp(l), 0, W, kp(l), L.w[l], kp(l), 0))
            fwd.append((L.we, W, 0, W, W, L.we, W, 0))
            fwd.append((L.wc1, W + CK, 0, W2, W + CK, L.wc1, W + CK, 0))
            fwd.append((L.wc2, W2, 0, W2, W2, L.wc2, W2, 0))
            fwd.append((L.wr1, W + AUXK, 0, W2, W + AUXK, L.wr1, W + AUXK, 0))
            for l in range(D):
                if l == 0:
                    bwd.append((L.w[0], X0, 1, X0, W, L.t_w[0], W, 0))
                elif l == self.skip:
                    bwd.append((L.w[l], X0 + W, 1, X0, W, L.t_skipx, W, 0))
                    bwd.append((L.w[l] + X0, X0 + W, 1, W, W, L.t_w[l], W, 0))
                else:
                    bwd.append((L.w[l], W, 1, W, W, L.t_w[l], W, 0))
            bwd.append((L.we, W, 1, W, W, L.t_we, W, 0))
            bwd.append((L.wr1, W + AUXK, 1, W, W2, L.t_head, W, 0))
            bwd.append((L.wc1, W + CK, 1, W, W2, L.t_head, W, W2))
            bwd.append((L.wc2, W2, 1, W2, W2, L.t_wc2, W2, 0))
            mk = lambda lst: (FragDesc * len(lst))(*[FragDesc(*t) for t in lst])
            self._desc_cache = (mk(fwd), len(fwd), mk(bwd), len(bwd))
        return self._desc_cache

    @torch.no_grad()
    def frag_hip(self, P: torch.Tensor) -> torch.Tensor:
        """Kernel-side copy of P (matrices in fragment order) with one HIP launch."""
        from ._lib import check, lib, ptr, stream
        fd, nf, _, _ = self._descs()
        out = P.clone()
        check(lib.upnerf_frag_copy(ptr(P), ptr(out), fd, nf, stream()), "upnerf_frag_copy")
        return out

    @torch.no_grad()
    def frag_t_hip(self, P: torch.Tensor) -> torch.Tensor:
        """Fragment-ordered transposed copies (layout t_*) straight from the row-major P, one HIP launch."""
        from ._lib import check, lib, ptr, stream
        _, _, bd, nb = self._descs()
        out = torch.zeros(self.L.t_total, device=P.device, dtype=P.dtype)
        check(lib.upnerf_frag_copy(ptr(P), ptr(out), bd, nb, stream()), "upnerf_frag_copy")
        return out

    # ------------------------------------------------------------------ f16x3 re-layout (product path, W = 256)
    EXP_FINAL, EXP_C1, EXP_C2, EXP_R1, EXP_HEAD_T = 8, 9, 10, 11, 12  # exponent ids; trunk layer l uses id l

    def _descs16(self):
        if getattr(self, "_desc16_cache", None) is None:
            D = self.D
            fd, nf, bd, nb = self._descs()
            fid = list(range(D)) + [self.EXP_FINAL, self.EXP_C1, self.EXP_C2, self.EXP_R1]
            bid = []
            for l in range(D):
                bid += [l, l] if l == self.skip else [l]
            bid += [self.EXP_FINAL, self.EXP_HEAD_T, self.EXP_HEAD_T, self.EXP_C2]
            assert len(fid) == nf and len(bid) == nb

            def mk(src, n, ids):
                fields = [f for f, _ in FragDesc._fields_]
                return (Frag16Desc * n)(*[Frag16Desc(*[getattr(src[i], f) for f in fields], ids[i]) for i in range(n)])

            self._desc16_cache = (mk(fd, nf, fid), nf, mk(bd, nb, bid), nb)
        return self._desc16_cache

    @torch.no_grad()
    def frag16_hip(self, P: torch.Tensor, perm: bool = False):
        """(P16, PT16, wexp, wnorm): every matrix of P (and its transposed copy) as scaled fp16 (hi, lo) MFMA fragments with
        one power-of-two exponent per matrix id -- what upnerf_field_fwd_f16x3 / upnerf_field_bwd_f16x3 read.  perm: both
        sets in the k order of the register-resident kernels (csrc/field16rr.hip: include/upnerf_hip.h, tile_rows = 256), plus
        the row 1-norms they bound their exponents with (wnorm [64]: forward matrices at 0.., transposed ones at 32..)."""
        from ._lib import check, lib, ptr, stream
        fd, nf, bd, nb = self._descs16()
        t0 = (self.L.total + 63) // 64 * 64  # the transposed set starts on a 256-byte boundary
        both = zero_pool.zeros(t0 + self.L.t_total, P.device)  # one fill for the two sets
        P16, PT16 = both[:self.L.total], both[t0:]
        scratch = torch.empty(16, device=P.device, dtype=torch.float32)
        wexp = torch.empty(16, device=P.device, dtype=torch.int32)
        wnorm = torch.empty(64, device=P.device, dtype=torch.float32) if perm else None
        check(lib.upnerf_frag16(ptr(P), ptr(P16), ptr(PT16), fd, nf, bd, nb, ptr(scratch), ptr(wexp), int(perm), int(perm),
                                ptr(wnorm), stream()), "upnerf_frag16")
        return P16, PT16, wexp, wnorm

    # ------------------------------------------------------------------ MFMA fragment order (no grad)
    @staticmethod
    def _frag_into(dst, src, off, n, kp):
        """dst[off:off+n*kp] = fragment-ordered copy of the row-major [n][kp] matrix at src[off:]:
        [n/32][kp/8][half(2)][lane%32][4]  (csrc/common.cuh:mma_lds)."""
        v = src[off:off + n * kp].view(n // 32, 32, kp // 8, 2, 4).permute(0, 2, 3, 1, 4)
        dst[off:off + n * kp].view(n // 32, kp // 8, 2, 32, 4).copy_(v)

    @torch.no_grad()
    def frag(self, P: torch.Tensor) -> torch.Tensor:
        """Kernel-side copy of P: matrices in fragment order, vectors (biases, 1/3-wide heads) unchanged."""
        W, W2, D, L = self.W, self.W2, self.D, self.L
        out = P.clone()
        for l in range(D):
            self._frag_into(out, P, L.w[l], W, X0 if l == 0 else (X0 + W if l == self.skip else W))
        self._frag_into(out, P, L.we, W, W)
        self._frag_into(out, P, L.wc1, W2, W + CK)
        self._frag_into(out, P, L.wc2, W2, W2)
        self._frag_into(out, P, L.wr1, W2, W + AUXK)
        return out

    @torch.no_grad()
    def frag_t(self, PT: torch.Tensor) -> torch.Tensor:
        W, W2, D, L = self.W, self.W2, self.D, self.L
        out = PT.clone()
        for l in range(D):
            self._frag_into(out, PT, L.t_w[l], X0 if l == 0 else W, W)
        self._frag_into(out, PT, L.t_skipx, X0, W)
        self._frag_into(out, PT, L.t_we, W, W)
        self._frag_into(out, PT, L.t_head, W, W)
        self._frag_into(out, PT, L.t_wc2, W2, W2)
        return out

    # ------------------------------------------------------------------ transposed copies (no grad)
    @torch.no_grad()
    def pack_t(self, P: torch.Tensor) -> torch.Tensor:
        W, W2, D, L = self.W, self.W2, self.D, self.L
        PT = torch.empty(L.t_total, device=P.device, dtype=P.dtype)

        def mat(o, n, k):
            return P[o:o + n * k].view(n, k)

        def putT(o, m):  # m: [rows][cols] written row-major at PT[o:]
            PT[o:o + m.numel()] = m.reshape(-1)

        for l in range(D):
            if l == 0:
                putT(L.t_w[0], mat(L.w[0], W, X0).t())
            elif l == self.skip:
                w = mat(L.w[l], W, X0 + W)
                putT(L.t_skipx, w[:, :X0].t())
                putT(L.t_w[l], w[:, X0:].t())
            else:
                putT(L.t_w[l], mat(L.w[l], W, W).t())
        if self.skip < 0:
            PT[L.t_skipx:L.t_skipx + X0 * W] = 0
        putT(L.t_we, mat(L.we, W, W).t())
        wr1 = mat(L.wr1, W2, W + AUXK)[:, :W]
        wc1 = mat(L.wc1, W2, W + CK)[:, :W]
        putT(L.t_head, torch.cat([wr1.t(), wc1.t()], 1))
        putT(L.t_wc2, mat(L.wc2, W2, W2).t())
        return PT
