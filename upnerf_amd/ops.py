"""Thin torch.autograd wrappers over single C-ABI entry points (linear layer, weight gradients, Adam).

PyTorch is used here for device memory, streams and autograd bookkeeping only; the arithmetic is in
libupnerf_hip.so.  Nothing in this file falls back to ATen matmuls."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, lib, ptr, stream

_WS: Dict[Tuple[str, int], torch.Tensor] = {}


class KernelTimer:
    """Per-kernel device timing with HIP events recorded on the stream the kernels are enqueued on (torch's current
    stream).  Disabled by default (zero overhead); bench.py enables it over the timed region to obtain the average
    launch duration of each hot kernel for the roofline figures."""

    def __init__(self):
        self.enabled = False
        self.only = None  # None = every instrumented kernel class; else the set of names to record
        self.records = {}

    def run(self, name: str, fn, units: int = 0):
        if not self.enabled or (self.only is not None and name not in self.only):
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.records.setdefault(name, []).append((e0, e1, units))
        return out

    def reset(self):
        self.records = {}

    def summary(self):
        """name -> dict(launches, total_ms, avg_ms, units_per_launch); synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, evs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            out[name] = dict(launches=len(ms), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                             units_per_launch=sum(u for _, _, u in evs) / len(evs))
        return out


TIMER = KernelTimer()


def workspace(tag: str, numel: int, device) -> torch.Tensor:
    """Grow-only fp32 scratch, one per (tag, device); stream-ordered reuse."""
    key = (tag, device.index if device.index is not None else torch.cuda.current_device())
    t = _WS.get(key)
    if t is None or t.numel() < numel:
        t = torch.empty(max(numel, 1), device=device, dtype=torch.float32)
        _WS[key] = t
    return t


def nsplit_for(M: int) -> int:
    """M-splits of the weight-gradient kernels: one workgroup per CU for the big per-sample reductions, and at
    least 64 rows per split so that the small per-ray ones (M = rays) are not a serial chain of chunks."""
    return max(1, min(256, M // 64))


def wgrad_into(M: int, A: torch.Tensor, lda: int, N: int, B: torch.Tensor, ldb: int, K: int, dW_ptr: int, ldo: int,
               db_ptr: Optional[int], device, a_off: int = 0, b_off: int = 0):
    """dW[N][ldo] = sum_m A[m][a_off + n] B[m][b_off + k];  db[n] = sum_m A[m][a_off+n]  (pointers may be offsets
    into a larger gradient buffer)."""
    ns = nsplit_for(M)
    ws = workspace("wgrad", ns * (256 * 256 + 256), device)
    rc = TIMER.run(f"wgrad_{N}x{K}", lambda: lib.upnerf_wgrad(M, A.data_ptr() + 4 * a_off, lda, N,
                                                              B.data_ptr() + 4 * b_off, ldb, K, dW_ptr, ldo, db_ptr,
                                                              ptr(ws), ns, stream()), units=M)
    check(rc, "upnerf_wgrad")


def wgrad_blocks_into(M: int, A: torch.Tensor, lda: int, N: int, B: torch.Tensor, ldb: int, K: int, dW_ptr: int, ldo: int,
                      db_ptr: Optional[int], device, a_off: int = 0, b_off: int = 0):
    """wgrad_into for a small-M problem whose N or K exceeds one 256 x 256 block: ONE grouped launch + ONE reduction
    (upnerf_wgrad_grouped cuts any N x K into 128 x 128 blocks) instead of a launch pair per block."""
    g = _lib.WgradGroup(A=A.data_ptr() + 4 * a_off, B=B.data_ptr() + 4 * b_off, dW=dW_ptr, db=db_ptr, M=M, N=N, K=K, lda=lda,
                        ldb=ldb, ldo=ldo)
    arr = (_lib.WgradGroup * 1)(g)
    nsplit = max(1, min(_DeferredWgrads.NSPLIT, M // 64))
    n = lib.upnerf_wgrad_grouped_scratch(arr, 1, nsplit)
    if n < 0:
        check(n, "upnerf_wgrad_grouped_scratch")
    ws = workspace("wgrad_blocks", n, device)
    check(TIMER.run(f"wgrad_blocks_{N}x{K}", lambda: lib.upnerf_wgrad_grouped(arr, 1, ptr(ws), nsplit, stream())),
          "upnerf_wgrad_grouped")


class _DeferredWgrads:
    """Small (per-ray, M = rays) weight gradients of HipLinear, collected during a backward pass and computed by ONE
    grouped launch + ONE reduction when the autograd engine finishes that pass (`queue_callback`, the hook DDP's reducer
    uses), instead of a 25 us launch pair per layer.  The gradient tensors are handed to autograd right away and filled
    at the flush, so nothing may read them earlier: only call sites whose weight has NO other consumer in the graph ask
    for it (`hip_linear(..., defer_wgrad=True)`: TransientNet, the candidate feature projection) -- autograd would sum a
    second contribution into the still empty tensor (feat_share_layer.weight also feeds the packed colour weights and
    therefore stays on the immediate path) -- and their AccumulateGrad only adopts the tensor (gradients are reset to
    None every step).  `enabled = False` restores one launch per layer everywhere."""

    enabled = __import__("os").environ.get("UPNERF_DEFER_WGRADS", "1") != "0"  # diagnostic switch for A/B runs
    check_adopted = __import__("os").environ.get("UPNERF_CHECK_DEFERRED", "0") == "1"  # debug: verify at every flush
    MAX_M = 16384  # larger problems go to upnerf_wgrad directly (they fill the GPU on their own)
    NSPLIT = 64

    def __init__(self):
        self.groups, self.keep, self.owners = [], [], []

    def clear(self):
        """Drop whatever a failed backward pass left behind (its end-of-pass callback never ran)."""
        self.groups, self.keep, self.owners = [], [], []

    def add(self, M, gy, lda, N, x, ldb, K, gw, gb, owners=()):
        self.owners += [(p, t.data_ptr()) for p, t in owners if t is not None]
        # one callback per entry: the first to run launches everything pending, the others find nothing (no state that
        # an exception inside a backward pass could leave behind; `keep` holds every tensor a pending group points at)
        torch.autograd.Variable._execution_engine.queue_callback(self.flush)
        self.groups.append(_lib.WgradGroup(A=gy.data_ptr(), B=x.data_ptr(), dW=gw.data_ptr(),
                                           db=None if gb is None else gb.data_ptr(), M=M, N=N, K=K, lda=lda, ldb=ldb,
                                           ldo=K))
        # alive until the grouped kernels have been enqueued.  For the outputs only the STORAGE is held: a second
        # reference to the tensor itself would make AccumulateGrad copy it (before it is filled) instead of adopting it
        self.keep.append((gy, x, gw.untyped_storage(), None if gb is None else gb.untyped_storage()))
        if len(self.groups) == _lib.MAX_WGRAD_GROUPS:
            self._launch()

    def _launch(self):
        groups, self.groups = self.groups, []
        keep, self.keep = self.keep, []
        owners, self.owners = self.owners, []
        if not groups:
            return
        if self.check_adopted:
            for p, addr in owners:
                if p.grad is None or p.grad.data_ptr() != addr:
                    raise RuntimeError("deferred weight gradient was not adopted by its parameter (autograd copied or "
                                       "summed the still empty buffer)")
        arr = (_lib.WgradGroup * len(groups))(*groups)
        n = lib.upnerf_wgrad_grouped_scratch(arr, len(groups), self.NSPLIT)
        if n < 0:
            check(n, "upnerf_wgrad_grouped_scratch")
        ws = workspace("wgrad_grouped", n, keep[0][0].device)
        check(TIMER.run("wgrad_grouped", lambda: lib.upnerf_wgrad_grouped(arr, len(groups), ptr(ws), self.NSPLIT,
                                                                          stream())), "upnerf_wgrad_grouped")

    def flush(self):
        self._launch()


DEFERRED_WGRADS = _DeferredWgrads()


def can_adopt(p) -> bool:
    """True when autograd's AccumulateGrad will ADOPT a freshly allocated gradient tensor for the leaf `p` instead of
    reading it (p.grad += g, a clone for a hook, ...): the precondition for handing it a buffer that is filled only at
    the end of the backward pass.  Anything else takes the immediate path."""
    return (p is not None and isinstance(p, torch.nn.Parameter) and p.is_leaf and p.grad is None
            and not p._backward_hooks and not getattr(p, "_post_accumulate_grad_hooks", None))


_DEFER_USES: Dict[int, list] = {}  # id(leaf) -> [forward uses with deferral requested since the last reset]


def _defer_token(*leaves):
    """Per-step use counters of the leaves a deferring call site consumes: a leaf that two deferring sites consume in
    the same graph would have its two gradients SUMMED by autograd before AccumulateGrad sees them (i.e. read while still
    empty), so the backward defers only while every counter is 1."""
    toks = []
    for p in leaves:
        if p is not None:
            t = _DEFER_USES.setdefault(id(p), [0])
            t[0] += 1
            toks.append(t)
    return toks


def reset_deferred():
    """Start of a training step: forget groups a failed backward pass may have left pending and the use counters."""
    DEFERRED_WGRADS.clear()
    DEFERRED_EMBEDS.items = []
    _DEFER_USES.clear()


def scale_exponents(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Device int32 [2]: power-of-two exponents that bring max|A|, max|B| to ~2^14 (no host sync)."""
    amax = torch.stack([A.abs().amax(), B.abs().amax()]).clamp_min(1e-30)
    return (14 - torch.ceil(torch.log2(amax))).to(torch.int32)


def wgrad_f16x3_into(M: int, A: torch.Tensor, lda: int, N: int, B: torch.Tensor, ldb: int, K: int, dW_ptr: int, ldo: int,
                     db_ptr: Optional[int], device, expo: Optional[torch.Tensor] = None, expo_a: Optional[int] = None,
                     expo_b: Optional[int] = None, a_off: int = 0, b_off: int = 0, planes: int = 2):
    """upnerf_wgrad with the contraction on the f16 matrix cores (3-term hi/lo split, fp32-level accuracy).
    Scale exponents: `expo` (device int32 [2]) or raw device pointers expo_a / expo_b; computed from A, B if absent."""
    ns = nsplit_for(M)
    ws = workspace("wgrad", ns * (256 * 256 + 256), device)
    if expo_a is None:
        if expo is None:
            expo = scale_exponents(A, B)
        expo_a, expo_b = expo.data_ptr(), expo.data_ptr() + 4
    rc = TIMER.run(f"wgrad16_{N}x{K}", lambda: lib.upnerf_wgrad_f16x3(M, A.data_ptr() + 4 * a_off, lda, N,
                                                                      B.data_ptr() + 4 * b_off, ldb, K, dW_ptr, ldo,
                                                                      db_ptr, ptr(ws), ns, expo_a, expo_b, planes,
                                                                      stream()),
                   units=M)
    check(rc, "upnerf_wgrad_f16x3")


class WgradChain:
    """A run of f16x3 weight gradients whose slab reductions ride on the NEXT launch (upnerf_wgrad_f16x3_chain): each kernel's
    first workgroups sum the previous problem's slabs before their own work, the last problem is summed by `finish()`.
    Nothing may read a gradient of the run before `finish()`; the slabs alternate between two workspaces."""

    def __init__(self, device):
        self.device, self.pending, self.flip = device, _lib.WgradPending(), 0

    def _slabs(self, ns):
        self.flip ^= 1
        return workspace(f"wgrad_chain{self.flip}", ns * (256 * 256 + 256 + 260), self.device)  # slabs + bias slabs + a riding vector head's

    def wgrad(self, M, A, lda, N, B, ldb, K, dW_ptr, ldo, db_ptr, expo_a, expo_b, a_off=0, b_off=0, planes=2, v=None, dv_ptr=None,
              dbv_ptr=None):
        """v / dv_ptr / dbv_ptr: a 1-wide head fed by the same B rows rides on the launch (upnerf_wgrad_f16x3_chain_v: dv[k] =
        sum_m v[m] B[m][k], dbv = sum_m v[m]) -- 256 x 256 problems of the f16x3 arithmetic only."""
        ns = nsplit_for(M)
        ws = self._slabs(ns)
        if v is not None:
            rc = TIMER.run(f"wgrad16_{N}x{K}", lambda: lib.upnerf_wgrad_f16x3_chain_v(
                M, A.data_ptr() + 4 * a_off, lda, N, B.data_ptr() + 4 * b_off, ldb, K, dW_ptr, ldo, db_ptr, ptr(v), dv_ptr, dbv_ptr,
                ptr(ws), ns, expo_a, expo_b, planes, C.byref(self.pending), stream()), units=M)
            check(rc, "upnerf_wgrad_f16x3_chain_v")
            return
        rc = TIMER.run(f"wgrad16_{N}x{K}", lambda: lib.upnerf_wgrad_f16x3_chain(
            M, A.data_ptr() + 4 * a_off, lda, N, B.data_ptr() + 4 * b_off, ldb, K, dW_ptr, ldo, db_ptr, ptr(ws), ns, expo_a, expo_b,
            planes, C.byref(self.pending), stream()), units=M)
        check(rc, "upnerf_wgrad_f16x3_chain")

    def wgrad2(self, M, A, lda, N, B, ldb, K, dW_ptr, ldo, db_ptr, n2, dW2_ptr, ldo2, db2_ptr, expo_a, expo_b, planes=2):
        """Rows [0, n2) of the result go to dW / db, rows [n2, N) to dW2 / db2 (two layers fed by the same B, A side by side)."""
        ns = nsplit_for(M)
        ws = self._slabs(ns)
        rc = TIMER.run(f"wgrad16_{N}x{K}", lambda: lib.upnerf_wgrad_f16x3_chain2(
            M, ptr(A), lda, N, ptr(B), ldb, K, dW_ptr, ldo, db_ptr, n2, dW2_ptr, ldo2, db2_ptr, ptr(ws), ns, expo_a, expo_b, planes,
            C.byref(self.pending), stream()), units=M)
        check(rc, "upnerf_wgrad_f16x3_chain2")

    def wgrad_p(self, M, A16, lda, aexp, N, B, ldb, bexp, K, dW_ptr, ldo, db_ptr, expo_a, expo_b, frag=False, n2=0, dW2_ptr=None,
                ldo2=0, db2_ptr=None, v=None, dv_ptr=None, dbv_ptr=None):
        """wgrad_f16p_into (fp16-stored operands of the f16 field mode) as a link of the run; n2 > 0: rows [n2, N) of the
        result go to dW2 / db2 (as wgrad2).  v / dv_ptr / dbv_ptr (fragment-ordered 256 x 256 problems): a 1-wide head fed by the
        same B rows rides on the launch (upnerf_wgrad_f16p_chain_v)."""
        ns = nsplit_for(M)
        ws = self._slabs(ns)
        if v is not None:
            assert frag and N == 256 and K == 256 and bexp is not None and n2 == 0
            rc = TIMER.run(f"wgrad16p_{N}x{K}", lambda: lib.upnerf_wgrad_f16p_chain_v(
                M, ptr(A16), ptr(aexp), ptr(B), ptr(bexp), dW_ptr, ldo, db_ptr, ptr(v), dv_ptr, dbv_ptr, ptr(ws), ns, expo_a, expo_b,
                C.byref(self.pending), stream()), units=M)
            check(rc, "upnerf_wgrad_f16p_chain_v")
            return
        rc = TIMER.run(f"wgrad16p_{N}x{K}", lambda: lib.upnerf_wgrad_f16p_chain(
            M, ptr(A16), lda, ptr(aexp), N, ptr(B), ldb, ptr(bexp), int(bexp is not None) | (2 if frag else 0), K, dW_ptr, ldo, db_ptr,
            n2, dW2_ptr, ldo2, db2_ptr, ptr(ws), ns, expo_a, expo_b, C.byref(self.pending), stream()), units=M)
        check(rc, "upnerf_wgrad_f16p_chain")

    def wgrad_p24(self, M, A16, Alo8, lda, aexp, N, B, Blo8, ldb, bexp, K, dW_ptr, ldo, db_ptr, expo_a, expo_b):
        """The "24-bit" operands of the f16x3 mode (hi fp16 + residual byte, upnerf_wgrad_f24p_chain) as a link of the run."""
        ns = nsplit_for(M)
        ws = self._slabs(ns)
        rc = TIMER.run(f"wgrad24p_{N}x{K}", lambda: lib.upnerf_wgrad_f24p_chain(
            M, ptr(A16), ptr(Alo8), lda, ptr(aexp), N, ptr(B), ptr(Blo8), ldb, ptr(bexp), int(bexp is not None), K, dW_ptr, ldo, db_ptr,
            ptr(ws), ns, expo_a, expo_b, C.byref(self.pending), stream()), units=M)
        check(rc, "upnerf_wgrad_f24p_chain")

    def finish(self):
        check(lib.upnerf_wgrad_finish(C.byref(self.pending), stream()), "upnerf_wgrad_finish")


def wgrad_f16p_into(M: int, A16: torch.Tensor, lda: int, aexp: torch.Tensor, N: int, B: torch.Tensor, ldb: int,
                    bexp: Optional[torch.Tensor], K: int, dW_ptr: int, ldo: int, db_ptr: Optional[int], device, expo_a: int,
                    expo_b: int, frag: bool = False):
    """upnerf_wgrad for the f16 field mode's fp16-STORED operands: A16 [M][lda] fp16 scaled per 64-row tile by 2^aexp[tile]
    (gz16 / gzexp), B the same (fp16, bexp) or fp32 rows (bexp None: the encoding x0).  frag: the fp16 operands are the
    operand fragments of the register-resident field kernels (one exponent per 32 rows; include/upnerf_hip.h, tile_rows = 256)."""
    ns = nsplit_for(M)
    ws = workspace("wgrad", ns * (256 * 256 + 256), device)
    rc = TIMER.run(f"wgrad16p_{N}x{K}", lambda: lib.upnerf_wgrad_f16p(M, ptr(A16), lda, ptr(aexp), N, ptr(B), ldb, ptr(bexp),
                                                                     int(bexp is not None) | (2 if frag else 0), K, dW_ptr, ldo, db_ptr, ptr(ws), ns,
                                                                     expo_a, expo_b, stream()), units=M)
    check(rc, "upnerf_wgrad_f16p")


def vec_wgrad_into(M: int, v: torch.Tensor, ldv: int, nvec: int, X: torch.Tensor, ldx: int, K: int, dw_ptr: int,
                   dbv_ptr: Optional[int], device):
    ns = nsplit_for(M)
    ws = workspace("vec_wgrad", ns * 4 * (K + 1), device)
    check(lib.upnerf_vec_wgrad(M, ptr(v), ldv, nvec, ptr(X), ldx, K, dw_ptr, dbv_ptr, ptr(ws), ns, stream()),
          "upnerf_vec_wgrad")


def vec_wgrad_frag16_into(M: int, v: torch.Tensor, ldv: int, nvec: int, X16: torch.Tensor, xexp: torch.Tensor, K: int, dw_ptr: int,
                          dbv_ptr: Optional[int], device):
    """vec_wgrad_into against a 256- or 128-wide tensor of the register-resident kernels' fp16 operand fragments."""
    ns = nsplit_for(M)
    ws = workspace("vec_wgrad", ns * 4 * 257, device)
    check(lib.upnerf_vec_wgrad_frag16(M, ptr(v), ldv, nvec, ptr(X16), ptr(xexp), K, dw_ptr, dbv_ptr, ptr(ws), ns, stream()),
          "upnerf_vec_wgrad_frag16")


def linear_raw(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int, w_is_kn: bool = False) -> torch.Tensor:
    """y = act(x w^T + b) with x [M][K], w [N][K]; K is padded to a multiple of 8 when needed.
    w_is_kn: w is given as [K][N] (y = act(x w + b)): the kernel reads it in place, no transposed copy."""
    M, K = x.shape
    if w_is_kn:
        if K % 8 == 0 and w.is_contiguous():
            x = x.contiguous()
            N = w.shape[1]
            b = b.contiguous() if b is not None else None
            y = torch.empty(M, N, device=x.device, dtype=torch.float32)
            check(lib.upnerf_linear(M, N, K, ptr(x), K, ptr(w), N, ptr(b), ptr(y), N, act | 2, stream()), "upnerf_linear")
            return y
        w = w.t()
    N = w.shape[0]
    if K % 8:
        pad = 8 - K % 8
        x, w = F.pad(x, (0, pad)), F.pad(w, (0, pad))
        K += pad
    x, w = x.contiguous(), w.contiguous()
    b = b.contiguous() if b is not None else None  # (named: must outlive the launch call)
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    check(lib.upnerf_linear(M, N, K, ptr(x), K, ptr(w), K, ptr(b), ptr(y), N, act, stream()), "upnerf_linear")
    return y


def linear_kn_view(x: torch.Tensor, w_base: torch.Tensor, w_off: int, ldw: int, N: int) -> torch.Tensor:
    """y[M][N] = x[M][K] . Wv[K][N] where Wv is the strided view  w_base[w_off + k*ldw + n]  (a column block of a packed
    row-major matrix), read in place by the kernel."""
    M, K = x.shape
    if K % 8:
        raise ValueError("linear_kn_view needs K to be a multiple of 8")
    x = x.contiguous()
    y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    check(lib.upnerf_linear(M, N, K, ptr(x), K, w_base.data_ptr() + 4 * w_off, ldw, None, ptr(y), N, 2, stream()),
          "upnerf_linear")
    return y


class HipLinear(torch.autograd.Function):
    """y = act(x W^T + b) on the fp32 MFMA kernel (nn.Linear [+ ReLU] of models/transient_net.py:11-25 and the
    per-ray feature projection)."""

    @staticmethod
    def forward(ctx, x, w, b, act: int, defer_wgrad: bool = False):
        y = linear_raw(x.detach(), w.detach(), None if b is None else b.detach(), act)
        ctx.save_for_backward(x, w, y if act == 1 else None)
        ctx.has_bias, ctx.act, ctx.defer = b is not None, act, defer_wgrad
        ctx.owners = (w, b) if defer_wgrad else None  # the leaves themselves: the deferral rule is checked on them
        ctx.tokens = _defer_token(w, b) if defer_wgrad else None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        gy = gy.contiguous()
        if ctx.act == 1:
            gy = torch.ops.aten.threshold_backward(gy, y, 0.0)  # ReLU backward in one launch
        M, K = x.shape
        N = w.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = linear_raw(gy, w, None, 0, w_is_kn=True)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            xc = x.contiguous()
            gw = torch.empty(N, K, device=x.device, dtype=torch.float32)
            gb = torch.empty(N, device=x.device, dtype=torch.float32)
            if N <= 3:
                vec_wgrad_into(M, gy, N, N, xc, K, K, gw.data_ptr(), gb.data_ptr(), x.device) \
                    if K in (32, 64, 128, 256) else _vec_wgrad_wide(M, gy, N, xc, K, gw, gb)
            else:
                Kp, Np = (K + 3) // 4 * 4, (N + 3) // 4 * 4
                if Kp != K or Np != N:
                    raise ValueError("HipLinear needs in/out features that are multiples of 4 (or out <= 4)")
                if (ctx.defer and DEFERRED_WGRADS.enabled and M <= DEFERRED_WGRADS.MAX_M and not torch.is_grad_enabled()
                        and can_adopt(ctx.owners[0]) and (not ctx.has_bias or can_adopt(ctx.owners[1]))
                        and all(t[0] == 1 for t in ctx.tokens)):
                    DEFERRED_WGRADS.add(M, gy, N, N, xc, K, K, gw, gb if ctx.has_bias else None,
                                        owners=((ctx.owners[0], gw), (ctx.owners[1], gb if ctx.has_bias else None)))
                    return gx, gw, gb if ctx.has_bias else None, None, None
                if (N > 256 or K > 256) and M <= DEFERRED_WGRADS.MAX_M:
                    wgrad_blocks_into(M, gy, N, N, xc, K, K, gw.data_ptr(), K, gb.data_ptr(), x.device)
                else:
                    for n0 in range(0, N, 256):
                        nn_ = min(256, N - n0)
                        for k0 in range(0, K, 256):
                            kk = min(256, K - k0)
                            wgrad_into(M, gy, N, nn_, xc, K, kk, gw.data_ptr() + 4 * (n0 * K + k0), K,
                                       gb.data_ptr() + 4 * n0 if k0 == 0 else None, x.device, a_off=n0, b_off=k0)
            if not ctx.has_bias:
                gb = None
        return gx, gw, gb, None, None


def _vec_wgrad_wide(M, gy, N, xc, K, gw, gb):
    for k0 in range(0, K, 256):
        kk = min(256, K - k0)
        part = torch.empty(N, kk, device=xc.device, dtype=torch.float32)
        xs = xc[:, k0:k0 + kk].contiguous()
        vec_wgrad_into(M, gy, N, N, xs, kk, kk, part.data_ptr(), gb.data_ptr(), xc.device)
        gw[:, k0:k0 + kk] = part


def hip_linear(x, w, b=None, relu: bool = False, defer_wgrad: bool = False):
    """defer_wgrad: the weight gradient may be computed at the end of the backward pass, grouped with the other small
    ones -- only for a weight (and bias) that nothing else in the graph consumes (see _DeferredWgrads)."""
    return HipLinear.apply(x, w, b, 1 if relu else 0, defer_wgrad)


class _DeferredEmbeds:
    """Table gradients of `embed_rows(..., defer_grad=True)`, collected during a backward pass and computed at its end
    (same mechanism and same single-consumer rule as _DeferredWgrads): tables gathered with the same index tensor share
    ONE launch and one scan of the indices (seven per-image tables per training step)."""

    def __init__(self):
        self.items = []

    def add(self, idx, N, dim, g, out, owner=None):
        torch.autograd.Variable._execution_engine.queue_callback(self.flush)
        # storage only: see _DeferredWgrads
        self.items.append((idx, N, dim, g, out.data_ptr(), out.untyped_storage(), owner))

    def flush(self):
        items, self.items = self.items, []
        if _DeferredWgrads.check_adopted:
            for it in items:
                if it[6] is not None and (it[6].grad is None or it[6].grad.data_ptr() != it[4]):
                    raise RuntimeError("deferred table gradient was not adopted by its parameter")
        buckets = {}
        for it in items:
            buckets.setdefault((it[0].data_ptr(), it[0].numel(), it[1]), []).append(it)
        for (_, R, N), its in buckets.items():
            for lo in range(0, len(its), _lib.MAX_EMBED_GROUPS):
                part = its[lo:lo + _lib.MAX_EMBED_GROUPS]
                arr = (_lib.EmbedGroup * len(part))(*[_lib.EmbedGroup(g=i[3].data_ptr(), out=i[4], dim=i[2]) for i in part])
                check(lib.upnerf_embed_bwd_grouped(R, N, ptr(part[0][0]), arr, len(part), stream()),
                      "upnerf_embed_bwd_grouped")


DEFERRED_EMBEDS = _DeferredEmbeds()


class _EmbedPrefetch:
    """Forward gathers of a training step's per-image tables in ONE launch (upnerf_embed_fwd_grouped) instead of an
    index_select launch per table: inside `with EMBED_PREFETCH.scope(modules):` the first embed_rows() of a registered table
    gathers the rows of every registered table of the same height for that index tensor; the later calls with the same index
    tensor are handed their slice.  The tables do not change inside the scope (the optimisers run after it).  Outside a
    scope, and for any table or index the scope does not know, embed_rows gathers on its own as before.

    Stream rule (r5 ADVICE): the arena belongs to the stream that filled it.  A request issued on ANOTHER stream (the
    TransientNet's side stream of hparams["hip.side_stream"]) is not served from it -- it gathers its own rows with an
    index_select on the consuming stream, as before the prefetch existed -- so no slice is ever read by a stream that has not
    been ordered behind the gather, and the allocator never hands the arena's block to one stream while another still reads it.

    Index range (r5 ADVICE): the grouped kernel writes NaN rows for an out-of-range index where index_select raised a device-side
    assert; the first prefetch of a process (and every prefetch with UPNERF_CHECK_EMBED_IDX=1; never with =0, never under graph
    capture) checks min / max of the index tensor against the table height on the host and raises IndexError."""

    def __init__(self):
        self.tables, self.cache = None, {}
        self.checked = 0

    @contextlib.contextmanager
    def scope(self, modules):
        if self.tables is not None or not ENABLE_EMBED_PREFETCH:  # (nested: the outer scope keeps the cache)
            yield
            return
        self.tables = [m.weight for m in modules if m is not None and getattr(m, "weight", None) is not None]
        self.cache = {}
        try:
            yield
        finally:
            self.tables, self.cache = None, {}

    def rows(self, table, idx):
        if self.tables is None:
            return None
        key = (idx.data_ptr(), idx.numel())
        hit = self.cache.get(key)
        cur = torch.cuda.current_stream(idx.device) if idx.is_cuda else None
        if hit is not None and hit["_stream"] != cur:
            return None  # filled on another stream: the caller gathers on its own stream
        if hit is None:
            ok = lambda w: (w.is_cuda and w.device == idx.device and w.dtype == torch.float32 and w.dim() == 2
                            and w.shape[1] <= 256 and w.shape[0] == table.shape[0] and w.is_contiguous())
            live, seen = [], set()
            for w in self.tables:
                if ok(w) and w.data_ptr() not in seen:
                    seen.add(w.data_ptr())
                    live.append(w)
            if table.data_ptr() not in seen:
                return None
            R = idx.numel()
            mode = os.environ.get("UPNERF_CHECK_EMBED_IDX")
            if R and mode != "0" and (mode == "1" or self.checked == 0) and not torch.cuda.is_current_stream_capturing():
                self.checked += 1
                lo_i, hi_i = int(idx.min()), int(idx.max())
                if lo_i < 0 or hi_i >= table.shape[0]:
                    raise IndexError(f"embedding index out of range: [{lo_i}, {hi_i}] for a table of {table.shape[0]} rows")
            arena = torch.empty(R * sum(w.shape[1] for w in live), device=idx.device, dtype=torch.float32)
            hit, off = {"_keep": (idx, arena), "_stream": cur}, 0  # (idx kept alive: its address is the cache key)
            for w in live:
                hit[w.data_ptr()] = (off, w.shape[1])
                off += R * w.shape[1]
            for lo in range(0, len(live), _lib.MAX_EMBED_GROUPS):
                part = live[lo:lo + _lib.MAX_EMBED_GROUPS]
                arr = (_lib.EmbedRowsGroup * len(part))(*[_lib.EmbedRowsGroup(table=w.data_ptr(), rows=arena.data_ptr() + 4 * hit[w.data_ptr()][0],
                                                                              dim=w.shape[1]) for w in part])
                check(lib.upnerf_embed_fwd_grouped(R, table.shape[0], ptr(idx), arr, len(part), stream()),
                      "upnerf_embed_fwd_grouped")
            self.cache[key] = hit
        at = hit.get(table.data_ptr())
        if at is None:
            return None
        # a fresh view per request: autograd attaches the calling node to the tensor object a forward returns
        return hit["_keep"][1][at[0]:at[0] + idx.numel() * at[1]].view(idx.numel(), at[1])


class _Fanout(torch.autograd.Function):
    """Two aliases of each input, for tensors that feed two consumers inside a step.  Autograd would add the two gradients of every
    such tensor with an ATen launch each; here the backward adds ALL pairs of the fan-out in one upnerf_add_pairs launch (a + b in
    fp32 either way: same bits).  A missing gradient passes the other one through."""

    @staticmethod
    def forward(ctx, *xs):
        ctx.set_materialize_grads(False)  # an alias nobody differentiated hands back None, not a zero tensor to be added
        return tuple(x.view_as(x) for x in xs) + tuple(x.view_as(x) for x in xs)

    @staticmethod
    def backward(ctx, *gs):
        n = len(gs) // 2
        outs, pairs = [None] * n, []
        for j in range(n):
            a, b = gs[j], gs[n + j]
            if a is None or b is None:
                outs[j] = a if b is None else b
                continue
            a, b = a.contiguous(), b.contiguous()
            if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape):
                outs[j] = a + b
                continue
            out = torch.empty_like(a)
            outs[j] = out
            pairs.append(_lib.AddPair(a=a.data_ptr(), b=b.data_ptr(), out=out.data_ptr(), n=a.numel()))
            ctx_keep = (a, b)  # noqa: F841  (alive until the launch below is enqueued; the caching allocator is stream-ordered)
        for lo in range(0, len(pairs), _lib.MAX_ADD_PAIRS):
            part = pairs[lo:lo + _lib.MAX_ADD_PAIRS]
            arr = (_lib.AddPair * len(part))(*part)
            check(lib.upnerf_add_pairs(arr, len(part), stream()), "upnerf_add_pairs")
        return tuple(outs)


def fanout(*xs):
    """(x_1a, ..., x_na), (x_1b, ..., x_nb): two aliases of every tensor that requires a gradient and lives on the GPU (others are
    handed through twice): consumer A takes the first tuple, consumer B the second."""
    live = [i for i, x in enumerate(xs) if isinstance(x, torch.Tensor) and x.is_cuda and x.requires_grad and ENABLE_FANOUT
            and torch.is_grad_enabled()]
    if not live:
        return tuple(xs), tuple(xs)
    res = _Fanout.apply(*[xs[i] for i in live])
    a, b = list(xs), list(xs)
    for k, i in enumerate(live):
        a[i], b[i] = res[k], res[len(live) + k]
    return tuple(a), tuple(b)


ENABLE_FANOUT = os.environ.get("UPNERF_FANOUT", "1") != "0"  # (0: autograd's own accumulation, for A/B runs)
ENABLE_EMBED_PREFETCH = os.environ.get("UPNERF_EMBED_PREFETCH", "1") != "0"  # (0: one gather launch per table, for A/B runs)
EMBED_PREFETCH = _EmbedPrefetch()


class _EmbedRows(torch.autograd.Function):
    """table[idx] whose backward is one HIP kernel (dense, deterministic) instead of ATen's sort-based embedding
    backward (14 small launches per table and step)."""

    @staticmethod
    def forward(ctx, table, idx, defer: bool = False):
        ctx.save_for_backward(idx)
        ctx.shape, ctx.defer = tuple(table.shape), defer
        ctx.owner = table if defer else None
        ctx.tokens = _defer_token(table) if defer else None
        rows = EMBED_PREFETCH.rows(table, idx)
        return rows if rows is not None else table.detach().index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        N, dim = ctx.shape
        g = g.contiguous()
        out = torch.empty(N, dim, device=g.device, dtype=torch.float32)
        if ctx.defer and DEFERRED_WGRADS.enabled and not torch.is_grad_enabled() and can_adopt(ctx.owner) \
                and all(t[0] == 1 for t in ctx.tokens):
            DEFERRED_EMBEDS.add(idx, N, dim, g, out, ctx.owner)
        else:
            check(lib.upnerf_embed_bwd(idx.numel(), N, dim, ptr(idx), ptr(g), ptr(out), stream()), "upnerf_embed_bwd")
        return out, None, None


def embed_rows(emb, idx: torch.Tensor, defer_grad: bool = False) -> torch.Tensor:
    """emb(idx) for an nn.Embedding with a float32 CUDA table of width <= 256 and a 1-D int64 index; anything else is
    handed to the module itself.  defer_grad: the table gradient may be computed at the end of the backward pass together
    with the other tables' -- only when nothing else in the graph consumes the table (see _DeferredEmbeds)."""
    w = getattr(emb, "weight", None)
    if (w is None or not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or w.shape[1] > 256 or idx.dim() != 1
            or idx.dtype != torch.int64 or getattr(emb, "padding_idx", None) is not None
            or getattr(emb, "max_norm", None) is not None):
        return emb(idx)
    return _EmbedRows.apply(w, idx.contiguous(), defer_grad)


def adam_flat_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """In-place Adam update of a flat fp32 buffer (utils/optim.py:20-33 -> torch.optim.Adam semantics)."""
    import math
    check(lib.upnerf_adam(p.numel(), ptr(p), ptr(g), ptr(m), ptr(v), beta1, beta2, eps, lr / (1 - beta1 ** step),
                          math.sqrt(1 - beta2 ** step), None, stream()), "upnerf_adam")
