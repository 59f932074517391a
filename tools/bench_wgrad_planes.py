"""The DMA-staged weight-gradient kernel on producer-split operands (upnerf_wgrad_planes_chain, round 6) against the shipped
f16x3 kernel on the fp32 rows those planes decode to: accuracy of both against fp64, bitwise / relative difference between the
two, launch time (HIP events, kernel + the chained reduction of the previous problem, as inside a training step).

    python tools/bench_wgrad_planes.py [M]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from upnerf_amd import _lib
from upnerf_amd._lib import check, lib, ptr, stream
from upnerf_amd.ops import nsplit_for, workspace


def split_planes(x, relu):
    """[M][256] fp32 -> hi, lo fp16 planes + one exponent per 64 rows (what the field kernels hold in LDS), and the fp32 rows they
    decode to."""
    M = x.shape[0]
    t = x.view(M // 64, 64 * 256)
    mx = t.abs().amax(1).clamp_min(1e-30)
    ex = torch.frexp(mx)[1]
    e = (14 - ex).clamp(-100, 100).to(torch.int32)
    sc = torch.ldexp(torch.ones_like(mx), e)
    xs = (t * sc[:, None])
    hi = xs.to(torch.float16)
    lo = (xs - hi.float()).to(torch.float16)
    dec = ((hi.float() + lo.float()) / sc[:, None]).view(M, 256).contiguous()
    return hi.view(M, 256).contiguous(), lo.view(M, 256).contiguous(), e.contiguous(), dec


def tensor_exp(x):
    mx = float(x.abs().max())
    import math
    return 14 - math.frexp(mx)[1] if mx > 0 else 0


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 786432
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(3)
    A = (torch.randn(M, 256, generator=g) * torch.rand(M, 1, generator=g) ** 3).to(dev)          # gradients: signed, rows of very different size
    A = A * (torch.rand(M, 256, generator=g).to(dev) > 0.5)                                       # ReLU-masked
    B = torch.relu(torch.randn(M, 256, generator=g)).to(dev)                                      # activations
    Ah, Al, aexp, Ad = split_planes(A, False)
    Bh, Bl, bexp, Bd = split_planes(B, True)
    ea = torch.tensor([tensor_exp(Ad)], dtype=torch.int32, device=dev)
    eb = torch.tensor([tensor_exp(Bd)], dtype=torch.int32, device=dev)
    ns = nsplit_for(M)
    ws = [workspace(f"bwp{i}", ns * (256 * 256 + 256 + 260), dev) for i in range(2)]
    out = {}
    for name in ("f16x3 on fp32 rows", "planes + LDS-DMA"):
        dW, db = torch.zeros(256, 256, device=dev), torch.zeros(256, device=dev)
        pend = _lib.WgradPending()

        def launch(i, dW=dW, db=db, pend=pend, name=name):
            if name.startswith("f16x3"):
                return lib.upnerf_wgrad_f16x3_chain(M, ptr(Ad), 256, 256, ptr(Bd), 256, 256, ptr(dW), 256, ptr(db), ptr(ws[i & 1]), ns, ptr(ea), ptr(eb),
                                                   2, C.byref(pend), stream())
            return lib.upnerf_wgrad_planes_chain(M, ptr(Ah), ptr(Al), ptr(aexp), ptr(Bh), ptr(Bl), ptr(bexp), ptr(dW), 256, ptr(db), ptr(ws[i & 1]), ns,
                                                 ptr(ea), ptr(eb), C.byref(pend), stream())

        check(launch(0), name)
        check(lib.upnerf_wgrad_finish(C.byref(pend), stream()), "finish")
        torch.cuda.synchronize()
        out[name] = (dW.clone(), db.clone())
        for _ in range(3):
            launch(_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for i in range(n):
            launch(i)
        e1.record()
        check(lib.upnerf_wgrad_finish(C.byref(pend), stream()), "finish")
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"{name:22s}: {ms * 1e3:7.1f} us per launch (incl. the previous problem's slab reduction), {2 * M * 1024 / ms / 1e9:.2f} TB/s of operand bytes")
    ref = Ad.double().t() @ Bd.double()
    refb = Ad.double().sum(0)
    for name, (dW, db) in out.items():
        print(f"{name:22s}: dW vs fp64 {float((dW.double() - ref).abs().max() / ref.abs().max()):.2e}, db vs fp64 {float((db.double() - refb).abs().max() / refb.abs().max()):.2e}")
    a, b = out["f16x3 on fp32 rows"], out["planes + LDS-DMA"]
    print(f"planes vs f16x3: dW max |diff| / max {float((a[0] - b[0]).abs().max() / a[0].abs().max()):.2e} (bitwise equal: {torch.equal(a[0], b[0])}), "
          f"db {float((a[1] - b[1]).abs().max() / a[1].abs().max()):.2e}")


if __name__ == "__main__":
    main()
