#!/usr/bin/env python3
"""A/B of the two weight-gradient kernels in one process: accuracy vs fp64 and time, at the sizes of one field layer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upnerf_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (M, N, K) in [(786432, 256, 256), (262144, 256, 256), (786432, 128, 256), (4096, 256, 256), (1000, 64, 80)]:
    a = (torch.randn(M, N, device=dev) * torch.rand(M, 1, device=dev) ** 4 * 1e-5)   # gradient-like: tiny, heavy tailed
    b = torch.relu(torch.randn(M, K, device=dev))                                     # activation-like
    ref = (a[:65536].double().t() @ b[:65536].double()) if M > 65536 else (a.double().t() @ b.double())
    out = {}
    for name in ("f32", "f16x3"):
        dW = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        fn = ops.wgrad_into if name == "f32" else ops.wgrad_f16x3_into
        expo = ops.scale_exponents(a, b) if name == "f16x3" else None
        kw = dict(expo=expo) if name == "f16x3" else {}
        for _ in range(2):
            fn(M, a, N, N, b, K, K, dW.data_ptr(), K, db.data_ptr(), dev, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            fn(M, a, N, N, b, K, K, dW.data_ptr(), K, db.data_ptr(), dev, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if M > 65536:
            dW2 = torch.empty(N, K, device=dev)
            kw2 = dict(expo=ops.scale_exponents(a[:65536], b[:65536])) if name == "f16x3" else {}
            fn(65536, a[:65536].contiguous(), N, N, b[:65536].contiguous(), K, K, dW2.data_ptr(), K, None, dev, **kw2)
            err = float((dW2.double() - ref).abs().max() / ref.abs().max())
        else:
            err = float((dW.double() - ref).abs().max() / ref.abs().max())
        dberr = float((db.double() - a.double().sum(0)).abs().max() / a.double().sum(0).abs().max())
        out[name] = (dt * 1e3, err, dberr)
        print(f"M={M} N={N} K={K} {name:6s}: {dt*1e3:8.3f} ms  {2*M*N*K/dt/1e12:7.1f} TFLOP/s  {M*(N+K)*4/dt/1e12:5.2f} TB/s  max-norm err {err:.2e}  bias err {dberr:.1e}")
