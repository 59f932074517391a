"""Does a weight-gradient kernel (HBM-bound) overlap usefully with a field kernel (issue / L2-bound) when the two run on
different streams?  Times 8 x wgrad_f16x3 (786 432 x 256 x 256) and two coarse + fine field forwards (no stores) back to back and concurrently.  python tools/overlap_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from upnerf_amd import ops
from upnerf_amd.rendering import render_rays

dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.8)
b = bench.make_batches(dev, 1, 100)[0]
rays = sysm.rays_from_batch(b).detach()
M = 4096 * 192
A = torch.randn(M, 256, device=dev)
B = torch.randn(M, 256, device=dev)
dW = torch.empty(8, 256, 256, device=dev)
db = torch.empty(8, 256, device=dev)
expo = ops.scale_exponents(A, B)
side = torch.cuda.Stream()


def wgrads():
    for i in range(8):
        ops.wgrad_f16x3_into(M, A, 256, 256, B, 256, 256, dW[i].data_ptr(), 256, db[i].data_ptr(), dev, expo=expo)


def field():
    with torch.no_grad():  # the inference variant (no activation stores): the purest issue / L2-bound partner
        for _ in range(2):
            render_rays(sysm.models, sysm.embeddings, rays, b["img_idx"], 1.0, N_samples=64, perturb=0, N_importance=128)


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        wgrads()
    field()
    torch.cuda.current_stream().wait_stream(side)


tw, tf, tb = timed(wgrads), timed(field), timed(both)
print(f"8 weight gradients {tw:.2f} ms, 2 x (coarse + fine) forward {tf:.2f} ms, serial {tw + tf:.2f} ms, concurrent {tb:.2f} ms "
      f"({100 * (1 - tb / (tw + tf)):.1f} % saved)")
