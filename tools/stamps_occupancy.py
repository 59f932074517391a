import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UPNERF_LIB", os.path.join(ROOT, "upnerf_amd", "libupnerf_hip_stamps.so"))
import torch
import bench
from upnerf_amd import _lib
from upnerf_amd.rendering import render_rays
dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.3)
rd = _lib.lib.upnerf_stamps_read
rd.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
names = ["loop top", "K loop", "epilogue", "barrier 1", "planes write", "barrier 2", "tile store"]
import contextlib
for R, nograd in ((256, False), (512, False), (4096, False), (256, True), (512, True), (4096, True)):
    b = bench.make_batches(dev, 1, 100)[0]
    b = {k: v[:R].contiguous() for k, v in b.items()}
    rays = sysm.rays_from_batch(b).detach()
    for it in range(2):
        torch.cuda.synchronize(); rd(buf, 1)
        with (torch.no_grad() if nograd else contextlib.nullcontext()):
            res = render_rays(sysm.models, sysm.embeddings, rays, b["img_idx"], 0.5, N_samples=64, perturb=0, N_importance=0)
        torch.cuda.synchronize(); rd(buf, 1)
    tiles = R
    waves = tiles * 4 / 16
    print(f"R={R} ({tiles} workgroups{', no stores' if nograd else ''}):", "  ".join(f"{n} {buf[i]/waves/8:.0f}" for i, n in enumerate(names)), " sum", f"{sum(buf[:7])/waves/8:.0f}")
