"""Per-phase shader-clock breakdown of the slab loop of the register-resident fp16 kernels (csrc/field16rr.hip; diagnostic
build: make -C upnerf_amd/csrc stamps).

    UPNERF_LIB=upnerf_amd/libupnerf_hip_stamps.so python tools/stamps_rr16.py      (GPU box, repo root)
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UPNERF_LIB", os.path.join(ROOT, "upnerf_amd", "libupnerf_hip_stamps.so"))
import torch
import bench
from upnerf_amd import _lib, rendering

rendering.FIELD_MODE, rendering.FIELD_RR = "f16", 1
dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.3, rays=8192, n_images=1689)
batches = bench.make_batches(dev, 2, 100, rays=8192, n_images=1689)
for i in range(3):
    sysm.training_step(batches[i % 2], i)
rd = _lib.lib.upnerf_stamps_read_rr
rd.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 24)()
rd(buf, 1)
N = 5
for i in range(N):
    sysm.training_step(batches[i % 2], i)
rd(buf, 1)
names = ["wait for the slab's DMA (vmcnt, in order: older stores too)", "barrier", "DMA issue", "MFMAs + epilogues + stores"]
for base, tag in ((0, "forward"), (8, "backward")):
    waves = max(1, buf[base + 5])
    tot = sum(buf[base:base + 4])
    print(f"{tag}: {waves} waves sampled, slab loop {buf[base + 4] / waves:.0f} cycles per wave")
    for i, n in enumerate(names):
        print(f"  {n:62s} {buf[base + i] / waves:10.0f}   {100 * buf[base + i] / max(1, tot):5.1f} %")
    if base == 0:
        w = max(1, buf[5])
        print(f"  forward trunk epilogue detail per slab (64 slabs): before {buf[16] / w / 64:.0f}, arithmetic {buf[17] / w / 64:.0f}, stores {buf[18] / w / 64:.0f}")
        print(f"  forward trunk (64 of the 84 slabs), per wave: contraction {buf[6] / waves:.0f}, epilogues + stores {buf[7] / waves:.0f} "
              f"(both inside the last row; per slab {buf[6] / waves / 64:.0f} + {buf[7] / waves / 64:.0f})")
