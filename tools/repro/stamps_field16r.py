"""Per-phase shader-clock breakdown of the register-resident forward kernel (diagnostic build: make -C upnerf_amd/csrc stamps).

    UPNERF_LIB=upnerf_amd/libupnerf_hip_stamps.so [UPNERF_FIELD_MODE=f16] python tools/stamps_field16r.py
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UPNERF_LIB", os.path.join(ROOT, "upnerf_amd", "libupnerf_hip_stamps.so"))
import torch
import bench
from upnerf_amd import _lib

dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.3)
batches = bench.make_batches(dev, 2, 100)
rd = _lib.lib.upnerf_stamps_read_r
rd.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
N = 4
with torch.no_grad():  # forward kernels only (the backward kernel of field16.hip shares the table)
    for i in range(2):
        sysm.compute_loss(batches[i % 2])
    rd(buf, 1)
    for i in range(N):
        sysm.compute_loss(batches[i % 2])
rd(buf, 1)
names = ["wait for the slab's DMA", "barrier", "DMA issue", "MFMA loop", "epilogue", "layer tail", "prologue", "heads outside slabs"]
wgs = N * (4096 * 64 + 4096 * 192) // 128
waves = wgs / 16  # lane 0 of every wave of one workgroup in 16 reports -> 4 waves per reporting workgroup
waves = wgs * 4 / 16
slabs = 8 * 9 + 4 + 4 + 4  # per workgroup in phase 1
tot = sum(buf[:8])
print(f"{wgs} workgroups x 4 waves; stamp units per wave per WORKGROUP (84 slabs):")
for i, n in enumerate(names):
    print(f"  {n:28s} {buf[i] / waves:10.0f}   {100 * buf[i] / tot:5.1f} %   per slab {buf[i] / waves / slabs:8.0f}")
print(f"  {'sum':28s} {tot / waves:10.0f}")
