"""Fine-grained stamps inside the steps of the pipelined forward trunk (make variant NAME=st_FINE EXP="-DUPNERF_STAMPS -DPL_EXP_FINE")."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("UPNERF_LIB", os.path.join(ROOT, "upnerf_amd", "libupnerf_hip_st_FINE.so"))
os.environ["UPNERF_FIELD_TILE"] = "128"
import torch
import bench
from upnerf_amd import _lib
dev = torch.device("cuda", 0)
sysm = bench.build_system(dev, 0.3)
batches = bench.make_batches(dev, 2, 100)
for i in range(3):
    sysm.training_step(batches[i % 2], i)
rd = _lib.lib.upnerf_stamps_read
rd.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 16)()
rd(buf, 1)
N = 5
for i in range(N):
    sysm.training_step(batches[i % 2], i)
rd(buf, 1)
names = ["P1: MFMAs of A (8 x 6)", "P1: x reads + epilogue chunk", "P2: MFMAs of B (8 x 6)", "P2: reads, request, MFMAs of A",
         "P3: MFMAs of B (8 x 6)", "P3: x reads + epilogue chunk", "before the MFMAs (late epilogue, requests, pieces)", "phase ends, barriers, rest"]
tiles = N * (4096 * 64 + 4096 * 192) // 128
waves = tiles * 8 / 16
for i, n in enumerate(names):
    print(f"  {n:52s} {buf[i] / waves / 7:9.0f}  per stage")
