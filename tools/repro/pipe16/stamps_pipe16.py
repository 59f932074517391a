"""Per-phase shader-clock breakdown of the pipelined forward trunk (128-sample tile; diagnostic build: make -C upnerf_amd/csrc stamps).

    UPNERF_LIB=upnerf_amd/libupnerf_hip_stamps.so python tools/stamps_pipe16.py
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
names = ["phase 1 (A | epilogue B)", "barrier 1", "phase 2 (A + B)", "barrier 2 + bound", "phase 3 (B | epilogue A)",
         "barrier 3 + bound", "layer 0 (lockstep)", "drain"]
tiles = N * (4096 * 64 + 4096 * 192) // 128
waves = tiles * 8 / 16
print("forward pipelined trunk, cycles per wave:")
for i, n in enumerate(names):
    per = buf[i] / waves / (7 if i < 6 else 1)
    print(f"  {n:32s} {per:9.0f}  per {'stage' if i < 6 else 'tile'}")
