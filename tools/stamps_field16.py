"""Per-phase shader-clock breakdown of the f16x3 forward trunk loop (diagnostic build: make -C upnerf_amd/csrc stamps).

    UPNERF_LIB=upnerf_amd/libupnerf_hip_stamps.so python tools/stamps_field16.py
    ... --heads   (build: make -C upnerf_amd/csrc stamps EXP=-DUPNERF_STAMPS_HEADS): eight pieces of the forward kernel's head stage
    ... --bheads  (build: ... EXP=-DUPNERF_STAMPS_BHEADS): seven pieces of the backward kernel's head stage
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
names = ["loop top", "K loop (issue)", "epilogue: fma/relu/pack/max", "barrier 1", "planes write (split16 + ds_write)",
         "barrier 2", "tile store (planes -> fp32 global)"]
tiles = N * (4096 * 64 + 4096 * 192) // 64
waves = tiles * 4 / 16  # one workgroup in 16 reports
tot = sum(buf[:7])
print(f"{tiles} tiles x 8 layers, 4 waves each; cycles per wave per layer:")
for i, n in enumerate(names):
    print(f"  {n:40s} {buf[i] / waves / 8:9.0f}   {100 * buf[i] / tot:5.1f} %")
print(f"  {'sum':40s} {tot / waves / 8:9.0f}")

if "--heads" in sys.argv:
    hn = ["colour: 256-deep K loop + store of e", "colour: side-input contraction + bias / relu / max", "candidate: 256-deep K loop",
          "candidate: side inputs + relu / sign bits / max", "barrier, exponent, two plane writes, barrier", "r1 store + three colour outputs",
          "g1 store, candidate_encoding.2, epilogue, planes", "g2 store + candidate density"]
    ht = sum(buf[8:16])
    print("forward kernel, head stage, cycles per wave per tile:")
    for i, n in enumerate(hn):
        print(f"  {n:52s} {buf[8 + i] / waves:9.0f}   {100 * buf[8 + i] / ht:5.1f} %")
    print(f"  {'sum':52s} {ht / waves:9.0f}")
    sys.exit(0)
if "--bheads" in sys.argv:
    hn = ["candidate: row loads (g2, g_G_c), d g2, store, column sums", "barrier, plane write, barrier",
          "candidate_encoding.2^T: 128-deep contraction, scale, mask, max", "colour: d r1 (rank 3), store, column / ray sums",
          "barrier, exponent, two plane writes, barrier", "gz_g1 store (+ its per-ray sums)", "partial sums: folds through LDS, three barriers"]
    ht = sum(buf[8:15])
    print("backward kernel, head stage, cycles per wave per tile:")
    for i, n in enumerate(hn):
        print(f"  {n:62s} {buf[8 + i] / waves:9.0f}   {100 * buf[8 + i] / ht:5.1f} %")
    print(f"  {'sum':62s} {ht / waves:9.0f}")
    sys.exit(0)
print(f"forward kernel outside the trunk, cycles per wave per tile: prologue + encoding {buf[7] / waves:.0f}, "
      f"density head + final layer {buf[13] / waves:.0f}, colour / candidate heads {buf[14] / waves:.0f}")
bnames = ["head stages (d g2, d r1, 128-wide contraction)", "d e (256-wide + rank-1 feature term)", "d h_{D-1}",
          "trunk, 7 layers", "d x0 -> d xyz"]
btot = sum(buf[8:13])
print("backward kernel, cycles per wave per tile:")
for i, n in enumerate(bnames):
    print(f"  {n:48s} {buf[8 + i] / waves:9.0f}   {100 * buf[8 + i] / btot:5.1f} %")
print(f"  {'sum':48s} {btot / waves:9.0f}")
