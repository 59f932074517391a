"""Fraction of exact zeros in the stored trunk tensors of one training step at the bench shape: h_l (post-ReLU
activations) and gz_l (gradients at the pre-activations), per layer and per field pass, plus the share of 128-byte lines
a zero-compacted row would touch.  (what a lossless compaction of the weight-gradient operands can save.)
    python tools/sparsity_probe.py [--progress 0.3] [--after N]   # N optimiser steps first (weights move away from init)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from upnerf_amd import rendering as rd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--progress", type=float, default=0.3)
    ap.add_argument("--after", type=int, default=0)
    ap.add_argument("--rays", type=int, default=4096)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sysm = bench.build_system(dev, a.progress, a.rays, bench.CONFIGS["brandenburg"]["n_images"])
    batches = bench.make_batches(dev, 4, 100, a.rays, bench.CONFIGS["brandenburg"]["n_images"])
    for i in range(a.after):
        sysm.training_step(batches[i % 4], i)
    seen = []
    real = rd._empty

    def spy(*shape, device):
        t = real(*shape, device=device)
        if len(shape) == 3 and shape[2] == 256:
            seen.append(t)
        return t

    rd._empty = spy
    sysm.training_step(batches[0], a.after)
    torch.cuda.synchronize()
    rd._empty = real
    for t in seen:
        D, M, W = t.shape
        nz = (t != 0).sum(-1)  # [D][M] non-zeros per row
        frac0 = 1.0 - nz.double().mean(-1) / W
        lines = torch.ceil(nz.double() * 4 / 128).mean(-1) / 8
        print(f"tensor {tuple(t.shape)}: zeros per layer", [round(float(x), 3) for x in frac0],
              "| lines touched by compact rows", [round(float(x), 3) for x in lines])


if __name__ == "__main__":
    main()
