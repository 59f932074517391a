"""Inference render rate: full validation images through NeRFSystem.validation_step -- chunked,
perturb = 0, no gradient -- at BASELINE.json configs[1]'s field shape (64 + 128 samples, two 8x256 fields).

    python tools/bench_render.py [--pixels 262144] [--chunk 16384] [--images 4] [--progress 0.8] [--field f16x3]

Prints one JSON line: rendered rays/s, ms per image, the chunk size and the peak HBM in use."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pixels", type=int, default=512 * 512)
    ap.add_argument("--chunk", type=int, default=16384)
    ap.add_argument("--images", type=int, default=4)
    ap.add_argument("--progress", type=float, default=0.8)
    ap.add_argument("--field", default="f16x3", choices=("f16x3", "f32"))
    ap.add_argument("--tto", type=int, default=-1, metavar="0|1",
                    help="render through NeRFSystemOptimize.validation_step (eval / TTO shape: schedule finished, no "
                         "candidate head) with coarse_sigma_only = this value")
    a = ap.parse_args()
    import bench
    from upnerf_amd import rendering, synth
    rendering.FIELD_MODE = a.field
    dev = torch.device("cuda", 0)
    sysm = bench.build_system(dev, a.progress)
    sysm.hparams["val.chunk_size"] = a.chunk
    b = synth.batch(a.pixels, 763, seed=7)
    b["img_idx"] = torch.full_like(b["img_idx"], 3)  # one image
    batch = {k: v.to(dev)[None] for k, v in b.items()}
    if a.tto >= 0:
        from upnerf_amd.nerf_system import SyntheticDataset
        from upnerf_amd.nerf_system_optimize import NeRFSystemOptimize
        tto = NeRFSystemOptimize(dict(sysm.hparams), SyntheticDataset(763), pose_optimize=True)
        tto.model_setup(trained_state=sysm.state_dict(), n_test_images=1)
        tto.coarse_sigma_only = bool(a.tto)
        sysm = tto.to(dev)
        b["img_idx"] = torch.zeros_like(b["img_idx"])
        batch = {k: v.to(dev) for k, v in b.items()}
    out = sysm.validation_step(batch)  # warm-up (and allocator growth)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(a.images):
        out = sysm.validation_step(batch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.images
    print(json.dumps({"metric": "rendered rays/sec (validation_step, no grad)", "value": a.pixels / dt,
                      "ms_per_image": dt * 1e3, "pixels": a.pixels, "chunk": a.chunk, "progress": a.progress,
                      "field": a.field, "tto_coarse_sigma_only": a.tto, "val_psnr": float(out["val_psnr"]),
                      "peak_hbm_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))


if __name__ == "__main__":
    main()
