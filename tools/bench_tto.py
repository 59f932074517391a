#!/usr/bin/env python3
"""Test-time optimisation step rate (BASELINE.json configs[4]: frozen fields, per-image pose + appearance Adam) at the
Brandenburg field shape: eager launches vs one graph replay per step.

    python tools/bench_tto.py [--rays 4096] [--steps 30] [--stage pose|appearance]"""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--stage", default="pose", choices=("pose", "appearance"))
    a = ap.parse_args()
    import bench
    from upnerf_amd import synth
    from upnerf_amd.graph_step import GraphedTrainingStep
    from upnerf_amd.nerf_system import SyntheticDataset
    from upnerf_amd.nerf_system_optimize import NeRFSystemOptimize
    dev = torch.device("cuda", 0)
    trained = bench.build_system(dev, 0.8)
    out = {"metric": "TTO training rays/sec", "rays": a.rays, "stage": a.stage}
    for mode in ("eager", "graph"):
        tto = NeRFSystemOptimize(dict(trained.hparams), SyntheticDataset(763), pose_optimize=a.stage == "pose")
        tto.model_setup(trained_state=trained.state_dict(), n_test_images=1)
        tto = tto.to(dev)
        batches = []
        for i in range(4):
            b = {k: v.to(dev) for k, v in synth.batch(a.rays, 1, seed=50 + i).items()}
            b["img_idx"] = torch.zeros_like(b["img_idx"])
            batches.append(b)
        step = GraphedTrainingStep(tto) if mode == "graph" else tto.training_step
        for i in range(4):
            step(batches[i % 4], i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(batches[i % 4], i)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[mode] = {"rays_per_s": a.rays / dt, "ms_per_step": dt * 1e3, "host_issue_ms_per_step": host / a.steps * 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
