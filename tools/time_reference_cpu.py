#!/usr/bin/env python3
"""Time the REAL reference (imported from /root/reference) on this container's CPU cores: the reproducible form of the
survey's probes.  Build container only -- the reference does not travel to the GPU box.

    python tools/time_reference_cpu.py [--iters 3] [--out profiles/ref_cpu_container.json]

Workloads (forward + UPNeRFLoss + backward + both Adam updates, the unit bench.py counts rays for):
  cfg1       BASELINE.json configs[0]: 4096 rays, 64 coarse samples, 4x64 field, poses frozen, sched 0
  cfg2_r512  BASELINE.json configs[1] shape at 512 rays (full 4096 needs ~22 GB and minutes per step): 64 + 128 samples,
             8x256 fields, pose optimisation ON, the three schedule phases (progress 0.05 / 0.3 / 0.8)
Each: 1 warm-up iteration, then `iters` timed ones; mean, min and max are recorded together with the host description.
The glue of NeRFSystem.training_step (models/nerf_system.py:150-200, needs pytorch_lightning) is the restatement of
tools/make_goldens.py (same leaf functions of the reference)."""
import argparse
import json
import os
import platform
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402  (registers the import shims and imports the reference's modules)

from upnerf_amd import synth  # noqa: E402


def step_fn(case):
    models, tn, emb = mg.build(case)
    b = synth.batch(case["R"], case["n_img"], seed=case["seed"] + 1)
    idx = b["img_idx"]
    fine = case["Nf"] > 0
    field_params = [p for m in models.values() for p in m.parameters()] + list(tn.parameters()) + \
        [e.weight for k, e in emb.items() if k.startswith("embedding_")]
    opts = [torch.optim.Adam(field_params, lr=5e-4, eps=1e-8),  # utils/optim.py:20-33; nerf_system.py:41-73
            torch.optim.Adam([emb["depth_scale"].weight, emb["se3_refine"].weight], lr=2e-3, eps=1e-8)]
    loss_fn = mg.ref_losses.UPNeRFLoss(depth_mult=1e-3, alpha_reg=1.0, encode_feat=True, fine=fine)
    m = mg.schedule_mult(case["progress"], (0.1, 0.5))
    embeddings = {k[len("embedding_"):]: v for k, v in emb.items() if k.startswith("embedding_")}

    def step():
        for o in opts:
            o.zero_grad()
        if case["pose_opt"]:
            pose = mg.ref_camera.pose.compose([mg.ref_camera.lie.se3_to_SE3(emb["se3_refine"](idx)), b["c2w"]])
        else:
            pose = b["c2w"]
        rays_o, rays_d = mg.ref_ray.get_rays(b["directions"], pose)
        rays = torch.cat([rays_o, rays_d, b["ray_infos"]], 1)
        scale, shift = torch.unbind(emb["depth_scale"](idx), 1)
        pred_inv = b["inv_depths"] * torch.exp(scale) + shift
        pred_inv[pred_inv < 1 / 5.0] = 1 / 5.0
        depth = 1.0 / pred_inv
        depth[depth < 0.1] = 0.1
        res = mg.ref_rendering.render_rays(models=models, embeddings=embeddings, rays=rays, img_idx=idx, sched_mult=m,
                                           sched_phase=0, N_samples=case["Nc"], use_disp=False, perturb=1.0,
                                           N_importance=case["Nf"], white_back=False, encode_feat=True, validation=False)
        if m > 0:
            t = tn(b["feats"], idx)
            res["rgb_coarse"] = res["s_rgb_coarse"] * (1 - t["alpha"].detach()) + t["rgb"].detach() * t["alpha"].detach()
            if fine:
                res["rgb_fine"] = res["s_rgb_fine"] * (1 - t["alpha"]) + t["rgb"] * t["alpha"]
            res["t_beta"], res["t_alpha"] = t["beta"], t["alpha"]
        loss = sum(l for l in loss_fn(res, b["rgbs"], b["feats"], depth, m).values())
        loss.backward()
        for o in opts:
            o.step()
        return float(loss)

    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "profiles", "ref_cpu_container.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    base = dict(n_img=763, seed=0, perturb=1.0)
    cases = {"cfg1": dict(base, R=4096, D=4, W=64, Nc=64, Nf=0, c2f=None, progress=0.0, pose_opt=False)}
    for p in (0.05, 0.3, 0.8):
        cases[f"cfg2_r512_progress{p}"] = dict(base, R=512, D=8, W=256, Nc=64, Nf=128, c2f=(0.1, 0.5), progress=p,
                                               pose_opt=True)
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), platform.processor())
    out = {"host": {"cpu": cpu, "logical_cpus": os.cpu_count(), "torch_threads": a.threads, "torch": torch.__version__},
           "what": "the reference's own modules imported from /root/reference; forward + loss + backward + both Adam updates",
           "cases": {}}
    for name, case in cases.items():
        step = step_fn(case)
        t0 = time.perf_counter()
        step()
        warm = time.perf_counter() - t0
        ts = []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        mean = sum(ts) / len(ts)
        out["cases"][name] = {"rays": case["R"], "iters": a.iters, "warmup_s": warm, "mean_s": mean, "min_s": min(ts),
                              "max_s": max(ts), "rays_per_s": case["R"] / mean,
                              "sched_mult": mg.schedule_mult(case["progress"], (0.1, 0.5))}
        print(name, json.dumps(out["cases"][name]), flush=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
