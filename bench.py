#!/usr/bin/env python3
"""Benchmark of the UP-NeRF training hot path on MI355X (BASELINE.json: "training rays/sec").

One step = one full optimisation step of the Brandenburg-Gate configuration (BASELINE.json configs[1]) on a batch
of synthetic rays already resident in HBM:  se(3) refine -> rays -> render_rays coarse (64) + resample + fine (192)
on two 8x256 fields -> TransientNet -> UPNeRFLoss -> backward (data + weight gradients, pose gradients) ->
[gradient all-reduce when --gpus > 1] -> Adam + ExponentialLR on both optimisers.  fp32 end to end.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is whole-job rays/s (N ranks x 4096 rays per step, weak scaling: every rank
renders its own 4096-ray shard; model state is replicated and gradients are averaged by one flat all-reduce).

Extra objects on the line:
  roofline      dominant kernel of the step (largest summed device time) against the fp32 MFMA peak; `achieved` =
                algorithmic FLOPs per launch (SURVEY.md 8d per-sample figure x samples per launch) / average
                launch duration measured with HIP events on the launch stream over the timed steps.
  cpu_baseline  the CPU oracle (oracle/upnerf_oracle.py, a port of the reference's arithmetic) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
# f16 dense MFMA: 1024 FLOP/clk/SIMD (v_mfma_f32_32x32x16_f16 = 32 cycles) x 1024 SIMDs x 2.4 GHz (same guide, "~2.5 PF")
PEAK_F16_MFMA_TFLOPS = 2516.6
# the f16x3 kernels issue three f16 MFMAs (hi*hi + hi*lo + lo*hi) per fp32-accurate multiply-accumulate
PEAK_F16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3
N_IMAGES = 763                 # Brandenburg Gate train split
RAYS, NC, NF = 4096, 64, 128

# per-sample forward MACs of the reference network, D=8 W=256
MAC = {"trunk": 491008, "sigma": 256, "final": 65536, "feat": 98304, "cand": 100480, "rgb": 59136}


def algorithmic_fwd_mac(sched):
    m = MAC["trunk"] + MAC["sigma"] + MAC["final"] + MAC["feat"]
    if sched < 1:
        m += MAC["cand"]
    if sched > 0:
        m += MAC["rgb"]
    return m


def make_batches(dev, n, seed0):
    from upnerf_amd import synth
    out = []
    for i in range(n):
        b = synth.batch(RAYS, N_IMAGES, seed=seed0 + i)
        out.append({k: v.to(dev) for k, v in b.items()})
    return out


def build_system(dev, progress):
    from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams
    hp = default_hparams(**{"nerf.N_samples": NC, "nerf.N_importance": NF, "train.batch_size": RAYS})
    torch.manual_seed(0)
    sysm = NeRFSystem(hp, SyntheticDataset(N_IMAGES))
    sysm.setup()
    with torch.no_grad():  # small non-zero pose/depth tables so that every gradient path does real work
        sysm.se3_refine.weight.normal_(0, 1e-2)
        sysm.depth_scale.weight.normal_(0, 1e-2)
    sysm.to(dev)
    sysm.global_step = int(round(progress * 2 * hp["max_steps"]))
    sysm.set_progress(progress)
    return sysm


CPU_BASELINE_THREADS = 32  # fastest of 16 / 32 / 64 / 128 on the GPU box's 256-thread host (tools/cpu_baseline_threads.py:
#                            186 / 219 / 127 / 64 rays/s -- the oracle's ATen kernels stop scaling past one socket's worth)


def cpu_baseline(progress, rays=768, iters=3):
    """Oracle forward+backward (no optimiser) on a bounded sample of the same workload; returns rays/s."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import upnerf_oracle as orc
    from upnerf_amd import synth
    kw = dict(D=8, W=256, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
    st = {}
    for typ in ("coarse", "fine"):
        sd = synth.nerf_state(typ, seed=0, **kw)
        sd.pop("progress")
        st[f"nerf_{typ}"] = {k: v.requires_grad_(True) for k, v in sd.items()}
    st["transient_net"] = {k: v.requires_grad_(True) for k, v in synth.transient_state(N_IMAGES, seed=0).items()}
    for k, v in synth.tables(N_IMAGES, seed=0).items():
        st[k] = v.requires_grad_(True)
    cfgs = {f"nerf_{t}": orc.NerfCfg(typ=t, c2f=(0.1, 0.5), **kw) for t in ("coarse", "fine")}
    hp = {"pose.optimize": True, "nerf.near": 0.1, "nerf.far": 5.0, "candidate_schedule": (0.1, 0.5),
          "nerf.N_samples": NC, "nerf.N_importance": NF, "nerf.perturb": 1.0}
    b = synth.batch(rays, N_IMAGES, seed=5)
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        losses, _ = orc.training_forward(st, cfgs, b, hp, progress)
        sum(losses.values()).backward()
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / iters
    return rays / dt, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--progress", type=float, default=0.3,
                    help="training progress in [0,1]: 0.05 -> sched 0 (candidate only), 0.3 -> sched 0.5 (all heads, "
                         "the heaviest phase; default), 0.8 -> sched 1 (colour only)")
    ap.add_argument("--field", choices=["f16x3", "f32"], default="f16x3",
                    help="arithmetic of the field contractions: f16x3 = 3-term fp16 split on the f16 matrix cores "
                         "(fp32-level accuracy, default); f32 = fp32 MFMA kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-timing", choices=["field", "all"], default="field",
                    help="HIP-event timing inside the timed region: the two field kernels only (what the roofline object "
                         "needs; 4 event pairs per step) or every instrumented kernel class (≈45 pairs per step, costs "
                         "≈1.5 %% of the step)")
    args = ap.parse_args()

    from upnerf_amd import parallel
    rank, local, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from upnerf_amd.ops import TIMER

    from upnerf_amd import rendering
    rendering.FIELD_MODE = args.field
    sysm = build_system(dev, args.progress)
    if world > 1:
        sysm.enable_data_parallel()
    batches = make_batches(dev, 4, seed0=100 + 10 * rank)
    sched = sysm.get_schedule_mult(args.progress)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        sysm.training_step(batches[i % len(batches)], i)
    TIMER.reset()
    TIMER.enabled = not args.no_kernel_timing
    TIMER.only = {"field_fwd", "field_bwd"} if args.kernel_timing == "field" else None
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sysm.training_step(batches[i % len(batches)], i)
    barrier()
    dt = time.perf_counter() - t0
    TIMER.enabled = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = world * RAYS * args.steps / dt
    mac = algorithmic_fwd_mac(sched)
    line = {
        "metric": "training rays/sec", "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: Brandenburg Gate shape, 4096 rays/GPU/step, 64 coarse + 128 "
                               "fine samples, two 8x256 fields + candidate/colour heads + TransientNet + appearance/"
                               "candidate embeddings (763 images), pose optimisation ON, full step incl. both Adam updates",
                   "rays_per_gpu": RAYS, "N_samples": NC, "N_importance": NF, "progress": args.progress,
                   "sched_mult": sched, "parallelism": f"dp{world}",
                   "contraction": ("fp32 MFMA (v_mfma_f32_32x32x2_f32)" if args.field == "f32" else
                                   "fp32 operands and results; products formed as a 3-term fp16 hi/lo split on "
                                   "v_mfma_f32_32x32x16_f16 with fp32 accumulation (<= 1e-6 of the fp32-MFMA kernels on "
                                   "every activation, tests/test_hip_kernels.py, tests/test_hip_fullsize.py; "
                                   "--field f32 selects the fp32-MFMA kernels)")},
        "algorithmic_tflop_per_step": 3 * 2 * mac * RAYS * (NC + NC + NF) / 1e12,
    }
    if not args.no_kernel_timing:
        summ = TIMER.summary()
        # algorithmic FLOPs per sample of each kernel class: forward, data-gradient and weight-gradient passes each
        # contract every layer once ("training step = fwd + dgrad + wgrad ~ 3x fwd")
        per_sample = {"field_fwd": 2 * mac, "field_bwd": 2 * mac, "wgrad_256x256": 2 * 256 * 256,
                      "wgrad16_256x256": 2 * 256 * 256}
        kern = {}
        for name, s in summ.items():
            k = dict(launches_per_step=s["launches"] / args.steps, avg_ms=s["avg_ms"],
                     ms_per_step=s["total_ms"] / args.steps)
            if name in per_sample:
                k["tflops_algorithmic"] = per_sample[name] * s["units_per_launch"] / (s["avg_ms"] * 1e-3) / 1e12
            kern[name] = k
        line["kernels"] = kern
        dom = max((n for n in kern if n in per_sample), key=lambda n: kern[n]["ms_per_step"])
        ach = kern[dom]["tflops_algorithmic"]
        traffic = None  # HBM bytes per launch from the committed rocprofv3 PMC passes (cannot be collected live)
        pmc = os.path.join(ROOT, "profiles", "r01_d_pmc.json")  # tools/pmc_collect.sh on the same command
        pmc_name = {"field_fwd": "field16_fwd_kernel<64>" if args.field == "f16x3" else "field_fwd_kernel<256, 64>",
                    "field_bwd": "field16_bwd_kernel<64>" if args.field == "f16x3" else "field_bwd_kernel<256, 64>",
                    "wgrad16_256x256": "wgrad_f16x3_kernel<4, 4>", "wgrad_256x256": "wgrad_kernel<4, 4>"}.get(dom)
        if os.path.exists(pmc) and pmc_name:
            t = json.load(open(pmc)).get(pmc_name)
            if t and "fetch_bytes_per_launch" in t:
                traffic = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
        f16 = dom.startswith("wgrad16") or (dom.startswith("field") and args.field == "f16x3")
        peak = PEAK_F16X3_TFLOPS if f16 else PEAK_FP32_MFMA_TFLOPS
        line["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak,
                            "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                            "avg_launch_ms": kern[dom]["avg_ms"],
                            "note": "achieved = algorithmic (fp32-equivalent) FLOPs / HIP-event launch time, averaged "
                                    "over the coarse (262144-sample) and fine (786432-sample) launches; " +
                                    ("peak = f16 dense MFMA 2516.6 TF / 3 MFMAs per fp32-accurate MAC (the hardware "
                                     "executes 3x the algorithmic FLOPs); for scale: fp32 MFMA peak is 157.3 TF"
                                     if f16 else "peak = fp32 MFMA")}
    if world == 1 and not args.no_cpu_baseline:
        nthreads = min(CPU_BASELINE_THREADS, os.cpu_count() or 1)
        prev = torch.get_num_threads()
        torch.set_num_threads(nthreads)
        v, sec = cpu_baseline(args.progress)
        torch.set_num_threads(prev)
        line["cpu_baseline"] = {"value": v, "unit": "rays/s", "cores": nthreads, "kind": "port",
                                "sample": f"oracle forward+backward (no optimiser step) on 768 rays of the same "
                                          f"configuration (64+128 samples, 8x256 fields, pose opt ON), mean of 3 warm "
                                          f"iterations ({sec:.1f} s each) after 1 warm-up, torch threads = {nthreads} of "
                                          f"{os.cpu_count()} host threads (the fastest setting measured)"}
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
