#!/usr/bin/env python3
"""Benchmark of the UP-NeRF training hot path on MI355X (BASELINE.json: "training rays/sec").

One step = one full optimisation step on a batch of synthetic rays already resident in HBM:  se(3) refine -> rays ->
render_rays coarse (64) + resample + fine (192) on two 8x256 fields -> TransientNet -> UPNeRFLoss -> backward (data +
weight gradients, pose gradients) -> [gradient all-reduce when --gpus > 1] -> Adam + ExponentialLR on both optimisers.

    python bench.py [--gpus 1] [--steps 20] [--warmup 5]
    python bench.py --gpus N           # no torchrun environment: spawns the N ranks itself (fresh processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is whole-job rays/s (N ranks x rays per step, weak scaling: every rank renders its
own shard; model state is replicated and gradients are averaged by one flat all-reduce over RCCL).

Workloads (--config):
  brandenburg   BASELINE.json configs[1]/[2]: 4096 rays per GPU, 763 images, fp32-accurate contractions (default)
  trevi         BASELINE.json configs[3]: 8192 rays per GPU, 1689 images, fp16 MLP weights/activations on MFMA (fp32 accumulate)

Extra objects on the line:
  roofline      dominant kernel of the step against the matrix peak of its arithmetic; `achieved` = algorithmic FLOPs per
                launch (SURVEY.md 8d per-sample figure x samples per launch) / average launch duration from HIP
                events on the launch stream.  The timed region replays captured HIP graphs (no place for event records
                between graph nodes), so the events bracket the same kernels in eager steps run right after it.
  phases        the three schedule phases BASELINE configs[1] asks for (progress 0.05 / 0.3 / 0.8), same K and W each
  strict_f32    the same step with every contraction on the fp32 MFMA (v_mfma_f32_32x32x2_f32) instead of the f16x3 split
  wgrad_f16     the same step with fp16-stored operands for the trunk weight gradients (an option; see its note)
  wgrad_f24     the same step with 24-bit stored operands (fp16 + residual byte) for the trunk weight gradients (an option)
  trevi         BASELINE.json configs[3] (8192 rays, 1689 images, fp16 field mode): value, ms_per_step, dtype, its own roofline
  tto           BASELINE.json configs[4]: pose stage and appearance stage at 1024 rays per step (graph replay), and the no-grad
                render rate of a full held-out image (4096-ray chunks, density-only coarse pass)
  cpu_baseline  the CPU oracle (oracle/upnerf_oracle.py, a port of the reference's arithmetic) timed on this box's host
                cores on a bounded sample of the same workload incl. the Adam update (rank 0, N == 1 only).
"""
import argparse
import gc
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
# f16 dense MFMA: 1024 FLOP/clk/SIMD (v_mfma_f32_32x32x16_f16 = 32 cycles) x 1024 SIMDs x 2.4 GHz (same guide, "~2.5 PF")
PEAK_F16_MFMA_TFLOPS = 2516.6
# the f16x3 kernels issue three f16 MFMAs (hi*hi + hi*lo + lo*hi) per fp32-accurate multiply-accumulate
PEAK = {"f32": PEAK_FP32_MFMA_TFLOPS, "f16x3": PEAK_F16_MFMA_TFLOPS / 3, "f16": PEAK_F16_MFMA_TFLOPS}
DTYPE = {"f32": "f32", "f16x3": "f32 (f16x3 emulated: 3 fp16 MFMAs per product, fp32 accumulate)",
         "f16": "f16 (fp32 accumulate)"}
NC, NF = 64, 128
CONFIGS = {
    "brandenburg": dict(rays=4096, n_images=763, field="f16x3",
                        workload="BASELINE.json configs[1]: Brandenburg Gate shape, 4096 rays/GPU/step, 64 coarse + 128 fine "
                                 "samples, two 8x256 fields + candidate/colour heads + TransientNet + appearance/candidate "
                                 "embeddings (763 images), pose optimisation ON, full step incl. both Adam updates"),
    "trevi": dict(rays=8192, n_images=1689, field="f16",
                  workload="BASELINE.json configs[3]: Trevi Fountain shape, 8192 rays/GPU/step, 1689 images, 64 coarse + 128 "
                           "fine samples, two 8x256 fields with fp16 weights and activations on MFMA (fp32 accumulate, fp32 "
                           "encoding / compositing / loss / Adam), pose optimisation ON, full step incl. both Adam updates"),
}
# per-sample forward MACs of the reference network, D=8 W=256
MAC = {"trunk": 491008, "sigma": 256, "final": 65536, "feat": 98304, "cand": 100480, "rgb": 59136}
CPU_BASELINE_THREADS = 32  # fastest of 16 / 32 / 64 / 128 on the GPU box's 256-thread host (tools/cpu_baseline_threads.py)


def algorithmic_fwd_mac(sched):
    m = MAC["trunk"] + MAC["sigma"] + MAC["final"] + MAC["feat"]
    if sched < 1:
        m += MAC["cand"]
    if sched > 0:
        m += MAC["rgb"]
    return m


def executed_mac(sched, direction="fwd", pose=True):
    """MACs per sample the field kernels actually ISSUE (csrc/field16.hip), as opposed to the reference network's
    (`algorithmic_fwd_mac`): composite-then-project removes the two 384-wide feature heads from the per-sample
    work (98 304 + 49 152 MAC run once per RAY instead), the colour head's first layer is folded onto e (256 + 80 padded side
    inputs instead of 459 inputs), and padded operand columns (63 -> 64 encoding columns) are counted because they are issued."""
    W, X0, AUXK, CK, W2 = 256, 64, 80, 16, 128
    if direction == "fwd":
        m = X0 * W + 3 * W * W + (X0 + W) * W + 3 * W * W  # trunk: layer 0, 1-3, skip layer, 5-7
        m += W + W * W                                     # density head, xyz_encoding_final
        if sched > 0:
            m += (W + AUXK) * W2 + 3 * W2                  # folded colour layer + its 3 outputs
        if sched < 1:
            m += (W + CK) * W2 + W2 * W2 + W2              # candidate layers + candidate density
        return m
    m = 7 * W * W + W * W                                  # d h_l for layers 7..1, d h_7 from d e
    m += ((W2 if sched > 0 else 0) + (W2 if sched < 1 else 0)) * W  # d e from [gz_r1 | gz_g1]
    if sched < 1:
        m += W2 * W2                                       # d g1 from d g2
    if pose:
        m += 2 * W * X0                                    # d x0: first layer + skip layer (-> d xyz)
    return m


HBM_COPY_TBS = 6.29  # /opt/skills/guides/MI355X_MICROARCH.md: measured float4 copy rate (8.0 TB/s spec)


def step_object(tflop_alg, tflop_exec, ms_per_step, field, step_bytes, comm_ms=None):
    """`roofline.step`: the whole step against its two floors -- algorithmic FLOPs at the dense peak of the field arithmetic,
    and the step's measured HBM bytes (PMC, all kernels; null without a profile of these sources) at the measured copy rate."""
    mfma_ms = tflop_alg / PEAK[field] * 1e3
    hbm_ms = None if step_bytes is None else step_bytes / (HBM_COPY_TBS * 1e12) * 1e3
    floor = max(mfma_ms, hbm_ms or 0.0)
    if comm_ms is not None:  # N > 1: the exchange is not the kernels' time -- the floors are compared with the step minus it
        compute_ms = max(ms_per_step - comm_ms, 1e-9)
        return {"algorithmic_tflop": tflop_alg, "algorithmic_tflops": tflop_alg / (compute_ms * 1e-3),
                "executed_tflop": tflop_exec, "mfma_floor_ms": mfma_ms, "hbm_bytes": step_bytes, "hbm_floor_ms": hbm_ms,
                "binding_floor": None if hbm_ms is None else ("hbm" if hbm_ms >= mfma_ms else "mfma"),
                "frac_of_floor": floor / compute_ms, "ms_per_step": ms_per_step, "comm_ms": comm_ms, "compute_ms": compute_ms,
                "note": "per rank; compute_ms = ms_per_step - comm.allreduce_ms (the serial exchange of a replayed step); floors as "
                        "at N = 1: algorithmic TFLOP / dense peak, PMC bytes / 6.29 TB/s; frac_of_floor = larger floor / compute_ms"}
    return {"algorithmic_tflop": tflop_alg, "algorithmic_tflops": tflop_alg / (ms_per_step * 1e-3),
            "executed_tflop": tflop_exec, "mfma_floor_ms": mfma_ms, "hbm_bytes": step_bytes, "hbm_floor_ms": hbm_ms,
            "binding_floor": None if hbm_ms is None else ("hbm" if hbm_ms >= mfma_ms else "mfma"),
            "frac_of_floor": floor / ms_per_step, "ms_per_step": ms_per_step,
            "note": "mfma_floor = algorithmic TFLOP / dense peak of the field arithmetic (2.4 GHz); hbm_floor = PMC bytes of "
                    "every kernel of a step / 6.29 TB/s (measured copy rate); frac_of_floor = the larger floor / the step"}


def comm_object(c, graph):
    """`comm` of an N > 1 line: what the gradient exchange cost per step, so that a scaling shortfall can be attributed without
    a second run.  allreduce_ms = time inside the all-reduce calls (max over ranks; includes waiting for the slowest rank to
    arrive), bytes = fp32 gradient bytes a rank contributes per step."""
    return {"allreduce_ms": c["allreduce_ms"], "bytes": c["bytes"], "allreduces_per_step": c["allreduces_per_step"],
            "early_launches": c["early_launches"], "late_only": c["late_only"], "clock": c["clock"],
            "placement": "between the two HIP graphs of a replayed step, in series (DESIGN.md section 7)" if graph
                         else "eager: fine-field bucket on a side stream under the rest of backward, the rest at its end"}


def source_sha16():
    """Hash of the HIP sources + C header: identifies the build a committed PMC profile belongs to."""
    h = hashlib.sha256()
    files = [os.path.join(ROOT, "include", "upnerf_hip.h")]
    d = os.path.join(ROOT, "upnerf_amd", "csrc")
    files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".cuh")))
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(n):
    """--gpus N without a torchrun environment: start the N ranks as fresh processes (this process has not touched the GPU
    and never will) and pass their output and exit code through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def make_batches(dev, n, seed0, rays=4096, n_images=763):
    from upnerf_amd import synth
    return [{k: v.to(dev) for k, v in synth.batch(rays, n_images, seed=seed0 + i).items()} for i in range(n)]


def build_system(dev, progress, rays=4096, n_images=763, nc=None, nf=None):
    import torch
    from upnerf_amd.nerf_system import NeRFSystem, SyntheticDataset, default_hparams
    hp = default_hparams(**{"nerf.N_samples": nc or NC, "nerf.N_importance": nf or NF, "train.batch_size": rays})
    torch.manual_seed(0)
    sysm = NeRFSystem(hp, SyntheticDataset(n_images))
    sysm.setup()
    with torch.no_grad():  # small non-zero pose/depth tables so that every gradient path does real work
        sysm.se3_refine.weight.normal_(0, 1e-2)
        sysm.depth_scale.weight.normal_(0, 1e-2)
    sysm.to(dev)
    sysm.global_step = int(round(progress * 2 * hp["max_steps"]))
    sysm.set_progress(progress)
    return sysm


def cpu_baseline(progress, n_images, rays=768, iters=3):
    """Oracle forward + backward + Adam update on a bounded sample of the same workload; returns rays/s."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import upnerf_oracle as orc
    from upnerf_amd import synth
    kw = dict(D=8, W=256, feat_dim=384, xyz_L=10, dir_L=4, appearance_dim=48, candidate_dim=16)
    st = {}
    for typ in ("coarse", "fine"):
        sd = synth.nerf_state(typ, seed=0, **kw)
        sd.pop("progress")
        st[f"nerf_{typ}"] = {k: v.requires_grad_(True) for k, v in sd.items()}
    st["transient_net"] = {k: v.requires_grad_(True) for k, v in synth.transient_state(n_images, seed=0).items()}
    for k, v in synth.tables(n_images, seed=0).items():
        st[k] = v.requires_grad_(True)
    leaves = [v for x in st.values() for v in (x.values() if isinstance(x, dict) else [x])]
    pose = [st["depth_scale"], st["se3_refine"]]
    opts = [torch.optim.Adam([v for v in leaves if all(v is not p for p in pose)], lr=5e-4, eps=1e-8),
            torch.optim.Adam(pose, lr=2e-3, eps=1e-8)]  # utils/optim.py:20-33, nerf_system.py:41-73
    cfgs = {f"nerf_{t}": orc.NerfCfg(typ=t, c2f=(0.1, 0.5), **kw) for t in ("coarse", "fine")}
    hp = {"pose.optimize": True, "nerf.near": 0.1, "nerf.far": 5.0, "candidate_schedule": (0.1, 0.5),
          "nerf.N_samples": NC, "nerf.N_importance": NF, "nerf.perturb": 1.0}
    b = synth.batch(rays, n_images, seed=5)
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        for o in opts:
            o.zero_grad()
        losses, _ = orc.training_forward(st, cfgs, b, hp, progress)
        sum(losses.values()).backward()
        for o in opts:
            o.step()
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / iters
    return rays / dt, dt


class Bench:
    def __init__(self, args, rank, world, dev, config=None):
        self.args, self.rank, self.world, self.dev = args, rank, world, dev
        self.cfg = CONFIGS[config or args.config]
        self.rays, self.n_images = self.cfg["rays"], self.cfg["n_images"]
        if getattr(args, "strong", False):
            if self.rays % world:
                raise SystemExit(f"--strong: {self.rays} rays do not split over {world} ranks")
            self.rays //= world  # per rank; `value` stays whole-job rays/s

    def barrier(self):
        import torch
        import torch.distributed as dist
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def leg(self, progress, field, graph, timer_only=None, steps=None, warmup=None, wgrad_store=None, repeats=1, nc=None, nf=None):
        """Build a fresh system, warm up, time `steps` steps (barrier + synchronize on both sides, max over ranks).  repeats > 1:
        the same bracketed measurement is taken again (repeats - 1) times after the one that is reported; `value_spread` = min /
        max over all of them (box-to-box and run-to-run noise of this bench is ~1 %: claims below that are inside it)."""
        import torch
        import torch.distributed as dist
        from upnerf_amd import _lib, rendering
        from upnerf_amd.graph_step import GraphedTrainingStep
        from upnerf_amd.ops import TIMER
        steps = self.args.steps if steps is None else steps
        warmup = self.args.warmup if warmup is None else warmup
        rendering.FIELD_MODE = field
        rendering.WGRAD_STORE = wgrad_store or "f32"
        sysm = build_system(self.dev, progress, self.rays, self.n_images, nc, nf)
        if self.world > 1:
            sysm.enable_data_parallel()
        batches = make_batches(self.dev, 4, 100 + 10 * self.rank, self.rays, self.n_images)
        step = GraphedTrainingStep(sysm) if graph else sysm.training_step
        for i in range(2 if graph else 0):  # first step of a shape signature runs eagerly, the second one is captured
            step(batches[i % 4], i)
        for i in range(warmup):
            step(batches[i % 4], i)
        TIMER.reset()
        TIMER.enabled, TIMER.only = timer_only is not None, timer_only or None
        calls0 = _lib.CALLS[0]
        sync = sysm.grad_sync if self.world > 1 else None
        if sync is not None:
            sync.comm_reset(timing=True)  # HIP events around every all-reduce of the timed steps (read back after the barrier)
        # no cyclic garbage collection inside the timed region: a generation-2 pass over the module / ctypes objects of a freshly
        # built system costs tens of ms of host time and lands wherever the allocation counters put it (seen inside a 4-step
        # region: 134 k instead of 240 k rays/s); collected here instead, re-enabled after the last repeat
        gc.collect()
        gc.disable()
        try:  # (r4 ADVICE: an exception inside the timed region must not leave the collector off for the process)
            self.barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                step(batches[i % 4], i)
            host = time.perf_counter() - t0
            self.barrier()
            dt = time.perf_counter() - t0
            TIMER.enabled = False
            comm = None
            if sync is not None:
                comm = sync.comm_summary(steps)
                sync.comm_reset(timing=False)
            if self.world > 1:
                t = torch.tensor([dt, host, comm["allreduce_ms"]], device=self.dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt, host, comm["allreduce_ms"] = float(t[0]), float(t[1]), float(t[2])
            out = {"value": self.world * self.rays * steps / dt, "ms_per_step": dt / steps * 1e3,
                   "host_issue_ms_per_step": host / steps * 1e3, "c_abi_calls_per_step": (_lib.CALLS[0] - calls0) / steps,
                   "sched_mult": sysm.get_schedule_mult(progress), "graph": bool(graph)}
            if comm is not None:
                out["comm"] = comm_object(comm, graph)
            if repeats > 1 and timer_only is None:
                vals = [out["value"]]
                for _ in range(repeats - 1):
                    self.barrier()
                    t0 = time.perf_counter()
                    for i in range(steps):
                        step(batches[i % 4], i)
                    self.barrier()
                    d2 = time.perf_counter() - t0
                    if self.world > 1:
                        t = torch.tensor([d2], device=self.dev, dtype=torch.float64)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        d2 = float(t[0])
                    vals.append(self.world * self.rays * steps / d2)
                out["value_spread"] = {"min": min(vals), "max": max(vals), "repeats": repeats, "steps_each": steps,
                                       "note": "the first of these measurements is `value`"}
        finally:
            TIMER.enabled = False
            gc.enable()
        if graph:
            out["graph_stats"] = dict(step.stats)
        summ = TIMER.summary() if timer_only is not None else None
        del step, sysm, batches
        gc.collect()
        torch.cuda.empty_cache()
        return out, summ


def roofline_of(kern, field, mac, note_extra="", sched=0.5):
    """`roofline` object of the dominant field kernel from a kernel-timing summary (HIP events on the launch stream)."""
    per_sample = {"field_fwd": 2 * mac, "field_bwd": 2 * mac}
    cand = [n for n in kern if n in per_sample]
    if not cand:
        return None  # (no field kernel in the timing summary)
    dom = max(cand, key=lambda n: kern[n]["ms_per_step"])
    ach = per_sample[dom] * kern[dom]["units_per_launch"] / (kern[dom]["avg_ms"] * 1e-3) / 1e12
    exe = 2 * executed_mac(sched, "fwd" if dom == "field_fwd" else "bwd") * kern[dom]["units_per_launch"] / (kern[dom]["avg_ms"] * 1e-3) / 1e12
    return {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK[field], "unit": "TFLOP/s", "frac": ach / PEAK[field],
            "executed_frac": exe / PEAK[field], "traffic": None, "avg_launch_ms": kern[dom]["avg_ms"], "note": note_extra}


def trevi_object(args, rank, world, dev):
    """BASELINE.json configs[3] beside the headline: 8192 rays, 1689 images, fp16 field mode; same step, same timing rules."""
    B = Bench(args, rank, world, dev, config="trevi")
    leg, _ = B.leg(0.3, "f16", not args.no_graph, repeats=3)
    _, summ = B.leg(0.3, "f16", False, timer_only={"field_fwd", "field_bwd"}, steps=min(args.steps, 6), warmup=2)
    nk = min(args.steps, 6)
    kern = {n: dict(launches_per_step=v["launches"] / nk, avg_ms=v["avg_ms"], ms_per_step=v["total_ms"] / nk,
                    units_per_launch=v["units_per_launch"]) for n, v in summ.items()}
    mac = algorithmic_fwd_mac(leg["sched_mult"])
    roof = roofline_of(kern, "f16", mac, "peak = f16 dense MFMA 2516.6 TF; HIP events over eager steps", sched=leg["sched_mult"])
    meta_p = os.path.join(ROOT, "profiles", "pmc_current_trevi.json")  # tools/pmc_current.py --out ...: same hash rule as the headline
    if roof and os.path.exists(meta_p):
        meta = json.load(open(meta_p))
        t = meta.get("kernels", {}).get(roof["kernel"])
        if meta.get("src_sha16") == source_sha16() and meta.get("field") == "f16" and meta.get("config") == "trevi" and t:
            roof["traffic"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
            roof["traffic_source"] = f"profiles/{meta.get('file')} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same sources)"
            roof["hbm_bytes_per_step"] = meta.get("hbm_bytes_per_step")
    if roof:
        spr = B.rays * (NC + NC + NF)  # field evaluations per step
        roof["step"] = step_object(3 * 2 * mac * spr / 1e12,
                                   2 * (2 * executed_mac(leg["sched_mult"], "fwd") + executed_mac(leg["sched_mult"], "bwd")) * spr / 1e12,
                                   leg["ms_per_step"], "f16", roof.get("hbm_bytes_per_step"))
    return {"value": leg["value"], "value_spread": leg.get("value_spread"), "unit": "rays/s", "ms_per_step": leg["ms_per_step"],
            "dtype": DTYPE["f16"],
            "config": {"workload": CONFIGS["trevi"]["workload"], "rays_per_gpu": B.rays, "n_images": B.n_images, "progress": 0.3,
                       "field": "f16", "kernels": "register-resident (csrc/field16rr.hip)"},
            "host_issue_ms_per_step": leg["host_issue_ms_per_step"],
            "roofline": roof}


def default_yaml_object(args, rank, world, dev):
    """The reference's SHIPPED sampling shape (configs/default.yaml:8-9, 50: N_samples 128, N_importance 128, batch 2048) at the
    Brandenburg field shape, f16x3 arithmetic: what `python train.py --config configs/brandenburg_gate.yaml` runs."""
    B = Bench(args, rank, world, dev, config="brandenburg")
    B.rays = 2048 if not getattr(args, "strong", False) else 2048 // world
    leg, _ = B.leg(0.3, "f16x3", not args.no_graph, nc=128, nf=128)
    return {"value": leg["value"], "unit": "rays/s", "ms_per_step": leg["ms_per_step"], "dtype": DTYPE["f16x3"],
            "config": {"workload": "reference configs/default.yaml sampling shape: 2048 rays/GPU/step, 128 coarse + 128 fine samples "
                                   "(fine pass S = 256), two 8x256 fields, 763 images, pose optimisation ON",
                       "rays_per_gpu": B.rays, "N_samples": 128, "N_importance": 128, "progress": 0.3, "field": "f16x3"}}


def tto_object(args, dev):
    """BASELINE.json configs[4]: test-time optimisation on frozen fields (per-image pose, then appearance; tto.py:100 batch of
    1024 rays; one graph replay per step) and the no-grad render of a full held-out image (175 k rays in 4096-ray chunks,
    density-only coarse pass)."""
    import torch
    from upnerf_amd import rendering, synth
    from upnerf_amd.graph_step import GraphedTrainingStep
    from upnerf_amd.nerf_system import SyntheticDataset
    from upnerf_amd.nerf_system_optimize import NeRFSystemOptimize
    rendering.FIELD_MODE, rendering.WGRAD_STORE = "f16x3", "f32"
    trained = build_system(dev, 0.8)
    out = {"field": "f16x3", "rays_per_step": 1024}
    steps, warm = max(args.steps, 20), 5
    for stage in ("pose", "appearance"):
        tto = NeRFSystemOptimize(dict(trained.hparams), SyntheticDataset(763), pose_optimize=stage == "pose")
        tto.model_setup(trained_state=trained.state_dict(), n_test_images=1)
        tto = tto.to(dev)
        batches = []
        for i in range(4):
            b = {k: v.to(dev) for k, v in synth.batch(1024, 1, seed=50 + i).items()}
            b["img_idx"] = torch.zeros_like(b["img_idx"])
            batches.append(b)
        step = GraphedTrainingStep(tto)
        for i in range(2 + warm):
            step(batches[i % 4], i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(batches[i % 4], i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[stage + "_stage"] = {"value": 1024 / dt, "unit": "rays/s", "ms_per_step": dt * 1e3, "steps": steps}
        if stage == "appearance":  # the evaluation render of the same system
            pixels = 175104  # ~500 x 350
            tto.hparams["val.chunk_size"] = 4096
            tto.coarse_sigma_only = True
            b = synth.batch(pixels, 1, seed=7)
            b["img_idx"] = torch.zeros_like(b["img_idx"])
            vb = {k: v.to(dev) for k, v in b.items()}
            tto.validation_step(vb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                tto.validation_step(vb)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            out["render"] = {"value": pixels / dt, "unit": "rays/s", "ms_per_image": dt * 1e3, "pixels": pixels, "chunk": 4096,
                             "no_grad": True, "coarse_sigma_only": True}
        del tto, step, batches
        gc.collect()
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="brandenburg")
    ap.add_argument("--progress", type=float, default=0.3,
                    help="training progress in [0,1]: 0.05 -> sched 0 (candidate only), 0.3 -> sched 0.5 (all heads, "
                         "the heaviest phase; default), 0.8 -> sched 1 (colour only)")
    ap.add_argument("--field", choices=["f16x3", "f32", "f16"], default=None,
                    help="arithmetic of the field contractions: f16x3 = 3-term fp16 split on the f16 matrix cores "
                         "(fp32-level accuracy; default of --config brandenburg); f32 = fp32 MFMA kernels; f16 = fp16 weights "
                         "and activations, one MFMA per product (default of --config trevi)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling (SURVEY.md 8d config 3): the configuration's rays per step are split over the ranks "
                         "(4096 rays total = 512 per rank at --gpus 8) instead of every rank rendering a full batch")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replay")
    ap.add_argument("--wgrad-store", choices=["f32", "f16", "f24"], default="f32",
                    help="diagnostic: storage of the trunk weight-gradient operands in the MAIN leg (f32 = the strict default; the "
                         "other two are the options the default line reports as wgrad_f16 / wgrad_f24); named in config.wgrad_store")
    ap.add_argument("--no-extras", action="store_true", help="skip the `phases` and `strict_f32` legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs34", action="store_true", help="skip the `trevi` (configs[3]) and `tto` (configs[4]) objects")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-timing", choices=["field", "all"], default="field",
                    help="HIP-event leg: the two field kernels only (what `roofline` needs) or every instrumented class")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise launcher, rendezvous, barriers and the output contract without touching a GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))  # before anything initialises a GPU in this process

    if os.environ.get("UPNERF_BENCH_KILL_RANK") is not None and os.environ.get("RANK") == os.environ["UPNERF_BENCH_KILL_RANK"]:
        sys.exit(3)  # test hook (tests/test_bench_contract.py): a rank that dies before the rendezvous fails the whole run
    import torch
    import torch.distributed as dist
    from upnerf_amd import parallel
    rank, local, world = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s)")
    observed = dist.get_world_size() if dist.is_initialized() else 1
    backend = dist.get_backend() if dist.is_initialized() else None
    cfg = CONFIGS[args.config]
    field = args.field or cfg["field"]

    if args.dry_run:  # plumbing only: same barriers / max-over-ranks / one line from rank 0
        t = torch.zeros(1)
        sync = None
        if world > 1:  # ... and the gradient exchange of the real step at its real size (2.25 M floats, fine-field bucket early)
            n_fine, n_rest = 823_000, 1_426_577
            ps = [torch.nn.Parameter(torch.zeros(n_fine)), torch.nn.Parameter(torch.zeros(n_rest))]
            sync = parallel.GradSync(ps, early=ps[:1])
            sync.comm_reset(timing=True)
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            t += 1
            if sync is not None:
                sync.begin(1)
                for q in ps:
                    q.grad = torch.full_like(q, float(rank))
                    if q is ps[0]:
                        sync._on_grad(q)  # (what the post-accumulate-grad hook does inside a backward pass)
                sync()
        if world > 1:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0, sync.comm_summary(args.steps)["allreduce_ms"] if sync else 0.0], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            assert float(ps[1].grad[0]) == (world - 1) / 2.0  # the mean over ranks arrived
        comm = None
        if sync is not None:
            c = sync.comm_summary(args.steps)
            c["allreduce_ms"] = float(dt[1])
            comm = comm_object(c, False)
        if rank == 0:
            print(json.dumps({"metric": "training rays/sec", "value": 0.0, "unit": "rays/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(dt[0]) / args.steps * 1e3,
                              **({"comm": comm, "roofline": {"step": step_object(0.0, 0.0, float(dt[0]) / args.steps * 1e3, field, None, comm["allreduce_ms"])}} if comm else {}),
                              "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
                              "dtype": DTYPE[field], "data": "synthetic", "dry_run": True, "world_size_observed": observed, "backend": backend,
                              "config": {"workload": cfg["workload"], "parallelism": f"dp{world}"}}))
        if world > 1:
            dist.destroy_process_group()
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = Bench(args, rank, world, dev)
    graph = not args.no_graph

    main_leg, _ = B.leg(args.progress, field, graph, repeats=3, wgrad_store=args.wgrad_store)
    extras = {}
    if not args.no_extras:
        phases = {}
        for p in (0.05, 0.3, 0.8):
            phases[str(p)] = main_leg if abs(p - args.progress) < 1e-9 else B.leg(p, field, graph)[0]
        extras["phases"] = {k: {x: v[x] for x in ("value", "ms_per_step", "sched_mult")} for k, v in phases.items()}
        if field == "f16x3":
            w16, _ = B.leg(args.progress, field, graph, wgrad_store="f16")
            extras["wgrad_f16"] = {"value": w16["value"], "ms_per_step": w16["ms_per_step"],
                                   "note": "same step with the operands of the trunk WEIGHT gradients (activations and their "
                                           "gradients) stored as fp16 tiles + per-tile exponents: half the HBM bytes of the "
                                           "field / weight-gradient kernels, one MFMA per product in dW = gz^T h.  Forward pass "
                                           "and data-gradient chain are unchanged (bitwise the outputs of `value`); a weight "
                                           "gradient carries ~3e-4 of unbiased rounding noise (golden gradient gates 1e-3 stay "
                                           "green, tests/test_hip_parity.py).  An option (rendering.WGRAD_STORE), not `value`."}
        if field == "f16x3":
            w24, _ = B.leg(args.progress, field, graph, wgrad_store="f24")
            extras["wgrad_f24"] = {"value": w24["value"], "ms_per_step": w24["ms_per_step"],
                                   "note": "same step with the operands of the trunk WEIGHT gradients stored as fp16 + a residual "
                                           "byte (hi + lo to 2^-20 of the tile's maximum: 3 bytes per element instead of 4), three "
                                           "MFMAs per product as in `value`.  Forward pass and data-gradient chain unchanged "
                                           "(bitwise); trunk weight gradients within 2e-5 of the fp32-stored ones "
                                           "(test_24bit_stored_weight_gradient_operands_option).  An option "
                                           "(rendering.WGRAD_STORE = \"f24\"), not `value`."}
        if field != "f32":
            f32, _ = B.leg(args.progress, "f32", graph)
            extras["strict_f32"] = {"value": f32["value"], "ms_per_step": f32["ms_per_step"], "dtype": "f32",
                                    "contraction": "fp32 MFMA (v_mfma_f32_32x32x2_f32) in every field contraction"}
    if not args.no_extras and args.config == "brandenburg" and not args.no_configs34:
        extras["trevi"] = trevi_object(args, rank, world, dev)  # BASELINE configs[3]
        extras["default_yaml"] = default_yaml_object(args, rank, world, dev)  # the reference's shipped sampling shape
        if world == 1:
            extras["tto"] = tto_object(args, dev)               # BASELINE configs[4]
    summ = None
    if not args.no_kernel_timing:
        only = {"field_fwd", "field_bwd"} if args.kernel_timing == "field" else set()
        _, summ = B.leg(args.progress, field, False, timer_only=only if only else set(),
                        steps=min(args.steps, 10), warmup=2, wgrad_store=args.wgrad_store)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    sched = main_leg["sched_mult"]
    mac = algorithmic_fwd_mac(sched)
    rays = B.rays
    line = {
        "metric": "training rays/sec", "value": main_leg["value"], "unit": "rays/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": main_leg["ms_per_step"], "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None, "dtype": DTYPE[field], "data": "synthetic",
        "config": {"workload": cfg["workload"], "rays_per_gpu": rays, "n_images": B.n_images, "N_samples": NC,
                   "N_importance": NF, "progress": args.progress, "sched_mult": sched, "parallelism": f"dp{world}",
                   "field": field, **({"wgrad_store": args.wgrad_store} if args.wgrad_store != "f32" else {}),
                   "launch": "HIP graph replay (one graph per shape signature, per-step scalars in device memory)"
                             if graph else "eager launches"},
        "world_size_observed": observed, "backend": backend,
        "host_issue_ms_per_step": main_leg["host_issue_ms_per_step"],
        "c_abi_calls_per_step": main_leg["c_abi_calls_per_step"],
        "value_spread": main_leg.get("value_spread"),
        "algorithmic_tflop_per_step": 3 * 2 * mac * rays * (NC + NC + NF) / 1e12,
    }
    if "graph_stats" in main_leg:
        line["graph_stats"] = main_leg["graph_stats"]
    if "comm" in main_leg:
        line["comm"] = main_leg["comm"]
    line.update(extras)
    if summ:
        per_sample = {"field_fwd": 2 * mac, "field_bwd": 2 * mac, "wgrad_256x256": 2 * 256 * 256,
                      "wgrad16_256x256": 2 * 256 * 256}
        nk = min(args.steps, 10)
        kern = {}
        for name, s in summ.items():
            k = dict(launches_per_step=s["launches"] / nk, avg_ms=s["avg_ms"], ms_per_step=s["total_ms"] / nk)
            if name in per_sample:
                k["tflops_algorithmic"] = per_sample[name] * s["units_per_launch"] / (s["avg_ms"] * 1e-3) / 1e12
            kern[name] = k
        line["kernels"] = kern
        dom = max((n for n in kern if n in per_sample), key=lambda n: kern[n]["ms_per_step"])
        ach = kern[dom]["tflops_algorithmic"]
        # HBM bytes per launch come from rocprofv3 PMC passes (tools/pmc_collect.sh; they cannot be collected live): used
        # only when the committed profile was taken from THIS build on THIS workload, otherwise null
        traffic, source, step_bytes = None, None, None
        meta_p = os.path.join(ROOT, "profiles", "pmc_current_trevi.json" if args.config == "trevi" else "pmc_current.json")
        if os.path.exists(meta_p):
            meta = json.load(open(meta_p))
            if (meta.get("src_sha16") == source_sha16() and meta.get("field") == field and meta.get("config") == args.config
                    and abs(meta.get("progress", -1) - args.progress) < 1e-9):
                t = meta.get("kernels", {}).get(dom)
                if t:
                    traffic = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
                    source = f"profiles/{meta.get('file')} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same sources)"
                    step_bytes = meta.get("hbm_bytes_per_step")
        peak = PEAK[field] if dom.startswith("field") else (PEAK["f16"] if field == "f16" else PEAK["f16x3"])
        exe = (2 * executed_mac(sched, "fwd" if dom == "field_fwd" else "bwd") * summ[dom]["units_per_launch"]
               / (kern[dom]["avg_ms"] * 1e-3) / 1e12) if dom.startswith("field") else ach
        spr = rays * (NC + NC + NF)
        line["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                            "frac": ach / peak,
                            # MACs the kernel ISSUES (composite-then-project and the folded colour layer removed ~20 % of the
                            # reference network's per-sample MACs: executed_mac) / the same launch time / the same peak
                            "executed_frac": exe / peak,
                            "step": step_object(line["algorithmic_tflop_per_step"],
                                                2 * (2 * executed_mac(sched, "fwd") + executed_mac(sched, "bwd")) * spr / 1e12,
                                                main_leg["ms_per_step"], field, step_bytes,
                                                main_leg["comm"]["allreduce_ms"] if "comm" in main_leg else None),
                            "traffic": traffic, "traffic_source": source,
                            "hbm_bytes_per_step": step_bytes,  # all kernels of a step, same PMC passes (null with traffic)
                            "avg_launch_ms": kern[dom]["avg_ms"],
                            "note": "achieved = algorithmic (fp32-equivalent) FLOPs / HIP-event launch time, averaged over "
                                    "the coarse and fine launches of eager steps run right after the timed region; peak = "
                                    + {"f16x3": "f16 dense MFMA 2516.6 TF / 3 MFMAs per fp32-accurate product (the hardware "
                                                "executes 3x the algorithmic FLOPs); for scale: fp32 MFMA peak is 157.3 TF",
                                       "f16": "f16 dense MFMA 2516.6 TF", "f32": "fp32 MFMA 157.3 TF"}[field]}
    if world == 1 and not args.no_cpu_baseline:
        nthreads = min(CPU_BASELINE_THREADS, os.cpu_count() or 1)
        prev = torch.get_num_threads()
        torch.set_num_threads(nthreads)
        v, sec = cpu_baseline(args.progress, B.n_images)
        torch.set_num_threads(prev)
        line["cpu_baseline"] = {"value": v, "unit": "rays/s", "cores": nthreads, "kind": "port",
                                "sample": f"oracle forward + backward + both Adam updates on 768 rays of the same "
                                          f"configuration (64+128 samples, 8x256 fields, pose opt ON), mean of 3 warm "
                                          f"iterations ({sec:.1f} s each) after 1 warm-up, torch threads = {nthreads} of "
                                          f"{os.cpu_count()} host threads (the fastest setting measured)"}
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
