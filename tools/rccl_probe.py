"""What can be said about the RCCL path on a ONE-GPU box (no N > 1 hardware run exists).

    python tools/rccl_probe.py            # part A: one rank, backend "nccl" (= RCCL): every collective call the N > 1 step makes
    python tools/rccl_probe.py --two      # part B: two ranks on the same device (RCCL is expected to refuse; the error is printed)
    python tools/rccl_probe.py --step     # part C: bench.py's replayed configs[1] training step with its exchange through RCCL

Part A drives upnerf_amd.parallel.GradSync with its world-size short cuts disabled (a subclass reporting world = 2 over a
one-rank group: the sums are those of one rank, the calls, streams, events and buffer handling are the real ones): the blocking
flat all-reduce, the asynchronous early bucket on its side stream with the wait at the end of backward, the fp64 MAX reduction and
the barrier of bench.py, and the comm attribution read back afterwards."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def part_a():
    import torch
    import torch.distributed as dist
    from upnerf_amd import parallel
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)

    class TwoSync(parallel.GradSync):
        @property
        def world(self):
            return 2

    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n, device=dev)) for n in (1_000_000, 250_000, 1_000_000, 3)]
    sync = TwoSync(ps, early=ps[:2])
    sync.comm_reset(timing=True)
    for step in range(3):
        for p in ps:
            p.grad = None
        sync.begin("phase")
        loss = sum((p * p).sum() for p in ps)
        loss.backward()
        want = [p.grad.clone() for p in ps]
        n = sync()
        torch.cuda.synchronize()
        for p, w in zip(ps, want):
            assert torch.equal(p.grad, w * 0.5), "sum over one rank / reported world 2"
        print(f"step {step}: {n} floats exchanged, early launches so far {sync.stats['early_launches']}")
    t = torch.tensor([1.5, 2.5, 3.5], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    print("comm:", sync.comm_summary(3))
    print("backend:", dist.get_backend(), "| torch", torch.__version__, "| nccl", ".".join(map(str, torch.cuda.nccl.version())))
    dist.destroy_process_group()
    print("part A OK")


def part_c():
    """The real step (graph 1 -> eager all-reduce -> graph 2) with the all-reduce going through RCCL: one
    rank, GradSync told the world is 2 (so it packs, reduces, scales and unpacks; the sum is this rank's own).  Beside it the same
    step without the exchange: the difference is what the serial placement costs on one device (no link involved)."""
    import time
    import torch
    import torch.distributed as dist
    import bench
    from upnerf_amd import parallel, rendering
    from upnerf_amd.graph_step import GraphedTrainingStep
    os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    rendering.FIELD_MODE = "f16x3"
    out = {}
    for tag in ("single", "exchange"):
        if tag == "exchange":
            parallel.GradSync.world = property(lambda self: 2)
        sysm = bench.build_system(dev, 0.3, 4096, 763, None, None)
        if tag == "exchange":
            sysm.enable_data_parallel()
        batches = bench.make_batches(dev, 4, 100, 4096, 763)
        step = GraphedTrainingStep(sysm)
        for i in range(5):
            step(batches[i % 4], i)
        if tag == "exchange":
            sysm.grad_sync.comm_reset(timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for i in range(n):
            r = step(batches[i % 4], i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        out[tag] = ms
        loss = r[0] if isinstance(r, (tuple, list)) else r
        print(f"{tag}: {ms:.3f} ms per step, {4096 / ms:.1f} k rays/s, last loss {float(loss):.6f}, graph stats {dict(step.stats)}")
        if tag == "exchange":
            print("comm:", sysm.grad_sync.comm_summary(n))
    print(f"exchange placed between the two graphs costs {out['exchange'] - out['single']:.3f} ms per step on one device")
    dist.destroy_process_group()
    print("part C OK")


def part_b_rank():
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"])
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
        t = torch.ones(1024, device="cuda") * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"rank {rank}: two ranks on one device all-reduced: {float(t[0])}")
    except Exception as e:  # noqa: BLE001 -- the point is to print what RCCL says
        print(f"rank {rank}: refused: {type(e).__name__}: {str(e)[:300]}")


if __name__ == "__main__":
    if "--rank" in sys.argv:
        part_b_rank()
    elif "--two" in sys.argv:
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=free_port(), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank"], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
                 for r in range(2)]
        for p in procs:
            try:
                p.wait(timeout=120)
            except subprocess.TimeoutExpired:
                p.kill()
                print("timed out (killed)")
    elif "--step" in sys.argv:
        part_c()
    else:
        part_a()
