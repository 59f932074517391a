"""Data-parallel gradient averaging (upnerf_amd/parallel.py) on CPU with the gloo backend, world_size 2:
the N>1 path of bench.py / NeRFSystem.enable_data_parallel() without GPUs.

Checked: (1) the flat all-reduce equals the mean of the per-rank gradients for every parameter that has one;
(2) parameters without a gradient (phase-dependent heads, `progress`, TransientNet.rgb_layer -- SURVEY.md Q12) are
left untouched and do not desynchronise the ranks; (3) two ranks that each see half of a batch end up with the
gradient a single process computes on the full batch (mean-of-means with equal shards)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.mlp = torch.nn.Sequential(torch.nn.Linear(7, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
        self.unused = torch.nn.Parameter(torch.zeros(5))      # never receives a gradient
        self.table = torch.nn.Embedding(10, 4)                # row-sparse gradient, reduced densely


def _model():
    torch.manual_seed(0)
    return _Net()


def _loss(m, x, idx):
    return (m.mlp(x) ** 2).mean() + (m.table(idx) ** 2).mean()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from upnerf_amd import parallel
    r, _, w = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    m = _model()
    g = torch.Generator().manual_seed(123)
    x, idx = torch.randn(8, 7, generator=g), torch.randint(0, 10, (8,), generator=g)
    xs, ids = x[rank * 4:(rank + 1) * 4], idx[rank * 4:(rank + 1) * 4]
    _loss(m, xs, ids).backward()
    local = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    sync = parallel.GradSync(m.parameters(), check=True)
    n = sync()
    out = {n_: p.grad.clone() for n_, p in m.named_parameters() if p.grad is not None}
    # numpy arrays travel by value; torch tensors would travel as shared-memory handles that die with this process
    q.put((rank, n, {k: v.numpy().copy() for k, v in local.items()}, {k: v.numpy().copy() for k, v in out.items()},
           m.unused.grad is None))
    dist.barrier()
    dist.destroy_process_group()


def _run_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    return res


def test_flat_allreduce_world2_gloo():
    try:
        res = _run_world2()
    except Exception:  # a rendezvous port can be taken between _free_port() and bind: one retry on a new port
        res = _run_world2()
    res = [(r, n, {k: torch.from_numpy(v) for k, v in a.items()}, {k: torch.from_numpy(v) for k, v in b.items()}, u)
           for r, n, a, b, u in res]
    (_, n0, loc0, out0, un0), (_, n1, loc1, out1, un1) = res
    assert n0 == n1 == sum(v.numel() for v in loc0.values())
    assert un0 and un1                                   # unused parameter still has no gradient
    for k in loc0:
        mean = (loc0[k] + loc1[k]) / 2
        assert torch.allclose(out0[k], mean, atol=1e-7) and torch.equal(out0[k], out1[k]), k
    # equal shards: mean of shard gradients == full-batch gradient
    m = _model()
    g = torch.Generator().manual_seed(123)
    x, idx = torch.randn(8, 7, generator=g), torch.randint(0, 10, (8,), generator=g)
    _loss(m, x, idx).backward()
    for n, p in m.named_parameters():
        if p.grad is not None:
            assert torch.allclose(out0[n], p.grad, atol=1e-6), n


def test_single_process_is_a_noop():
    from upnerf_amd import parallel
    m = _model()
    _loss(m, torch.randn(4, 7), torch.randint(0, 10, (4,))).backward()
    before = m.mlp[0].weight.grad.clone()
    assert parallel.GradSync(m.parameters())() == 0
    assert torch.equal(m.mlp[0].weight.grad, before)


class _TwoFields(torch.nn.Module):
    """NeRFSystem-shaped parameter list: a `fine` field whose gradients autograd completes first (the early bucket), a
    `coarse` field, a head that receives a gradient only in some phases, a per-image table."""

    def __init__(self):
        super().__init__()
        self.coarse = torch.nn.Linear(6, 6)
        self.fine = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 2))
        self.fine_head = torch.nn.Linear(6, 2)      # part of the fine field, used in phase 1 only
        self.table = torch.nn.Embedding(9, 6)


def _loss2(m, idx, phase):
    h = torch.tanh(m.coarse(m.table(idx)))
    out = m.fine(h)
    if phase == 1:
        out = out + m.fine_head(h)
    return (out ** 2).mean()


def _worker_overlap(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from upnerf_amd import parallel
    parallel.init_from_env("gloo")
    torch.manual_seed(0)
    m = _TwoFields()
    early = list(m.fine.parameters()) + list(m.fine_head.parameters())
    sync = parallel.GradSync(m.parameters(), check=True, early=early)
    g = torch.Generator().manual_seed(7)
    idx_all = torch.randint(0, 9, (6, 8), generator=g)
    out = []
    for step, phase in enumerate([0, 0, 1, 1, 0, 1]):  # first step of a phase learns its count, later ones launch early
        for p in m.parameters():
            p.grad = None
        sync.begin(phase)
        idx = idx_all[step][rank * 4:(rank + 1) * 4]
        _loss2(m, idx, phase).backward()
        n = sync()
        out.append((n, {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in m.named_parameters()}))
    q.put((rank, out, dict(sync.stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_early_bucket_allreduce_world2_gloo():
    """The fine field's gradients are reduced from a post-accumulate-grad hook while backward still runs (overlap with the
    tail of backward; the reference gets this from DDP, train.py:70-72): same averages as the flat reduce,
    phase-dependent parameter sets handled (the head without a gradient in phase 0 stays None on both ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    (_, out0, st0), (_, out1, st1) = res
    assert st0 == st1 == {"early_launches": 4, "late_only": 2}, (st0, st1)
    # single-process reference on the full batches
    torch.manual_seed(0)
    m = _TwoFields()
    g = torch.Generator().manual_seed(7)
    idx_all = torch.randint(0, 9, (6, 8), generator=g)
    for step, phase in enumerate([0, 0, 1, 1, 0, 1]):
        for p in m.parameters():
            p.grad = None
        _loss2(m, idx_all[step], phase).backward()
        n0, g0 = out0[step]
        n1, g1 = out1[step]
        assert n0 == n1 == sum(p.grad.numel() for p in m.parameters() if p.grad is not None)
        for k, p in m.named_parameters():
            if p.grad is None:
                assert g0[k] is None and g1[k] is None, (step, k)
            else:
                a, b = torch.from_numpy(g0[k]), torch.from_numpy(g1[k])
                assert torch.equal(a, b) and torch.allclose(a, p.grad, atol=1e-6), (step, k)


def _worker_overlap8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from upnerf_amd import parallel
    parallel.init_from_env("gloo")
    torch.manual_seed(0)
    m = _TwoFields()
    early = list(m.fine.parameters()) + list(m.fine_head.parameters())
    sync = parallel.GradSync(m.parameters(), check=True, early=early)
    g = torch.Generator().manual_seed(11)
    per = 2
    idx_all = torch.randint(0, 9, (len(_PHASES8), per * world), generator=g)
    out = []
    for step, phase in enumerate(_PHASES8):
        for p in m.parameters():
            p.grad = None
        sync.begin(phase)
        _loss2(m, idx_all[step][rank * per:(rank + 1) * per], phase).backward()
        n = sync()
        out.append((n, dict(sync._expected), {k: (None if p.grad is None else p.grad.numpy().copy()) for k, p in m.named_parameters()}))
    q.put((rank, out, dict(sync.stats)))
    dist.barrier()
    dist.destroy_process_group()


_PHASES8 = [0, 0, 0, 1, 1, 0, 1, 1]  # phase boundary 0 -> 1 after three steps, and back (a resumed run re-enters a phase)


def test_grad_sync_world8_gloo_across_a_phase_boundary():
    """the exchange the first 8-GPU run will make, pinned on CPU -- eight gloo ranks, the early (fine-field)
    bucket launched from the gradient hook, the schedule phase changing under it: the number of early gradients is re-learned
    per phase (first step of a phase = late-only on EVERY rank, the same steps on every rank), the set of parameters with a
    gradient is identical on all ranks in every step, and every rank ends every step with the gradient one
    process computes on the whole batch."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    stats = [st for _, _, st in res]
    # steps 0 and 3 are the first of their phase: learned there, launched early in the six others
    assert all(st == {"early_launches": 6, "late_only": 2} for st in stats), stats
    torch.manual_seed(0)
    m = _TwoFields()
    g = torch.Generator().manual_seed(11)
    idx_all = torch.randint(0, 9, (len(_PHASES8), 2 * world), generator=g)
    n_fine = len(list(m.fine.parameters()))
    for step, phase in enumerate(_PHASES8):
        for p in m.parameters():
            p.grad = None
        _loss2(m, idx_all[step], phase).backward()
        want_n = sum(p.grad.numel() for p in m.parameters() if p.grad is not None)
        live = {k for k, p in m.named_parameters() if p.grad is not None}
        for rank, out, _ in res:
            n, expected, grads = out[step]
            assert n == want_n, (step, rank, n, want_n)
            assert {k for k, v in grads.items() if v is not None} == live, (step, rank)  # Q12: same set on every rank
            assert expected[phase] == n_fine + (2 if phase == 1 else 0), (step, rank, expected)  # learned per phase key
            for k, p in m.named_parameters():
                if p.grad is not None:
                    a = torch.from_numpy(grads[k])
                    assert torch.equal(a, torch.from_numpy(res[0][1][step][2][k])), (step, rank, k)  # bitwise equal across ranks
                    assert torch.allclose(a, p.grad, atol=1e-6), (step, rank, k)
