#!/usr/bin/env python3
"""Locator: the training loop with graph replay vs eager launches on the same batches; prints the first step after which
the two systems differ, the kind of step (eager / capture / replay) and the parameters that differ.
--val / --sync / --ckpt add the validation render, a host sync and a checkpoint read every third step (what Trainer.fit
does between steps); --separate runs the eager system to the end first and the graphed one alone afterwards."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_trainer import _scene, _system  # noqa: E402
from test_sampler import _sampler  # noqa: E402
from upnerf_amd.graph_step import GraphedTrainingStep  # noqa: E402
from upnerf_amd.checkpoint import checkpoint_dict  # noqa: E402

VAL, SYNC, CKPT, SEP = ("--val" in sys.argv, "--sync" in sys.argv, "--ckpt" in sys.argv, "--separate" in sys.argv)
bufs, I = _scene()
smp = _sampler(bufs)
val = {k: v[None] for k, v in smp.sample(torch.arange(0, 160)).items()}
val2 = [{k: v[None] for k, v in smp.sample(torch.arange(i * 160, (i + 1) * 160)).items()} for i in range(2)]
BATCHES = [(e, i, b) for e in range(2) for i, b in enumerate(smp.batches(128, seed=3, epoch=e))]


KEEP = []


def between(s, n):
    if "--trainer-val" in sys.argv and n % 3 == 2:
        outs = [s.validation_step(v, i) for i, v in enumerate(val2)]
        if "--keep" in sys.argv:
            KEEP.append([o.get("results") for o in outs])
        for o in outs:
            o.pop("results", None)
        res = s.validation_epoch_end(outs)
        if "--sync" in sys.argv:
            {k: float(v) for k, v in res.items()}
    if VAL and n % 3 == 2:
        o = s.validation_step(val, 0)
        if SYNC:
            float(o["val_psnr"])
    if CKPT and n % 3 == 2:
        checkpoint_dict(s)


def snap(s):
    return {k: v.clone() for k, v in s.state_dict().items()}


torch.manual_seed(11); b = _system(I)
ref = []
if SEP:
    for n, (_, i, batch) in enumerate(BATCHES):
        b.training_step(batch, i)
        ref.append(snap(b))
        between(b, n)
torch.manual_seed(11); a = _system(I)
ga = GraphedTrainingStep(a)
for n, (_, i, batch) in enumerate(BATCHES):
    before, key = dict(ga.stats), ga.key(batch)
    ga(batch, i)
    if not SEP:
        b.training_step(batch, i)
        ref.append(snap(b))
    kind = [k for k in ga.stats if ga.stats[k] != before[k]]
    bad = [k for k, p in a.state_dict().items() if not torch.equal(p, ref[n][k])]
    print(n, key, kind, "DIFF " + ",".join(bad[:6]) if bad else "same", flush=True)
    if bad:
        sys.exit(1)
    between(a, n)
    if not SEP:
        between(b, n)
