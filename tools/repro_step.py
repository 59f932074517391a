"""Run-to-run reproducibility of a whole training step, per gradient tensor, in both field modes (diagnostic companion of
tests/test_hip_fullsize.py::test_training_step_is_bitwise_reproducible)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_hip_fullsize as t
from upnerf_amd import rendering as rd
for mode in ("f16x3", "f32"):
    rd.FIELD_MODE = mode
    sysm, batch = t._system(), t._batch()
    u = t._draws(sysm, 1)
    sink1, sink2 = {}, {}
    rd._DEBUG_SINK = sink1
    l1, _, r1, g1 = t._loss_and_grads(sysm, batch, u)
    s1 = {k: v.clone() for k, v in sink1.items() if v is not None}
    rd._DEBUG_SINK = sink2
    l2, _, r2, g2 = t._loss_and_grads(sysm, batch, u)
    rd._DEBUG_SINK = None
    bad = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    print(mode, "differing grads:", len(bad), "of", len(g1))
    for k in bad[:40]:
        d = (g1[k] - g2[k]).abs().max() / (g1[k].abs().max() + 1e-30)
        print("   ", k, float(d))
    # the sink holds the LAST pass's buffers (coarse pass backward runs last)
    for k in s1:
        if k in sink2 and sink2[k] is not None and not torch.equal(s1[k], sink2[k]):
            print("   sink differs:", k)
