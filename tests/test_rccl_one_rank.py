"""The RCCL calls of the N > 1 training step on the hardware a one-GPU box has: ONE rank (RCCL refuses two ranks on one device).
tools/rccl_probe.py drives upnerf_amd.parallel.GradSync through backend "nccl" with its world-size short cuts off, and runs
bench.py's replayed configs[1] step as graph 1 -> RCCL all-reduce -> graph 2.  The second found a crash in round 6 that the gloo
tests could not: the process group's watchdog polling events while the step was being captured (graph_step.py drains it now).
Separate processes: a process group and its watchdog live until the process ends."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    env = dict(os.environ, PYTHONUNBUFFERED="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_probe.py"), *args], env=env, capture_output=True, text=True,
                          timeout=600)


def test_grad_sync_through_rccl_on_one_rank():
    """Blocking flat all-reduce, asynchronous early bucket on its side stream + the wait at the end of backward, the fp64 MAX
    reduction and the barrier of bench.py, the comm attribution: real RCCL calls, sums of one rank."""
    r = run()
    assert r.returncode == 0 and "part A OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
    assert "early launches so far 2" in r.stdout


def test_replayed_step_with_its_exchange_through_rccl_on_one_rank():
    """An eager step's all-reduce immediately followed by the capture of the next step -- the order of every training run, and the
    one that took the process down (hipErrorCapturedEvent from the watchdog thread) before the capture drained the watchdog --
    then twenty replays with the exchange between the two graphs."""
    r = run("--step")
    assert r.returncode == 0 and "part C OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    assert "'captures': 1, 'replays': 24" in r.stdout
