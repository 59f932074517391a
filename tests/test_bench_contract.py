"""bench.py's output contract: ONE JSON line on stdout with the driver's keys, the roofline and cpu_baseline objects, and
values that are consistent with each other (value = rays per step / time per step)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_default_shape_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines  # exactly one line, and it is JSON
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["metric"] == "training rays/sec" and d["unit"] == "rays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    rays = d["config"]["rays_per_gpu"]
    assert rays == 4096 and d["config"]["N_samples"] == 64 and d["config"]["N_importance"] == 128
    assert abs(d["value"] - rays / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert 50_000 < d["value"] < 5_000_000
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "rays/s" and c["cores"] >= 1 and c["value"] > 0
    assert isinstance(c["sample"], str) and c["sample"]
    assert d["value"] / c["value"] > 10  # north_star: >= 10x the CPU path
