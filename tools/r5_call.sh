# scratch driver of the round's gpurun calls (rewritten per call)
mkdir -p gpurun_out/full
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/full/pytest.log 2>&1; tail -6 gpurun_out/full/pytest.log
UPNERF_ZERO_POOL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"
