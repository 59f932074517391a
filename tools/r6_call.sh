#!/bin/bash
out=gpurun_out/r6h; mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_parity.py tests/test_graph_step.py tests/test_trainer.py tests/test_hip_fullsize.py -q -x -m gpu > $out/pytest.log 2>&1; tail -6 $out/pytest.log
python tools/launch_sources.py 2>/dev/null | tail -12
R=$(pwd); O=$(realpath -m $out)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o k -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-timing > "$O/prof.log" 2>&1
cd "$R"
python tools/step_sequence.py "$O/prof" > "$O/step_sequence.txt" 2>&1
rm -f "$O/prof/"*kernel_trace.csv
head -1 $O/step_sequence.txt; grep -c "at::native\|Functor\|elementwise_kernel\|CatArray" $O/step_sequence.txt; grep "at::native\|Functor\|elementwise_kernel\|CatArray\|rocclr" $O/step_sequence.txt | cut -c1-150
tail -1 $O/prof.log | cut -c1-120
