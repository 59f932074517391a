set -u
mkdir -p gpurun_out/c1
(timeout 900 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -150) > gpurun_out/c1/pytest.log 2>&1
bash tools/r3_ab_libs.sh "_base - _st1 _st2 _st3" "64" 2 > gpurun_out/c1/ab.log 2>&1
tail -5 gpurun_out/c1/pytest.log; cat gpurun_out/c1/ab.log
