R=/root/repo; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/seq -o k -- python3 $R/bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-kernel-timing --no-configs34 > $R/gpurun_out/seq.log 2>&1
cd $R; python tools/step_sequence.py gpurun_out/seq > gpurun_out/step_sequence.txt 2>&1; rm -rf gpurun_out/seq
