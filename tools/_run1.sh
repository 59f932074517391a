cd /root/repo
timeout 300 python tools/sparsity_probe.py > gpurun_out/sparsity0.txt 2>&1
timeout 300 python tools/sparsity_probe.py --after 30 > gpurun_out/sparsity30.txt 2>&1
bash tools/r3_ab_libs.sh "- _half36 _half32 -" "64" 1 > gpurun_out/ab_half.txt 2>&1
