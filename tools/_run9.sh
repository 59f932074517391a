cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/t9.txt
bash tools/_run8.sh; cat gpurun_out/t8.txt >> gpurun_out/t9.txt
