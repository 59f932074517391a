#!/bin/bash
# Copies the judged summaries of a tools/final_profile.sh run into profiles/ under a round tag:
#   bash tools/copy_profiles.sh gpurun_out/final_r03d r03
set -e
S=$1; T=$2; P=profiles
cp $S/bench.json $P/${T}_bench.json
cp $S/bench_trevi.json $P/${T}_trevi_bench.json
cp $S/kernel_summary.txt $P/${T}_kernel_summary.txt
cp $S/kernel_summary_trevi.txt $P/${T}_trevi_kernel_summary.txt
cp $S/prof/k_kernel_stats.csv $P/${T}_kernel_stats.csv
cp $S/prof_trevi/k_kernel_stats.csv $P/${T}_trevi_kernel_stats.csv
cp $S/pmc.json $P/${T}_pmc.json; cp $S/pmc.md $P/${T}_pmc_summary.md
cp $S/pmc_trevi.json $P/${T}_trevi_pmc.json; cp $S/pmc_trevi.md $P/${T}_trevi_pmc_summary.md
cp $S/pmc_current.json $P/pmc_current.json
cp $S/pmc_current_trevi.json $P/pmc_current_trevi.json
[ -f $S/step_sequence.txt ] && cp $S/step_sequence.txt $P/${T}_step_sequence.txt
[ -f $S/step_sequence_trevi.txt ] && cp $S/step_sequence_trevi.txt $P/${T}_trevi_step_sequence.txt
[ -f gpurun_out/sparsity0.txt ] && cat gpurun_out/sparsity0.txt gpurun_out/sparsity30.txt 2>/dev/null | grep -v amdgpu.ids > $P/${T}_relu_sparsity.txt
ls $P | grep ${T}_
