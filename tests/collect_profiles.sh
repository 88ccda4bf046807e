#!/bin/bash
# Copies what tests/gpu_round.sh TAG and tests/gpu_pmc.sh TAG left under gpurun_out/ into profiles/ under the names profiles/README.md lists.
# Usage (in the container, after the gpurun calls): bash tests/collect_profiles.sh TAG
set -e
TAG=${1:?tag}
G=gpurun_out; P=profiles
cp $G/bench_${TAG}_B1024.json $P/${TAG}_bench_B1024.json; cp $G/bench_${TAG}_B8192.json $P/${TAG}_bench_B8192.json
for sfx in "" _B8192 _N30; do f=$(ls -t $(find $G/prof_${TAG}${sfx} -name '*kernel_stats.csv') | head -1); [ -n "$f" ] && cp $f $P/${TAG}_kernel_stats${sfx}.csv; done
cp $G/phases_${TAG}.log $P/${TAG}_phase_cycles.log
for v in B256_one_wave B256_team B512_one_wave B512_pair; do cp $G/phases_${TAG}_$v.log $P/${TAG}_phase_cycles_$v.log; done
cp $G/pair_${TAG}.log $P/${TAG}_pair_vs_one_wave.txt; cp $G/team_${TAG}.log $P/${TAG}_team_vs_one_wave.txt
cp $G/single_call_latency_${TAG}.txt $P/${TAG}_single_call_latency.txt; cp $G/queue_order_${TAG}.log $P/${TAG}_queue_order.txt
cp $G/pytest_gpu_${TAG}.log $P/${TAG}_pytest_gpu.log
if [ -f $G/pmc_${TAG}/pmc_current.json ]; then cp $G/pmc_${TAG}/pmc_current.json $P/pmc_current.json; cp $G/pmc_${TAG}/summary.txt $P/${TAG}_pmc_summary.txt; fi
sed -i '/amdgpu.ids/d' $P/${TAG}_*.txt $P/${TAG}_*.log
python tests/kernel_resources.py > $P/${TAG}_kernel_resources.txt 2>/dev/null || echo "kernel_resources.py failed"
git status --short $P | head -40
