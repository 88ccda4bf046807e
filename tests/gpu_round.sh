#!/bin/bash
# Round-end measurement run on the GPU box: parity tests, smoke, the two bench
# sizes, the rocprofv3 kernel-stats summary of the bench command, the in-kernel
# phase profile.  Usage: bash tests/gpu_round.sh TAG
set -e -o pipefail
TAG=${1:-r01_x}
cd "${GRAFT_REPO_ROOT:-.}"
R=$PWD
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$TAG.log 2>&1 || echo "pytest FAILED"
tail -2 gpurun_out/pytest_gpu_$TAG.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
tail -1 gpurun_out/smoke_$TAG.log
timeout -k 10 400 python bench.py > gpurun_out/bench_${TAG}_B1024.json 2> gpurun_out/bench_err.log
cat gpurun_out/bench_${TAG}_B1024.json
timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > gpurun_out/bench_${TAG}_B8192.json 2>> gpurun_out/bench_err.log
cat gpurun_out/bench_${TAG}_B8192.json
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 5 --no-cpu-baseline --seed-sweep 0 --workers 1 > $R/gpurun_out/prof_$TAG.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_B8192 -- python3 $R/bench.py --config 2 --steps 5 --no-cpu-baseline --workers 1 > $R/gpurun_out/prof_${TAG}_B8192.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_N30 -- python3 $R/bench.py --config 3 --batch 2048 --steps 3 --warmup 1 --no-cpu-baseline --workers 1 > $R/gpurun_out/prof_${TAG}_N30.log 2>&1
cd $R
find gpurun_out/prof_$TAG -name '*kernel_stats.csv' -exec cat {} \;
timeout -k 10 300 python tests/gpu_profile_phases.py > gpurun_out/phases_$TAG.log 2>&1 || true
tail -40 gpurun_out/phases_$TAG.log
# teams (round 4): the batch sizes that leave SIMDs idle -- one wave per problem against a team of four, phase stamps of both at B = 256,
# the drop-in's single call
timeout -k 10 300 python tests/gpu_profile_phases.py 256 10 one > gpurun_out/phases_${TAG}_B256_one_wave.log 2>&1 || true
timeout -k 10 300 python tests/gpu_profile_phases.py 256 10 team > gpurun_out/phases_${TAG}_B256_team.log 2>&1 || true
timeout -k 10 300 python tests/gpu_team.py 1 16 64 256 > gpurun_out/team_${TAG}.log 2>&1 || true
tail -12 gpurun_out/team_${TAG}.log
timeout -k 10 300 python tests/gpu_single_latency.py > gpurun_out/single_call_latency_${TAG}.txt 2>&1 || true
tail -4 gpurun_out/single_call_latency_${TAG}.txt
# round 6: pairs (two waves per problem on the one-wave budget, batches 256 < B <= 512) and the longest-expected-first work queue (configs[3])
timeout -k 10 300 python tests/gpu_profile_phases.py 512 10 pair > gpurun_out/phases_${TAG}_B512_pair.log 2>&1 || true
timeout -k 10 300 python tests/gpu_profile_phases.py 512 10 one > gpurun_out/phases_${TAG}_B512_one_wave.log 2>&1 || true
timeout -k 10 300 python tests/gpu_pair.py 257 384 512 > gpurun_out/pair_${TAG}.log 2>&1 || true
tail -4 gpurun_out/pair_${TAG}.log
timeout -k 10 400 python tests/gpu_queue_order.py > gpurun_out/queue_order_${TAG}.log 2>&1 || true
tail -4 gpurun_out/queue_order_${TAG}.log
