#!/bin/bash
# round-4 profile set of the final code: rocprofv3 kernel stats, the three PMC passes, the default bench line as the driver runs it, and
# the other workloads' bench lines.  Run on the GPU box from the repo root; results under gpurun_out/ (copied to profiles/ afterwards).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04
bash tools/profile_bench.sh 53 gpurun_out/r04/stats &&
cp $(find gpurun_out/r04/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r04/r04_kernel_stats_dual_batch53.csv &&
bash tools/r4_pmc.sh r04 &&
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/r04_bench_dual_batch53.json 2> gpurun_out/r04/bench.err &&
for W in vit256 clip_s1 clip_c4a clip_c4b native_dual; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 10 --warmup 3 --no-extras > gpurun_out/r04/r04_bench_$W.json 2>> gpurun_out/r04/bench.err || exit 1
done
tail -c 600 gpurun_out/r04/r04_bench_dual_batch53.json
