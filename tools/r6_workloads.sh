#!/bin/bash
# the other workloads' bench lines of round 6 (configs[1], configs[3] variants, native geometry, configs[4], the fp16 line)
set -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r06}
mkdir -p gpurun_out/$TAG
for W in vit256 clip_s1 clip_c4a clip_c4b native_dual; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 10 --warmup 3 --no-extras > gpurun_out/$TAG/${TAG}_bench_$W.json 2>> gpurun_out/$TAG/bench.err || exit 1
done
timeout -k 10 300 python3 bench.py --workload mrg_c5 --steps 10 --warmup 3 --no-extras > gpurun_out/$TAG/${TAG}_bench_mrg_c5.json 2>> gpurun_out/$TAG/bench.err || exit 1
timeout -k 10 300 python3 bench.py --precision f16 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > gpurun_out/$TAG/${TAG}_bench_dual_batch53_f16.json 2>> gpurun_out/$TAG/bench.err || exit 1
ls -la gpurun_out/$TAG
