#!/bin/bash
# round-6 profile set of the final code: rocprofv3 kernel stats, the default bench line as the driver runs it, the three PMC passes merged
# into profiles/r06_pmc.json (stamped with the library's hs_build_id and the bench value), and the other workloads' bench lines.
# Run on the GPU box from the repo root; results under gpurun_out/r06 (copy to profiles/ afterwards).
set -o pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r06}
mkdir -p gpurun_out/$TAG
bash tools/profile_bench.sh 53 gpurun_out/$TAG/stats &&
cp $(find gpurun_out/$TAG/stats -name "*kernel_stats.csv" | head -1) gpurun_out/$TAG/${TAG}_kernel_stats_dual_batch53.csv &&
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/${TAG}_bench_dual_batch53.json 2> gpurun_out/$TAG/bench.err &&
bash tools/collect_traffic.sh 53 gpurun_out/pmc_$TAG &&
bash tools/collect_util.sh 53 gpurun_out/pmc_$TAG &&
python3 tools/util_to_json.py gpurun_out/pmc_$TAG 53 dual256 $TAG gpurun_out/$TAG/${TAG}_bench_dual_batch53.json | tee gpurun_out/$TAG/pmc_summary.txt &&
cp profiles/${TAG}_pmc.json gpurun_out/$TAG/${TAG}_pmc.json
tail -c 400 gpurun_out/$TAG/${TAG}_bench_dual_batch53.json
