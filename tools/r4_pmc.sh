#!/bin/bash
# round-4 PMC collection of the default bench (53 volumes): FETCH_SIZE | WRITE_SIZE | SQ+GRBM passes, merged into profiles/r04_pmc.json
set -e
TAG=${1:-r04}
bash tools/collect_traffic.sh 53 gpurun_out/pmc_$TAG
bash tools/collect_util.sh 53 gpurun_out/pmc_$TAG
python3 tools/util_to_json.py gpurun_out/pmc_$TAG 53 dual256 $TAG | tee gpurun_out/pmc_$TAG/summary.txt
cp profiles/${TAG}_pmc.json gpurun_out/${TAG}_pmc.json
