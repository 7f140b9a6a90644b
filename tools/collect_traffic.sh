#!/bin/bash
# HBM traffic per launch of every hsenet kernel class, from two separate rocprofv3 --pmc passes (FETCH_SIZE costs 3
# TCC slots, WRITE_SIZE 2: they do not fit one pass).  Run on the GPU box from the repo root:
#     bash tools/collect_traffic.sh <batch> <outdir>
# then  python tools/traffic_to_json.py <outdir> <batch>  writes profiles/r02_traffic.json.
# FETCH_SIZE is doubled when converting (MI355X_MICROARCH.md "HBM": gfx950 tallies 128-B requests at 64 B).
set -e
B=${1:-53}; OUT=${2:-gpurun_out/traffic}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/$C -- python3 $ROOT/bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-roofline --no-batch1 --no-extras ${WORKLOAD:+--workload $WORKLOAD} > $ROOT/$OUT/$C.log 2>&1
  echo "$C pass done"
done
