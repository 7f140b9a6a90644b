#!/bin/bash
# MFMA utilisation per kernel class of one bench.py run, from a rocprofv3 --pmc pass of its own (no trace domains besides the
# kernel trace).  Run on the GPU box from the repo root, after tools/collect_traffic.sh (same batch, same outdir):
#     bash tools/collect_util.sh <batch> <outdir>
# then  python tools/util_to_json.py <outdir> <batch> [workload] [round tag]  merges both into profiles/<tag>_pmc.json
set -e
B=${1:-53}; OUT=${2:-gpurun_out/traffic}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/UTIL -- python3 $ROOT/bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-roofline --no-batch1 --no-extras ${WORKLOAD:+--workload $WORKLOAD} > $ROOT/$OUT/UTIL.log 2>&1
echo "UTIL pass done"
