#!/bin/bash
# rocprofv3 kernel statistics of one bench.py run (the summary that goes to profiles/): bash tools/profile_bench.sh <batch> <outdir>
B=${1:-53}; OUT=${2:-gpurun_out/prof_bench}
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT -- python3 $ROOT/bench.py --steps 3 --warmup 1 --batch $B --no-cpu-baseline --no-batch1 --no-extras ${WORKLOAD:+--workload $WORKLOAD} > $ROOT/$OUT/bench.log 2>&1
echo "profile done"
