#!/bin/bash
# per-kernel durations of one kbench invocation: bash tools/ktrace.sh <outdir> <kbench args...>
OUT=$1; shift
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT -- $ROOT/tools/kbench "$@" > $ROOT/$OUT/run.log 2>&1
cd $ROOT; python3 tools/trace_shapes.py $OUT
