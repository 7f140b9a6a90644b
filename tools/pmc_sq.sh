#!/bin/bash
# SQ wave-state counters of one kbench invocation: bash tools/pmc_sq.sh <outdir> <kernel-name filter> <kbench args...>
OUT=$1; FILT=$2; shift; shift
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $ROOT/$OUT/p1 -- $ROOT/tools/kbench "$@" > $ROOT/$OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/p2 -- $ROOT/tools/kbench "$@" > $ROOT/$OUT/p2.log 2>&1
cd $ROOT; python3 tools/pmc_summary.py $OUT/p1 "$FILT"; python3 tools/pmc_summary.py $OUT/p2 "$FILT" | awk '{print $1, $2}'
