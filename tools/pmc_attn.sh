#!/bin/bash
# PMC passes over tools/kbench (no torch): bash tools/pmc_attn.sh <mode> <variant> <outdir>
MODE=${1:-attnf}; V=${2:-1}; OUT=${3:-gpurun_out/pmc_attn}
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $ROOT/$OUT/p1 -- $ROOT/tools/kbench $MODE 10 3 $V $V > $ROOT/$OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT/p2 -- $ROOT/tools/kbench $MODE 10 3 $V $V > $ROOT/$OUT/p2.log 2>&1
echo done
