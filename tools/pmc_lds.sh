#!/bin/bash
# LDS-side counters of one kbench invocation: bash tools/pmc_lds.sh <outdir> <kbench args...>
OUT=$1; shift
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/$OUT -- $ROOT/tools/kbench "$@" > $ROOT/$OUT/run.log 2>&1
cd $ROOT; python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:46]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8            # cycles of one XCD's clock over the kernel
    if gui < 1e4: continue
    print(k)
    for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_WAIT_INST_LDS", "SQ_INSTS_LDS"):
        print("    %-24s %.3e   per CU-cycle: %.3f" % (c, m.get(c, 0), m.get(c, 0) / (gui * 256)))
PY
