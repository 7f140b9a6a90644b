#!/bin/bash
# A/B of attention-backward kernel variants (hs_debug_set keys via kbench's variant number = 10*dkdv + dq):
#   bash tools/abl_run2.sh <outdir> <batch> <variantA> <variantB>
OUT=$1; B=$2; VA=$3; VB=$4
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
HS_PRESCALED=1 $ROOT/tools/kbench attnb $B 3 $VA $VB > $ROOT/$OUT/ab.log 2>&1; head -4 $ROOT/$OUT/ab.log
HS_PRESCALED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -- $ROOT/tools/kbench attnb $B 3 $VA $VB > $ROOT/$OUT/prof.log 2>&1
python3 - "$ROOT/$OUT/prof" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "attn_bwd" in n:
        print(f"{n.split('(')[0][-30:]:30s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
