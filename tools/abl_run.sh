#!/bin/bash
# per-kernel times of the attention backward under each variant library: bash tools/abl_run.sh <outdir> <batch> <variant names...>
OUT=$1; B=$2; shift; shift
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  LIB=$ROOT/hsenet_amd/lib/variants/libhsenet_$V.so
  [ "$V" = base ] && LIB=$ROOT/hsenet_amd/lib/libhsenet_hip.so
  HS_LIB=$LIB HS_PRESCALED=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/$V -- $ROOT/tools/kbench attnb $B 2 0 0 > $ROOT/$OUT/$V.log 2>&1
  python3 - "$ROOT/$OUT/$V" "$V" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "attn" in n:
        print(f"{sys.argv[2]:8s} {n.split('(')[0][-28:]:28s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
