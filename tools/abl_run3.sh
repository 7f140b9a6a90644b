#!/bin/bash
# steady-state per-kernel time of one attention-backward variant under each ablation library:
#   bash tools/abl_run3.sh <outdir> <batch> <variant> <lib names...>
OUT=$1; B=$2; VAR=$3; shift; shift; shift
ROOT=$(pwd); mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for V in "$@"; do
  LIB=$ROOT/hsenet_amd/lib/variants/libhsenet_$V.so
  [ "$V" = base ] && LIB=$ROOT/hsenet_amd/lib/libhsenet_hip.so
  HS_LIB=$LIB HS_PRESCALED=1 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/$V -- $ROOT/tools/kbench attnb $B 4 $VAR $VAR > $ROOT/$OUT/$V.log 2>&1
  python3 - "$ROOT/$OUT/$V" "$V" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    if "attn_bwd" in n: d[n[-26:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    v = sorted(v)[: max(1, len(v) // 2)]          # fastest half = warmed-up clock
    print(f"{sys.argv[2]:8s} {n:26s} n={len(d[n]):3d} fast-half mean {sum(v)/len(v):8.1f} us")
PY
done
