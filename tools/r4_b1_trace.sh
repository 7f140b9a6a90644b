#!/bin/bash
# kernel timeline of the one-volume-per-GPU dual step (two streams): busy fraction, concurrency, per-kernel time on the critical union
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4_b1_trace; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --batch 1 --steps 6 --warmup 3 --no-extras --no-cpu-baseline --no-roofline --no-batch1 > $OUT/bench.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r4_b1_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:], r.get("Stream_Id", r.get("Queue_Id", ""))) for r in csv.DictReader(open(f))]
rows.sort()
# last 3 steps: take the final 45 % of kernels by time
t0, t1 = rows[0][0], rows[-1][1]
cut = t1 - int(3 * 15.7e6)
rows = [r for r in rows if r[0] >= cut]
span = rows[-1][1] - rows[0][0]
ev = []
for s, e, n, q in rows: ev += [(s, 1), (e, -1)]
ev.sort()
busy = two = 0; c = 0; last = ev[0][0]
for t, d in ev:
    if c >= 1: busy += t - last
    if c >= 2: two += t - last
    c += d; last = t
print(f"span {span/1e6:.2f} ms for ~3 steps; some kernel running {busy/span:.3f} of it; two or more {two/span:.3f}; idle {(1-busy/span)*span/3e6:.2f} ms per step")
tot = collections.Counter(); cnt = collections.Counter()
for s, e, n, q in rows: tot[n] += e - s; cnt[n] += 1
for n, v in tot.most_common(14): print(f"  {v/3e6:7.3f} ms/step  {cnt[n]//3:4d} x {v/cnt[n]/1e3:7.1f} us  {n}")
print("  sum of kernel times per step", sum(tot.values())/3e6)
PY
