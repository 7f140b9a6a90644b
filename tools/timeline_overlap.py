"""Concurrency of a rocprofv3 --kernel-trace: python tools/timeline_overlap.py <kernel_trace.csv> [t0_frac t1_frac]
For the window [t0, t1] of the trace (fractions of its span; default the last 40 %): wall time, the share of it with 0 / 1 / >= 2 kernels
in flight, the sum of kernel durations, and per kernel name the mean duration when it ran alone vs overlapped with another kernel."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows)
lo, hi = ev[0][0], max(e[1] for e in ev)
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.6, 1.0)
w0, w1 = lo + (hi - lo) * f0, lo + (hi - lo) * f1
ev = [e for e in ev if e[0] >= w0 and e[1] <= w1]
pts = sorted([(s, 1) for s, _, _ in ev] + [(e, -1) for _, e, _ in ev])
depth, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[min(depth, 2)] += t - last
    depth, last = depth + d, t
wall = pts[-1][0] - pts[0][0]
busy = sum(e - s for s, e, _ in ev)
print(f"{len(ev)} kernels in {wall / 1e6:.2f} ms; sum of durations {busy / 1e6:.2f} ms; in flight 0: {100 * hist[0] / wall:.1f} %  1: {100 * hist[1] / wall:.1f} %  >=2: {100 * hist[2] / wall:.1f} %")
alone, both = collections.defaultdict(list), collections.defaultdict(list)
for i, (s, e, n) in enumerate(ev):
    ov = any(s2 < e and e2 > s for j, (s2, e2, _) in enumerate(ev[max(0, i - 8):i + 8]) if j + max(0, i - 8) != i)
    (both if ov else alone)[n].append(e - s)
tot = collections.Counter()
for s, e, n in ev:
    tot[n] += e - s
for n, t in tot.most_common(14):
    a, b = alone[n], both[n]
    print(f"  {n:60s} total {t / 1e6:7.2f} ms  alone {len(a):4d} x {sum(a) / max(1, len(a)) / 1e3:8.1f} us   overlapped {len(b):4d} x {sum(b) / max(1, len(b)) / 1e3:8.1f} us")
