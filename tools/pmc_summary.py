import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:34], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get("SQ_WAVE_CYCLES", 1)
    print(k, "n=%d" % len(next(iter(v.values()))))
    for c, val in sorted(m.items()):
        print("    %-28s %.3e  (%.1f%% of WAVE_CYCLES)" % (c, val, 100 * val / wc))
