import csv, collections, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    if 'gemm' not in n and 'attn' not in n: continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    by[(n[:44], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])].append(d)
for k, v in sorted(by.items()):
    v = sorted(v)
    print(k, len(v), 'med %.1f min %.1f max %.1f' % (v[len(v)//2], v[0], v[-1]))
