"""profiles/r02_traffic.json from the two --pmc passes of tools/collect_traffic.sh:  python tools/traffic_to_json.py <outdir> <batch> [workload]
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE x2 on gfx950)."""
import csv, glob, json, sys, collections
out, batch = sys.argv[1], int(sys.argv[2])
workload = sys.argv[3] if len(sys.argv) > 3 else "dual256"
CLASS = {"attn_fwd_fast_kernel": "attn_fwd", "attn_bwd_dkdv16_w4_kernel": "attn_bwd_dkdv", "attn_bwd_dq2_w4": "attn_bwd_dq",
         "gemm_bf16_kernel<false, false": "gemm_nt", "gemm_bf16_kernel<false, true": "gemm_nn", "gemm_bf16_kernel<true, true": "gemm_tn",
         "gemm_bf16_group_kernel<true, true": "gemm_tn_group",
         "layernorm_bwd_kernel": "layernorm_bwd", "layernorm_fwd_kernel": "layernorm_fwd"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        for k, name in CLASS.items():
            if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals[name][c].append(float(r["Counter_Value"]))
res = {}
for name, v in vals.items():
    fs, ws = v["FETCH_SIZE"], v["WRITE_SIZE"]
    if fs and ws:
        f, w = sum(fs) / len(fs), sum(ws) / len(ws)
        res[name] = dict(batch=batch, workload=workload, launches=len(fs), fetch_size_kib=f, write_size_kib=w,
                         hbm_bytes_per_launch=(2 * f + w) * 1024, note="FETCH_SIZE doubled (gfx950), separate --pmc passes")
json.dump(res, open("profiles/r02_traffic.json", "w"), indent=1)
for k, e in res.items():
    print(f"{k:16s} fetch {e['fetch_size_kib']/1024:9.1f} MiB  write {e['write_size_kib']/1024:9.1f} MiB  -> hbm {e['hbm_bytes_per_launch']/1e6:9.1f} MB/launch")
