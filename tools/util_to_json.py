"""profiles/<tag>_pmc.json: per kernel class of one bench.py run, rocprof-reported HBM bytes / GB/s and MFMA utilisation.

    python tools/util_to_json.py <outdir> <batch> [workload] [tag] [bench-line.json]

Inputs: the three separate --pmc passes of tools/collect_traffic.sh (FETCH_SIZE, WRITE_SIZE) and tools/collect_util.sh (SQ/GRBM).
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     (KiB counters; FETCH_SIZE doubled on gfx950, MI355X_MICROARCH.md "HBM")
  fetch_bytes_per_launch / write_bytes_per_launch: the two terms of that sum, kept separately (round 4)
  avg_us               = mean kernel duration in the UTIL pass' own kernel trace (a profiled pass: clocks read ~3 % low)
  hbm_gbs              = hbm_bytes_per_launch / avg_us, frac_of_8TBs against the 8 TB/s spec peak
  mfma_busy_frac       = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8) * 1024)    busy cycles summed over the 1024 SIMDs of the chip /
                         (kernel cycles of one XCD x 1024): the share of matrix-pipe cycles in which an MFMA executes
  mfma_tflops_executed = SQ_INSTS_MFMA x FLOPs of the class' MFMA shape / avg_us      (what the matrix pipes executed, recompute included)
"""
import collections
import csv
import glob
import json
import sys

out, batch = sys.argv[1], int(sys.argv[2])
workload = sys.argv[3] if len(sys.argv) > 3 else "dual256"
tag = sys.argv[4] if len(sys.argv) > 4 else "r05"
# kernel-name substring -> class (= hs_prof_name() of the library's per-kernel timers); first match wins
CLASS = [("attn_fwd_fast_kernel", "attn_fwd"), ("attn_bwd_dkdv16_w4_kernel", "attn_bwd_dkdv"), ("attn_bwd_dq2_w4", "attn_bwd_dq"),
         ("gemm_nt16_persist_kernel<0,", "gemm_nt16_plain"), ("gemm_nt16_persist_kernel<9,", "gemm_nt16_bias_resid"), ("gemm_nt16_persist_kernel<1,", "gemm_nt16_bias"),
         ("gemm_nt16_persist_kernel<67,", "gemm_nt16_bias_gelu"), ("gemm_nt16_persist_kernel<68,", "gemm_nt16_dgelu"),
         ("gemm_bf16_kernel<false, false, true, 0, true", "gemm_nt16_plain"), ("gemm_bf16_kernel<false, false, true, 9, true", "gemm_nt16_bias_resid"),
         ("gemm_bf16_kernel<false, false, true, 1, true", "gemm_nt16_bias"),
         ("gemm_bf16_kernel<false, false, true, 67, true", "gemm_nt16_bias_gelu"), ("gemm_bf16_kernel<false, false, true, 68, true", "gemm_nt16_dgelu"),
         ("gemm_bf16_group_kernel<true, true", "gemm_tn_group"), ("gemm_bf16_kernel<false, false", "gemm_nt"), ("gemm_bf16_kernel<false, true", "gemm_nn"),
         ("gemm_bf16_kernel<true, true", "gemm_tn"), ("layernorm_bwd_kernel", "layernorm_bwd"), ("layernorm_fwd_kernel", "layernorm_fwd"),
         ("adamw_masked_kernel", "adamw"), ("adamw_fused_kernel", "adamw"), ("patchify_kernel", "patchify")]


def mfma_flops(name):
    """FLOPs of one MFMA instruction of this kernel: v_mfma_f32_32x32x16_bf16 = 32768; the kernels built on 16x16x32 (the dK/dV
    sweep; the NT GEMM's 256x256 tile = gemm_bf16_kernel<..., true> with M16 as the LAST template argument) = 16384."""
    if "dkdv16" in name or "gemm_nt16_persist_kernel" in name:
        return 2 * 16 * 16 * 32
    if "gemm_bf16_kernel<" in name:           # template arguments <ATR, BTR, BIG, EF, M16[, F16]>
        targs = [t.strip() for t in name.split("gemm_bf16_kernel<")[1].split(">")[0].split(",")]
        if len(targs) >= 5 and targs[4] == "true":
            return 2 * 16 * 16 * 32
    return 2 * 32 * 32 * 16


def klass(name):
    for k, c in CLASS:
        if k in name:
            return c
    return None


def counters(sub):
    f = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return vals
    for r in csv.DictReader(open(f[0])):
        c = klass(r["Kernel_Name"])
        if c:
            v = float(r["Counter_Value"])
            vals[c][r["Counter_Name"]].append(v)
            if r["Counter_Name"] == "SQ_INSTS_MFMA":
                vals[c]["_mfma_flops"].append(v * mfma_flops(r["Kernel_Name"]))
    return vals


def durations():
    f = glob.glob(f"{out}/UTIL/**/*kernel_trace.csv", recursive=True)
    d = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f[0])):
            c = klass(r["Kernel_Name"])
            if c:
                d[c].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    return d


fetch, write, util, dur = counters("FETCH_SIZE"), counters("WRITE_SIZE"), counters("UTIL"), durations()
mean = lambda x: sum(x) / len(x) if x else None
res = {}
for c in sorted(set(list(util) + list(fetch))):
    e = dict(batch=batch, workload=workload)
    fs, ws = mean(fetch[c]["FETCH_SIZE"]), mean(write[c]["WRITE_SIZE"])
    us = mean(dur[c])
    if fs is not None and ws is not None:
        e["hbm_bytes_per_launch"] = (2 * fs + ws) * 1024
        e["fetch_bytes_per_launch"], e["write_bytes_per_launch"] = 2 * fs * 1024, ws * 1024      # kept apart: a re-fetched operand panel
        #   shows up in the first, write-allocate reads behind a streamed output would show up in the first too but scale with the second
    if us:
        e["launches"], e["avg_us"] = len(dur[c]), us
        if "hbm_bytes_per_launch" in e:
            e["hbm_gbs"] = e["hbm_bytes_per_launch"] / us / 1e3
            e["hbm_frac_of_8TBs"] = e["hbm_gbs"] / 8000.0
    u = util[c]
    busy, gui, insts = mean(u["SQ_VALU_MFMA_BUSY_CYCLES"]), mean(u["GRBM_GUI_ACTIVE"]), mean(u["_mfma_flops"])
    if busy is not None and gui:
        e["mfma_busy_frac"] = busy / (gui / 8.0 * 1024.0)
        e["clock_ghz_from_grbm"] = gui / 8.0 / (us * 1e3) if us else None
    if insts is not None and us:
        e["mfma_tflops_executed"] = insts / us / 1e6
    res[c] = e
# tie the counters to the library they were taken with: bench.py reports them only while the loaded library's hs_build_id matches
try:
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from hsenet_amd import _lib as _L
    res["_build_id"] = _L.build_id()
except Exception as e:                     # noqa: BLE001
    res["_build_id"] = None
    print("no build id:", e)
if len(sys.argv) > 5:                      # the bench line measured beside the passes (volumes/s of the un-profiled run)
    try:
        res["_bench_value"] = json.load(open(sys.argv[5]))["value"]
    except Exception as e:                 # noqa: BLE001
        print("no bench value:", e)
res["_note"] = ("separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ+GRBM), FETCH_SIZE doubled (gfx950); durations from the "
                "SQ+GRBM pass' kernel trace; see tools/util_to_json.py for the formulas")
json.dump(res, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
for k, e in res.items():
    if k.startswith("_"):
        continue
    print(f"{k:16s} {e.get('avg_us', 0):9.1f} us  hbm {e.get('hbm_bytes_per_launch', 0) / 1e6:8.1f} MB (fetch {e.get('fetch_bytes_per_launch', 0) / 1e6:8.1f} "
          f"write {e.get('write_bytes_per_launch', 0) / 1e6:8.1f})  {e.get('hbm_gbs', 0):7.0f} GB/s  "
          f"mfma busy {100 * e.get('mfma_busy_frac', 0):5.1f} %  executed {e.get('mfma_tflops_executed', 0):7.0f} TF")
