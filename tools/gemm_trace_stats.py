"""Summarise a tools/kbench HS_GEMM_DBG=22 block timeline (per-block s_memtime stamps; counters are only comparable on one CU)."""
import csv, sys
import numpy as np
for name in sys.argv[1:]:
    rows = list(csv.DictReader(open(name)))
    st = np.array([int(r['start']) for r in rows], dtype=np.int64)
    me = np.array([int(r['main_end']) for r in rows], dtype=np.int64)
    ee = np.array([int(r['epi_end']) for r in rows], dtype=np.int64)
    hw = np.array([int(r['hw_id']) for r in rows]); xcc = np.array([int(r['xcc_id']) for r in rows]) & 0xf
    key = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 0xf)
    gaps, periods = [], []
    for k in np.unique(key):
        i = np.where(key == k)[0]; o = i[np.argsort(st[i])]
        gaps += list(st[o][1:] - ee[o][:-1]); periods += list(st[o][1:] - st[o][:-1])
    gaps = np.array(gaps); periods = np.array(periods)
    ok = (gaps > 0) & (gaps < 100000)
    print(f"{name}: {len(st)} blocks on {len(np.unique(key))} CUs | main loop {np.median(me-st):.0f} (p10 {np.percentile(me-st,10):.0f} p90 {np.percentile(me-st,90):.0f})"
          f" | epilogue {np.median(ee-me):.0f} (p10 {np.percentile(ee-me,10):.0f} p90 {np.percentile(ee-me,90):.0f}) | gap to next block {np.median(gaps[ok]):.0f} | period {np.median(periods[ok]):.0f}")
