#!/bin/bash
# round-4 GEMM tile-order experiments: column panels (HS_GEMM_NP) x group height (HS_GEMM_GM) on the layer's N = 3072 / 2304 shapes,
# and the grouped weight-gradient split (HS_WGRAD_GROUP_SPLIT) in the bench itself
set -e
OUT=gpurun_out/r4_gemm1; mkdir -p $OUT
for NP in 0 6 4 3; do for GM in 0 8; do
  echo "== NP=$NP GM=$GM" | tee -a $OUT/kbench.txt
  for F in "fc1+gelu" "gelu' as NT" "qkv NT" "fc1 NT"; do
    HS_GEMM_NP=$NP HS_GEMM_GM=$GM timeout -k 10 120 tools/kbench gemm 53 20 "$F" >> $OUT/kbench.txt 2>&1
  done
done; done
cat $OUT/kbench.txt
for S in 0 8 16; do
  echo "== group split $S" | tee -a $OUT/bench.txt
  HS_WGRAD_GROUP_SPLIT=$S timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batch1 --no-extras > $OUT/bench_s$S.json 2>$OUT/bench_s$S.err
  python3 - $OUT/bench_s$S.json <<'PY' | tee -a $OUT/bench.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels'][:9]: print(k['name'], k['avg_us'], k['tflops'])
PY
done
