#!/bin/bash
# round 4: the 16x16x32 NT GEMMs as persistent blocks (HS_GEMM_PERSIST=1) vs one tile per block, kbench shapes at 53 volumes + the bench
set -e
OUT=gpurun_out/r4_gemm2; mkdir -p $OUT; : > $OUT/kbench.txt
for P in 0 1 0 1; do
  echo "== HS_GEMM_PERSIST=$P" | tee -a $OUT/kbench.txt
  for F in "qkv NT" "proj NT" "fc1 NT" "fc2 NT" "fc1+gelu" "fc2+resid" "proj+resid" "gelu' as NT"; do
    HS_GEMM_PERSIST=$P timeout -k 10 120 tools/kbench gemm 53 20 "$F" >> $OUT/kbench.txt 2>&1
  done
done
cat $OUT/kbench.txt
for P in 0 1; do
  echo "== bench HS_GEMM_PERSIST=$P" | tee -a $OUT/bench.txt
  HS_GEMM_PERSIST=$P timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-extras > $OUT/bench_p$P.json 2>$OUT/bench_p$P.err
  python3 - $OUT/bench_p$P.json <<'PY' | tee -a $OUT/bench.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels'][:9]: print(k['name'], k['avg_us'], k['tflops'])
PY
done
