#!/bin/bash
# round 4: weight-gradient (TN) kernels on 16x16x32 MFMAs (HS_GEMM_TN16=1, default) vs 32x32x16 (=0)
set -e
OUT=gpurun_out/r4_gemm5; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_gemm_sweep.py tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
: > $OUT/kbench.txt
for O in 0 1 0 1; do
  echo "== HS_GEMM_TN16=$O" | tee -a $OUT/kbench.txt
  for F in "wgrad fc1" "wgrad fc2" "wgrad qkv" "wgrad proj"; do
    HS_GEMM_TN16=$O timeout -k 10 120 tools/kbench gemm 53 10 "$F" >> $OUT/kbench.txt 2>&1
  done
  HS_GEMM_TN16=$O timeout -k 10 120 tools/kbench gemm 1 50 "wgrad" >> $OUT/kbench.txt 2>&1
done
grep -v amdgpu $OUT/kbench.txt
for O in 0 1 0 1; do
  echo "== bench HS_GEMM_TN16=$O" | tee -a $OUT/bench.txt
  HS_GEMM_TN16=$O timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-extras > $OUT/bench_o$O.json 2>$OUT/bench.err
  python3 - $OUT/bench_o$O.json <<'PY' | tee -a $OUT/bench.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels']:
    if k['name'] in ('gemm_tn_group','gemm_tn'): print(' ', k['name'], k['avg_us'], k['tflops'])
PY
done
