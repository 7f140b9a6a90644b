#!/bin/bash
# round 4: the 256x128 tile's hand-pipelined main loop (small_mainloop) vs the previous ring loop (HS_GEMM_OLD_SMALL=1)
set -e
OUT=gpurun_out/r4_gemm4; mkdir -p $OUT
timeout -k 10 500 python3 -m pytest tests/test_gpu_gemm_sweep.py tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
: > $OUT/kbench.txt
for O in 1 0 1 0; do
  echo "== HS_GEMM_OLD_SMALL=$O  (one volume)" | tee -a $OUT/kbench.txt
  for F in "proj NT" "fc2 NT" "dgrad qkv as NT" "dgrad proj NN" "wgrad fc1" "wgrad proj"; do
    HS_GEMM_OLD_SMALL=$O timeout -k 10 120 tools/kbench gemm 1 50 "$F" >> $OUT/kbench.txt 2>&1
  done
  echo "== HS_GEMM_OLD_SMALL=$O  (53 volumes)" | tee -a $OUT/kbench.txt
  for F in "wgrad fc1" "wgrad fc2" "wgrad qkv" "wgrad proj"; do
    HS_GEMM_OLD_SMALL=$O timeout -k 10 120 tools/kbench gemm 53 10 "$F" >> $OUT/kbench.txt 2>&1
  done
done
grep -v amdgpu $OUT/kbench.txt
for B in 53 1; do for O in 1 0; do
  echo "== bench batch $B HS_GEMM_OLD_SMALL=$O" | tee -a $OUT/bench.txt
  HS_GEMM_OLD_SMALL=$O timeout -k 10 300 python3 bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-batch1 --no-extras > $OUT/bench_b${B}_o$O.json 2>$OUT/bench.err
  python3 - $OUT/bench_b${B}_o$O.json <<'PY' | tee -a $OUT/bench.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])
for k in d['kernels']:
    if k['name'] in ('gemm_tn_group','gemm_nt','gemm_tn','gemm_nn'): print(' ', k['name'], k['avg_us'], k['tflops'])
PY
done; done
