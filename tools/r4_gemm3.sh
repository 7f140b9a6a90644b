#!/bin/bash
set -e
OUT=gpurun_out/r4_gemm3; mkdir -p $OUT
HS_GEMM_PERSIST=1 timeout -k 10 500 python3 -m pytest tests/test_gpu_gemm_sweep.py tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
HS_GEMM_PERSIST=1 timeout -k 10 300 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_grad_goldens.py -x -q -m gpu 2>&1 | tail -3
bash tools/r4_gemm2.sh > $OUT/log.txt 2>&1; tail -52 $OUT/log.txt
