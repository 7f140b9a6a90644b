#!/bin/bash
# the tile order's group height (HS_GEMM_GM) on the weight-gradient (TN) shapes: bash tools/gm_sweep_tn.sh <batch> <gm values...>
B=$1; shift
for GM in "$@"; do
  echo "== GM=$GM"
  HS_GEMM_GM=$GM tools/kbench gemm $B 10 2>&1 | grep TFLOP | grep "wgrad"
done
