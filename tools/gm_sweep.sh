#!/bin/bash
# sweep the GEMM tile order's group height (HS_GEMM_GM) over the layer's shapes: bash tools/gm_sweep.sh <batch> <gm values...>
B=$1; shift
for GM in "$@"; do
  echo "== GM=$GM"
  HS_GEMM_GM=$GM tools/kbench gemm $B 10 2>&1 | grep TFLOP | grep -v "sq NT\|wgrad"
done
