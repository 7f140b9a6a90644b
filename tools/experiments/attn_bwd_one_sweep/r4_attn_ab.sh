#!/bin/bash
# round 4: same-box A/B of the attention backward: the shipped two sweeps (variant 0) against the one-sweep kernel (variant 100:
# hs_attn_bwd2 with HS_ATTN_ONE_SWEEP) on N(0,1) operands, 20 and 53 volumes of 4097 tokens x 12 heads; output -> profiles/
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r04_attn_bwd_one_sweep_ab.txt; : > $O
for B in 20 53; do
  echo "=== kbench attnb $B 5 0 100   (variant 0 = two sweeps, 100 = one sweep)" | tee -a $O
  timeout -k 10 250 tools/kbench attnb $B 5 0 100 2>&1 | grep -v "amdgpu.ids" >> $O || exit 1
done
cat $O
