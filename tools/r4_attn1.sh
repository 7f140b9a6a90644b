#!/bin/bash
# round-4 development run: one-sweep attention backward vs the two-sweep kernels (tools/kbench attnb; variant 100 = hs_attn_bwd2)
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r4_attn1.log; : > $O
run() { echo "=== $*" | tee -a $O; timeout -k 10 150 "$@" >> $O 2>&1; rc=$?; echo "rc=$rc" | tee -a $O; return $rc; }
run env HS_ATTN_FUSED=12 tools/kbench attnb 2 3 0 100 &&
run env HS_ATTN_FUSED=2 tools/kbench attnb 2 3 0 100 &&
run env HS_ATTN_FUSED=2 HS_N=2049 tools/kbench attnb 4 3 0 100 &&
run env HS_ATTN_FUSED=2 HS_N=257 tools/kbench attnb 8 3 0 100 &&
run env HS_ATTN_FUSED=2 HS_LIB=hsenet_amd/lib/variants/libhsenet_stamps.so tools/kbench attnb 20 5 0 100 &&
run env HS_ATTN_FUSED=12 HS_LIB=hsenet_amd/lib/variants/libhsenet_stamps.so tools/kbench attnb 20 5 0 100
grep -v "round [01]" $O | tail -70
