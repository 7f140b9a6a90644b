#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r4_attn3.log; : > $O
V=hsenet_amd/lib/variants/libhsenet_stamps.so
run() { echo "=== $*" | tee -a $O; timeout -k 10 200 "$@" >> $O 2>&1; rc=$?; echo "rc=$rc" | tee -a $O; return $rc; }
run env HS_ATTN_FUSED=2 python tools/dev/attn1_debug.py 2 4097 12 &&
run env HS_ATTN_FUSED=12 python tools/dev/attn1_debug.py 5 2049 12 &&
run env HS_ATTN_FUSED=2 python tools/dev/attn1_debug.py 7 513 12 &&
run env HS_ATTN_FUSED=2 HS_LIB=$V tools/kbench attnb 20 5 0 100 &&
run env HS_ATTN_FUSED=2 HS_LIB=$V HS_DBG4=8 tools/kbench attnb 20 5 0 100 &&
run env HS_ATTN_FUSED=2 tools/kbench attnb 53 3 0 100
grep -v "amdgpu.ids\|by wave\|by key block\|by head\|%% 32\|round [01]\|by key half\|by d half\|ratio got\|iter 4[1-6]" $O | tail -60
