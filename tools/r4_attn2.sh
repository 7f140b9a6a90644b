#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r4_attn2.log; : > $O
V=hsenet_amd/lib/variants/libhsenet_stamps.so
run() { echo "=== $*" | tee -a $O; timeout -k 10 200 "$@" >> $O 2>&1; rc=$?; echo "rc=$rc" | tee -a $O; return $rc; }
run env HS_ATTN_FUSED=12 python tools/dev/attn1_debug.py 2 513 12 &&
run env HS_ATTN_FUSED=2 python tools/dev/attn1_debug.py 2 513 12 &&
run env HS_ATTN_FUSED=2 python tools/dev/attn1_debug.py 2 4097 12 &&
run env HS_ATTN_FUSED=2 python tools/dev/attn1_debug.py 3 2049 12 &&
run env HS_ATTN_FUSED=2 python tools/dev/attn1_debug.py 11 256 12 &&
run env HS_ATTN_FUSED=2 HS_LIB=$V tools/kbench attnb 20 5 0 100
grep -v "amdgpu.ids\|by wave\|by key block\|by head\|%% 32\|round [01]\|by key half\|by d half\|ratio got" $O | tail -60
