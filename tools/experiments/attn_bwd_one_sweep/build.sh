#!/bin/bash
# One-sweep attention backward (retired experiment): compile attn_bwd1.hip and link it with the PRODUCT's objects into its own library
#   bash tools/experiments/attn_bwd_one_sweep/build.sh   ->  tools/experiments/attn_bwd_one_sweep/libhsenet_one_sweep.so
# (python -m hsenet_amd.build first; the product library itself does not contain this kernel)
set -e
cd "$(dirname "$0")/../../.."
D=tools/experiments/attn_bwd_one_sweep; L=hsenet_amd/lib
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize ${HS_HOOKS:+-DHS_DEBUG_HOOKS} -c $D/attn_bwd1.hip -o $D/attn_bwd1.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libhsenet_one_sweep.so $D/attn_bwd1.o $L/gemm.o $L/attn.o $L/norm.o $L/small.o $L/vit.o $L/path.o $L/prof.o $L/prep.o $L/build_id.o
echo built $D/libhsenet_one_sweep.so
