#!/bin/bash
# ablation / A-B builds of ONE source with extra flags, linked against the product's other objects:
#   bash tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>   ->  hsenet_amd/lib/variants/libhsenet_<name>.so
# (run tools/kbench with HS_LIB=<that .so>; never shipped: *.so is git-ignored and the product loads hsenet_amd/lib/libhsenet_hip.so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift; shift
L=hsenet_amd/lib; mkdir -p $L/variants
EXTRA=""
[ "$SRC" = attn.hip ] && EXTRA="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
# -DHS_DEBUG_HOOKS: the diagnostic kernel-variant selectors (hs_debug_set keys, s_memtime stamps) exist only in these builds
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DHS_DEBUG_HOOKS $EXTRA "$@" -c hsenet_amd/csrc/$SRC -o $L/variants/${NAME}_${SRC%.hip}.o
if [ "$SRC" != prof.hip ]; then   # hs_debug_set itself lives in prof.hip: it must accept the keys too
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DHS_DEBUG_HOOKS -c hsenet_amd/csrc/prof.hip -o $L/variants/${NAME}_prof.o
fi
OBJS=""
for o in gemm attn norm small vit path prof prep; do
  if [ "$o.hip" = "$SRC" ] || [ "$o" = prof ]; then OBJS="$OBJS $L/variants/${NAME}_$o.o"; else OBJS="$OBJS $L/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $L/variants/libhsenet_$NAME.so $OBJS
echo built $L/variants/libhsenet_$NAME.so
