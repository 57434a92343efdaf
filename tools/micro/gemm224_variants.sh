#!/bin/bash
# Timing experiment on the 256x224 GEMM: a whole-library build whose K loop issues no copies (operands = whatever the LDS holds;
# results are wrong on purpose) -- what is left is the LDS-read -> MFMA chain, the barriers and the epilogue.
#   bash tools/micro/gemm224_variants.sh && gpurun -- python tools/gemm_ab_libs.py multimodaltopicsegmentation_amd/libmts_hip.so tools/ab_libs/libg224_NO_DMA.so
# (The copies-only counterpart quoted in DESIGN.md section 3 was a throwaway build of the end-of-tile-barrier schedule.)
set -e
cd "$(dirname "$0")/../.."
C=multimodaltopicsegmentation_amd/csrc
python multimodaltopicsegmentation_amd/build.py >/dev/null
OTHERS=$(ls $C/*.o | grep -v gemm224.o)
for v in NO_DMA "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DG224_DBG_$v -I$C -Iinclude -c $C/gemm224.hip -o /tmp/g224_$v.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab_libs/libg224_$v.so /tmp/g224_$v.o $OTHERS ) &
done
wait
ls -la tools/ab_libs/libg224_*.so
