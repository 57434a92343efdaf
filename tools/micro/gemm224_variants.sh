#!/bin/bash
# Timing experiments on the 256x224 GEMM: whole-library builds with one ingredient of the K loop taken out (results are wrong on
# purpose).  NO_DMA: no copies (operands = whatever the LDS holds); NO_MMA: copies, waits and barriers only.
#   bash tools/micro/gemm224_variants.sh && gpurun -- python tools/gemm_ab_libs.py multimodaltopicsegmentation_amd/libmts_hip.so tools/ab_libs/libg224_NO_DMA.so tools/ab_libs/libg224_NO_MMA.so
set -e
cd "$(dirname "$0")/../.."
C=multimodaltopicsegmentation_amd/csrc
python multimodaltopicsegmentation_amd/build.py >/dev/null
OTHERS=$(ls $C/*.o | grep -v gemm224.o)
for v in NO_DMA NO_MMA "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DG224_DBG_$v -I$C -Iinclude -c $C/gemm224.hip -o /tmp/g224_$v.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab_libs/libg224_$v.so /tmp/g224_$v.o $OTHERS ) &
done
wait
ls -la tools/ab_libs/libg224_*.so
