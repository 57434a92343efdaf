#!/bin/bash
# Diagnostic builds of the whole library with per-phase cycle sums in the 256x224 K loop (-DMTS_GEMM_STAMPS -DG224_PHASES), with
# and without copies:  bash tools/micro/gemm224_phases.sh && gpurun -- python tools/micro/gemm224_phases.py tools/ab_libs/libg224_PH*.so
set -e
cd "$(dirname "$0")/../.."
C=multimodaltopicsegmentation_amd/csrc
D=/tmp/g224_diag; mkdir -p $D
for f in $C/*.hip; do
  b=$(basename $f .hip)
  [ $b = gemm224 ] && continue
  [ $D/$b.o -nt $f ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMTS_GEMM_STAMPS -I$C -Iinclude -c $f -o $D/$b.o &
done
wait
for v in PH PH_NO_DMA; do
  def="-DG224_PHASES"; [ $v = PH_NO_DMA ] && def="$def -DG224_DBG_NO_DMA"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DMTS_GEMM_STAMPS $def -I$C -Iinclude -c $C/gemm224.hip -o $D/gemm224_$v.o && \
    hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab_libs/libg224_$v.so $D/gemm224_$v.o $(ls $D/*.o | grep -v gemm224_) ) &
done
wait
ls -la tools/ab_libs/libg224_PH*.so
