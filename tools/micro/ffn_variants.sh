#!/bin/bash
# Timing experiments on the fused feed-forward kernel: small libraries holding only csrc/ffn_fused.hip, one per FFN_DBG_* macro
# (results of the DBG variants are wrong on purpose -- they take one ingredient out of the step to show what the step waits for).
#   bash tools/micro/ffn_variants.sh && gpurun -- python tools/micro/ffn_ab_libs.py tools/ab_libs/libffn_*.so
set -e
cd "$(dirname "$0")/../.."
C=multimodaltopicsegmentation_amd/csrc
cat > /tmp/ffn_stub.cpp <<'EOT'
#include <cstdarg>
#include <cstdio>
static char g_err[512];
void mts_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" const char* mts_last_error() { return g_err; }
EOT
for v in base NO_MMA NO_DMA NO_LDSREAD STAMPS "$@"; do
  def=""; [ "$v" != base ] && def="-DFFN_DBG_$v"; [ "$v" = STAMPS ] && def="-DFFN_STAMPS"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $def -I$C -Iinclude -shared -o tools/ab_libs/libffn_$v.so $C/ffn_fused.hip /tmp/ffn_stub.cpp &
done
wait
ls -la tools/ab_libs/libffn_*.so
