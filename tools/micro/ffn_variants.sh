#!/bin/bash
# Timing builds of the fused feed-forward kernel: small libraries holding only csrc/ffn_fused.hip -- the production kernel (base)
# and the same with per-wave cycle sums (-DFFN_STAMPS: copy waves and compute waves, per step of both phases).
#   bash tools/micro/ffn_variants.sh && gpurun -- python tools/micro/ffn_stamps.py tools/ab_libs/libffn_STAMPS.so
#                                      gpurun -- python tools/micro/ffn_ab_libs.py tools/ab_libs/libffn_*.so
# (The first version of the kernel also had builds with the copies / the LDS reads / the MFMAs taken out:
#  profiles/r02_ffn_ingredients_first_version.txt.)
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
for v in base STAMPS "$@"; do
  def=""; [ "$v" != base ] && def="-DFFN_DBG_$v"; [ "$v" = STAMPS ] && def="-DFFN_STAMPS"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $def -I$C -Iinclude -shared -o tools/ab_libs/libffn_$v.so $C/ffn_fused.hip /tmp/ffn_stub.cpp &
done
wait
ls -la tools/ab_libs/libffn_*.so
