#!/bin/bash
# In-step A/B of mts_set_option switches on ONE box: bash tools/step_ab.sh "gemm_big_min_k=512" "" ...   (each argument = one MTS_OPTIONS string)
for rep in 1 2; do
  for opt in "$@"; do
    MTS_OPTIONS="$opt" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-steps 500 --no-other-configs 2>/dev/null | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('[$opt]', round(j['ms_per_step'],4), 'sustained', round(j['extra']['sustained_ms_per_step'],4))"
  done
done
