#!/bin/bash
# round-4 measurement batch 1 (gpurun, from the repo root)
set -o pipefail
OUT=gpurun_out/r4c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-other-configs"
echo "[t] kernels + prefetch + abi-level GPU tests"
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_prefetch.py tests/test_gpu_bench.py -x -q -m gpu > $OUT/t1.log 2>&1; echo "rc=$?"; tail -3 $OUT/t1.log
echo "[1] bench line (no cpu baseline; other configs + h2d legs incl. collater)"
python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
echo "[2] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 bench.py $B --sustained-steps 0 > $OUT/bench_profiled.json 2> $OUT/trace.err; echo "rc=$?"
echo "[3] single-rank DP"
MTS_BENCH_SINGLE_RANK_DP=1 python3 bench.py $B --sustained-steps 200 > $OUT/bench_dp1.json 2> /dev/null
MTS_BENCH_SINGLE_RANK_DP=1 MTS_DP_QKV_RELEASE=block python3 bench.py $B --sustained-steps 200 > $OUT/bench_dp1_block.json 2> /dev/null
MTS_BENCH_SINGLE_RANK_DP=1 MTS_DP_SCHEDULE=rs_ag python3 bench.py $B --sustained-steps 200 > $OUT/bench_dp1_rs_ag.json 2> /dev/null
echo "[4] vendor compare"
python3 tools/blas_compare.py > $OUT/gemm_vs_vendor.txt 2>&1
echo "[5] bilstm"
python3 bench.py --arch bilstm $B > $OUT/bench_bilstm.json 2> /dev/null
echo done
