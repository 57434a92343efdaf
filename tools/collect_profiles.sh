#!/bin/bash
# Collect the round's measurement artifacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r03_v1
# writes everything under gpurun_out/<tag>/; tools/publish_profiles.py copies the summaries into profiles/.
# rocprofv3 gets the program itself after "--" (python3 bench.py ...), counters in their own passes (no trace domains with --pmc).
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-other-configs"
echo "[1] default bench line (with cpu_baseline and sustained leg)"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "[2] kernel trace + stats of the same command (no cpu baseline)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o tr -- python3 bench.py $B --sustained-steps 0 > $OUT/bench_profiled.json 2> $OUT/trace.err || exit 1
echo "[3] HBM-side traffic: FETCH_SIZE and WRITE_SIZE in separate passes"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 2 $B --no-kernel-timer --sustained-steps 0 > /dev/null 2> $OUT/pmc_fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --steps 4 --warmup 2 $B --no-kernel-timer --sustained-steps 0 > /dev/null 2> $OUT/pmc_write.err || exit 1
echo "[4] SQ counters (matrix-core busy, waits) of every kernel of the step"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq -o s -- python3 bench.py --steps 4 --warmup 2 $B --no-kernel-timer --sustained-steps 0 > /dev/null 2> $OUT/pmc_sq.err || exit 1
echo "[5] the other BASELINE configurations: bench line + kernel stats"
for a in bilstm bilstm_crf; do
  python3 bench.py --arch $a $B > $OUT/bench_$a.json 2> /dev/null || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$a -o tr -- python3 bench.py --arch $a $B --sustained-steps 0 > /dev/null 2> $OUT/trace_$a.err || exit 1
done
python3 bench.py --arch latefusion --seq 512 --steps 10 --warmup 3 $B --sustained-steps 200 > $OUT/bench_latefusion512.json 2> /dev/null || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_latefusion512 -o tr -- python3 bench.py --arch latefusion --seq 512 --steps 10 --warmup 3 $B --sustained-steps 0 > /dev/null 2> $OUT/trace_lf.err || exit 1
echo "[6] inference latency lines"
for a in transformer bilstm bilstm_crf latefusion; do python3 bench.py --infer --arch $a --docs 1 --seq 2437 --steps 50 --warmup 5 2> /dev/null; done > $OUT/infer_latency.jsonl
echo "[6b] the N > 1 step path on one GPU: one-rank RCCL group, overlapped exchange forced on"
MTS_BENCH_SINGLE_RANK_DP=1 python3 bench.py $B --sustained-steps 200 > $OUT/bench_dp1.json 2> /dev/null
MTS_BENCH_SINGLE_RANK_DP=1 python3 bench.py --arch latefusion --seq 512 --steps 10 --warmup 3 $B --sustained-steps 100 > $OUT/bench_dp1_latefusion512.json 2> /dev/null
MTS_BENCH_SINGLE_RANK_DP=1 MTS_DP_SCHEDULE=rs_ag python3 bench.py $B --sustained-steps 200 > $OUT/bench_dp1_rs_ag.json 2> /dev/null
MTS_BENCH_SINGLE_RANK_DP=1 MTS_DP_QKV_RELEASE=projection python3 bench.py $B --sustained-steps 200 > $OUT/bench_dp1_projection.json 2> /dev/null
echo "[6c] host batches in the loop (PCIe-inclusive variant lines)"
for m in "pinned fp32" "pageable fp32" "pinned bf16" "collater fp32" "collater bf16"; do set -- $m; python3 bench.py --h2d $1 --h2d-wire $2 $B 2> /dev/null; done > $OUT/h2d.jsonl
echo "[7] fp32 (parity) mode throughput"
python3 bench.py --dtype fp32 --steps 5 --warmup 2 $B --sustained-steps 0 > $OUT/bench_fp32.json 2> /dev/null
python3 bench.py --dtype fp32 --arch bilstm --steps 5 --warmup 2 $B --sustained-steps 0 > $OUT/bench_fp32_bilstm.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_fp32_bilstm -o tr -- python3 bench.py --dtype fp32 --arch bilstm --steps 5 --warmup 2 $B --sustained-steps 0 > /dev/null 2> $OUT/trace_fp32_bilstm.err
echo "[8] in-process A/B lines"
python3 tools/band_ab.py > $OUT/band_fused_ab.txt 2> /dev/null
MTS_B=16 MTS_L=1024 python3 tools/band_ab.py >> $OUT/band_fused_ab.txt 2> /dev/null
bash tools/step_ab.sh "band_fused_bwd=0" "band_fused_bwd=1" > $OUT/step_ab_band.txt 2> /dev/null
echo "[9] GEMMs against the vendor library and against the kernels they replace, one process each"
python3 tools/blas_compare.py > $OUT/gemm_vs_vendor.txt 2>&1
python3 tools/gemm_ab.py gemm_variant 0 9 6 > $OUT/gemm_ab.txt 2>&1
python3 tools/tn_sweep.py > $OUT/tn_sweep.txt 2>&1
echo done
