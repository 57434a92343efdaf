#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/ under the round's names:
    python tools/publish_profiles.py r02_v1"""
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def one(pattern):
    m = glob.glob(os.path.join(src, pattern), recursive=True)
    assert m, pattern
    return m[0]


def cp(a, b):
    shutil.copy(a, os.path.join(dst, b))
    print('profiles/' + b)


cp(os.path.join(src, 'bench.json'), f'{tag}_bench.json')
cp(os.path.join(src, 'bench_profiled.json'), f'{tag}_bench_profiled.json')
cp(one('trace/**/*kernel_stats.csv'), f'{tag}_bench_kernel_stats.csv')
for a in ('bilstm', 'bilstm_crf', 'latefusion512'):
    cp(os.path.join(src, f'bench_{a}.json'), f'{tag}_{a}_bench.json')
    cp(one(f'trace_{a}/**/*kernel_stats.csv'), f'{tag}_{a}_kernel_stats.csv')
cp(os.path.join(src, 'infer_latency.jsonl'), f'{tag}_infer_latency.jsonl')
cp(os.path.join(src, 'bench_fp32.json'), f'{tag}_fp32_bench.json')
cp(os.path.join(src, 'bench_fp32_bilstm.json'), f'{tag}_fp32_bilstm_bench.json')
if glob.glob(os.path.join(src, 'trace_fp32_bilstm/**/*kernel_stats.csv'), recursive=True):
    cp(one('trace_fp32_bilstm/**/*kernel_stats.csv'), f'{tag}_fp32_bilstm_kernel_stats.csv')
for f, n in (('bench_dp1.json', 'dp1_bench.json'), ('bench_dp1_latefusion512.json', 'dp1_latefusion512_bench.json'), ('h2d.jsonl', 'h2d.jsonl'),
             ('bench_dp1_rs_ag.json', 'dp1_rs_ag_bench.json'), ('bench_dp1_projection.json', 'dp1_projection_bench.json'),
             ('band_fused_ab.txt', 'band_fused_ab.txt'), ('step_ab_band.txt', 'step_ab_band.txt'),
             ('gemm_vs_vendor.txt', 'gemm_vs_vendor.txt'), ('gemm_ab.txt', 'gemm_ab.txt'), ('tn_sweep.txt', 'tn_sweep.txt')):
    if os.path.exists(os.path.join(src, f)):
        cp(os.path.join(src, f), f'{tag}_{n}')
# HBM-side traffic per kernel (stamped with the kernel-source hash; bench.py reads r04_pmc_traffic.json)
out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_traffic.py'), one('pmc_fetch/**/*counter_collection.csv'),
                      one('pmc_write/**/*counter_collection.csv'), os.path.join(dst, 'r04_pmc_traffic.json')], capture_output=True, text=True, check=True)
open(os.path.join(dst, f'{tag}_pmc_traffic.txt'), 'w').write(out.stdout)
print(f'profiles/r04_pmc_traffic.json, profiles/{tag}_pmc_traffic.txt')
out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'pmc_summary.py'), one('pmc_sq/**/*counter_collection.csv')], capture_output=True, text=True, check=True)
open(os.path.join(dst, f'{tag}_pmc_sq_counters.txt'), 'w').write(out.stdout)
print(f'profiles/{tag}_pmc_sq_counters.txt')
