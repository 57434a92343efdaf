"""bench.py contract (driver-facing): one JSON line with the agreed keys, roofline and cpu_baseline objects, on a tiny run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_json_contract_transformer():
    j = _run('--steps', '3', '--warmup', '2', '--docs', '8', '--seq', '128', '--cpu-docs', '1')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in j, k
    assert j['n_gpus'] == 1 and j['steps'] == 3 and j['warmup'] == 2 and j['higher_is_better'] is True and j['scaling'] == 'weak'
    assert j['unit'] == 'sentences/s' and j['dtype'] == 'bf16' and j['data'] == 'synthetic' and j['vs_baseline'] is None
    assert 'workload' in j['config'] and 'model' not in j['config']
    assert abs(j['value'] - 8 * 128 * 1000.0 / j['ms_per_step']) < 1e-3 * j['value']
    r = j['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert 'traffic' in r
    assert j['extra']['sustained_steps'] == 1000 and j['extra']['sustained_value'] > 0
    assert j['config']['workload'].startswith('variant')            # 8 x 128 is not a BASELINE.json configuration
    c = j['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['cores'] >= 1 and c['value'] > 0 and c['unit'] == 'sentences/s' and 'sample' in c


@pytest.mark.parametrize('arch', ['bilstm', 'bilstm_crf', 'latefusion'])
def test_bench_runs_other_architectures(arch):
    j = _run('--arch', arch, '--steps', '2', '--warmup', '1', '--docs', '16', '--seq', '64', '--no-cpu-baseline')
    assert j['value'] > 0 and j['ms_per_step'] > 0
    assert 'configs[1]' not in j['config']['workload']


def test_bench_ragged_counts_valid_sentences_only():
    j = _run('--ragged', '--steps', '2', '--warmup', '1', '--docs', '8', '--seq', '128', '--no-cpu-baseline')
    assert j['value'] * j['ms_per_step'] / 1000.0 < 8 * 128


def test_default_line_witnesses_the_other_baseline_configurations():
    """the driver's BENCH record carries 50-step figures of configs[2] (focal and CRF heads), configs[4]'s per-GPU
    workload and the fp32 parity mode (transformer and BiLSTM) under extra.other_configs; `value` / `roofline` stay on configs[1]."""
    j = _run('--steps', '3', '--warmup', '2', '--cpu-docs', '1', '--sustained-steps', '0')
    assert j['config']['workload'].startswith('BASELINE configs[1]')
    oc = j['extra']['other_configs']
    assert len(oc) == 5 and sum('configs[2]' in k for k in oc) == 3 and sum('configs[4]' in k for k in oc) == 1 and sum('fp32' in k for k in oc) == 2
    for k, v in oc.items():
        assert v['steps'] == 50 and v['ms_per_step'] > 0 and v['final_loss'] == v['final_loss'], k
        seq = 512 if 'configs[4]' in k else 256
        assert abs(v['sentences_per_s'] - 64 * seq * 1e3 / v['ms_per_step']) < 1e-6 * v['sentences_per_s']
    assert j['roofline']['kernel'].startswith('gemm_bf16_224')
    # ... and the PCIe-inclusive legs, the product's own collater among them: documents -> collater -> prefetcher -> step
    h = j['extra']['h2d']
    assert {'collater, fp32 on the wire', 'collater, bf16 on the wire', 'pinned, fp32 on the wire', 'pageable, fp32 on the wire', 'pinned, bf16 on the wire'} <= set(h)
    for k in ('collater, fp32 on the wire', 'collater, bf16 on the wire'):
        assert h[k]['ms_per_step'] > 0 and h[k]['collater_ms_per_batch'] > 0 and h[k]['collate_threads'] >= 1, k


def test_bench_h2d_collater_line():
    j = _run('--h2d', 'collater', '--h2d-wire', 'bf16', '--steps', '6', '--warmup', '3', '--docs', '8', '--seq', '128', '--no-cpu-baseline', '--no-other-configs')
    assert j['h2d']['host_memory'].startswith("the collater's pinned ring") and j['h2d']['wire_dtype'].startswith('bf16')
    assert j['value'] > 0 and 'PCIe-inclusive' in j['config']['workload']


def _run_env(env, *flags):
    e = dict(os.environ)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *flags], capture_output=True, text=True, timeout=900, cwd=ROOT, env=e)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus N` with no WORLD_SIZE (the reference's entry is a plain Trainer(gpus=N), train_fit.py:284-296): the
    parent starts N ranks itself.  On a one-GPU box: N = 2 is refused with a clear message; with MTS_BENCH_REHEARSAL=1 both ranks
    share cuda:0 over gloo and rank 0 prints the one JSON line of a 2-rank run (sharding, hooks, max-over-ranks timing)."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    if torch.cuda.device_count() < 2:
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'], capture_output=True, text=True,
                             timeout=120, cwd=ROOT, env=env)
        assert out.returncode == 2 and 'GPU(s) are visible' in out.stderr and not out.stdout.strip()
    env['MTS_BENCH_REHEARSAL'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '2', '--docs', '8', '--seq', '128',
                          '--sustained-steps', '0', '--no-cpu-baseline'], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['config']['global_batch_docs'] == 16 and 'gloo' in j['config']['parallelism'] and 'RCCL' not in j['config']['parallelism']
    assert abs(j['value'] - 2 * 8 * 128 * 1000.0 / j['ms_per_step']) < 1e-3 * j['value']


@pytest.mark.parametrize('arch', ['transformer', 'latefusion'])
def test_single_rank_data_parallel_step_path_over_rccl(arch):
    """MTS_BENCH_SINGLE_RANK_DP=1: a ONE-rank RCCL group with the overlapped exchange path forced on -- the real collective launches
    (backend "nccl" = RCCL), hook order and stream waits of the N > 1 step on one GPU; the loss must be the plain run's."""
    flags = ['--arch', arch, '--steps', '3', '--warmup', '2', '--docs', '8', '--seq', '128', '--sustained-steps', '0', '--no-cpu-baseline']
    a = _run_env({'MTS_BENCH_SINGLE_RANK_DP': '1'}, *flags)
    assert a.returncode == 0, a.stderr[-3000:]
    ja = json.loads([ln for ln in a.stdout.splitlines() if ln.startswith('{')][0])
    jb = _run(*flags)
    assert 'one-rank RCCL group' in ja['config']['workload']
    # (not bitwise: under a hook the q/k/v weight gradient is three GEMMs with their own K splits, and bf16 weights follow)
    assert abs(ja['final_loss'] - jb['final_loss']) <= 2e-3 * max(1.0, abs(jb['final_loss'])), (ja['final_loss'], jb['final_loss'])
    # the reduce-scatter + all-gather schedule through RCCL itself (one rank: both collectives are copies, issued back to back on
    # RCCL's stream with no wait between them): the same loss as the all-reduce schedule, bit for bit
    c = _run_env({'MTS_BENCH_SINGLE_RANK_DP': '1', 'MTS_DP_SCHEDULE': 'rs_ag'}, *flags)
    assert c.returncode == 0, c.stderr[-3000:]
    jc = json.loads([ln for ln in c.stdout.splitlines() if ln.startswith('{')][0])
    assert 'reduce-scatter + all-gather' in jc['config']['parallelism'] and jc['final_loss'] == ja['final_loss'], (jc['final_loss'], ja['final_loss'])
