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
