"""Build libmts_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by hand:

    python multimodaltopicsegmentation_amd/build.py [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmts_hip.so')
SOURCES = ['gemm.hip', 'gemm256.hip', 'gemm224.hip', 'gemm224r.hip', 'gemm224t.hip', 'gemm224p.hip', 'gemm224n.hip', 'ffn_fused.hip', 'norm.hip', 'band_attn.hip', 'band_attn_mfma.hip', 'loss.hip', 'dropout.hip', 'optim.hip', 'lstm.hip', 'lstm_mfma.hip', 'lstm_pair.hip', 'crf.hip', 'collate.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(src):
    obj = os.path.join(CSRC, src.replace('.hip', '.o'))
    # every header of csrc/ is a dependency of every source (cheap, and no stale object when a shared header changes)
    deps = [os.path.join(CSRC, src), os.path.join(HERE, '..', 'include', 'mts.h')]
    deps += [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith('.h')]
    if any(_newer(d, obj) for d in deps):
        cmd = ['hipcc', *FLAGS, '-c', os.path.join(CSRC, src), '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if force:
        for s in srcs:
            o = os.path.join(CSRC, s.replace('.hip', '.o'))
            if os.path.exists(o):
                os.remove(o)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, srcs))
    if any(_newer(o, LIB) for o in objs):
        cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
