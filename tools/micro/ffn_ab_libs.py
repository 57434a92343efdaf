#!/usr/bin/env python3
"""Times mts_ffn_fwd / mts_ffn_bwd_data of several library builds in one process (see ffn_variants.sh):
    python tools/micro/ffn_ab_libs.py lib1.so lib2.so ... [--M 16384]"""
import ctypes as C
import sys

import torch

vp, i32 = C.c_void_p, C.c_int
paths = [a for a in sys.argv[1:] if a.endswith('.so')]
M = int(sys.argv[sys.argv.index('--M') + 1]) if '--M' in sys.argv else 16384
D, F, dev = 1792, 256, 'cuda'
Mp = (M + 63) // 64 * 64
bf = dict(dtype=torch.bfloat16, device=dev)
a1, ds2 = torch.randn(M, D, **bf), torch.randn(M, D, **bf)
w1, w2 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16), (torch.randn(D, F, device=dev) * F ** -0.5).to(torch.bfloat16)
b1, b2 = torch.randn(F, device=dev), torch.randn(D, device=dev)
u, f, du = (torch.zeros(Mp, F, **bf) for _ in range(3))
s2, da1 = (torch.zeros(Mp, D, **bf) for _ in range(2))
st = torch.cuda.current_stream().cuda_stream
runs = []
for p in paths:
    lib = C.CDLL(p)
    lib.mts_ffn_fwd.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp]
    lib.mts_ffn_bwd_data.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]
    name = p.split('/')[-1]
    runs.append((name + ' fwd', lambda lib=lib: lib.mts_ffn_fwd(st, M, D, F, a1.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 0,
                                                                 u.data_ptr(), f.data_ptr(), s2.data_ptr())))
    runs.append((name + ' bwd', lambda lib=lib: lib.mts_ffn_bwd_data(st, M, D, F, ds2.data_ptr(), w1.data_ptr(), w2.data_ptr(), u.data_ptr(), 0,
                                                                      du.data_ptr(), da1.data_ptr())))


def timed(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        assert fn() == 0
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


best = {n: 1e30 for n, _ in runs}
for rep in range(4):
    for n, fn in runs:
        fn()
        best[n] = min(best[n], timed(fn, 50))
for n, _ in runs:
    print('M=%d  %-34s %7.1f us' % (M, n, best[n]), flush=True)
