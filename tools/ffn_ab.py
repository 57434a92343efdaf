#!/usr/bin/env python3
"""In-process A/B of the fused feed-forward block (mts_ffn_fwd / mts_ffn_bwd_data) against the launches it replaces, at
BASELINE's d = 1792, ff = 256:  python tools/ffn_ab.py [M ...]   (default M = 16384).  Best of 5 x 50 back-to-back, and 500 sustained."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import ops  # noqa: E402
from multimodaltopicsegmentation_amd.flat import round_up  # noqa: E402

D, F, dev = 1792, 256, 'cuda'
bf = dict(dtype=torch.bfloat16, device=dev)
for M in [int(a) for a in sys.argv[1:]] or [16384]:
    Mp = round_up(M, 64)
    a1, ds2 = torch.randn(M, D, **bf), torch.randn(M, D, **bf)
    w1, w2 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16), (torch.randn(D, F, device=dev) * F ** -0.5).to(torch.bfloat16)
    b1, b2 = torch.randn(F, device=dev), torch.randn(D, device=dev)
    u, f, du = (torch.empty(Mp, F, **bf)[:M] for _ in range(3))
    s2, da1 = (torch.empty(Mp, D, **bf)[:M] for _ in range(2))

    def fwd2():
        ops.linear_fwd(a1, w1, b1, f, gelu=True, aux=u)
        ops.linear_fwd(f, w2, b2, s2, residual=a1)

    def bwd2():
        ops.linear_dgrad(ds2, w2, du)
        ops.gelu_bwd(u, du)
        ops.linear_dgrad(du, w1, da1, residual=ds2)

    runs = [('fwd  two launches', fwd2), ('fwd  fused', lambda: ops.ffn_fwd(a1, w1, b1, w2, b2, u, f, s2)),
            ('bwd  three launches', bwd2), ('bwd  fused', lambda: ops.ffn_bwd_data(ds2, w1, w2, u, du, da1))]

    def timed(fn, n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / n
    fwd2()
    best = {n: 1e30 for n, _ in runs}
    for rep in range(5):
        for n, fn in runs:
            fn()
            best[n] = min(best[n], timed(fn, 50))
    flops = 2 * 2.0 * M * D * F
    for n, fn in runs:
        sus = timed(fn, 500)
        print('M=%6d  %-20s best %7.1f us (%.2f PFLOP/s)   sustained %7.1f us' % (M, n, best[n], flops / best[n] / 1e9, sus), flush=True)
