#!/usr/bin/env python3
"""In-process A/B of the fused feed-forward block (mts_ffn_fwd / mts_ffn_bwd_data) against the launches it replaces, at
BASELINE's d = 1792, ff = 256:  python tools/ffn_ab.py [M ...] [--rotate]   (default M = 16384).  Best of 5 x 50 back-to-back, and
500 sustained.  --rotate: every launch works on another of 8 buffer sets (1 GB in total: nothing is served from the Infinity Cache,
as inside the training step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import ops  # noqa: E402
from multimodaltopicsegmentation_amd.flat import round_up  # noqa: E402

D, F, dev = 1792, 256, 'cuda'
bf = dict(dtype=torch.bfloat16, device=dev)
ROT = '--rotate' in sys.argv
for M in [int(a) for a in sys.argv[1:] if not a.startswith('--')] or [16384]:
    Mp = round_up(M, 64)
    NSET = 8 if ROT else 1
    sets = [dict(a1=torch.randn(M, D, **bf), ds2=torch.randn(M, D, **bf), u=torch.empty(Mp, F, **bf)[:M], f=torch.empty(Mp, F, **bf)[:M],
                 du=torch.empty(Mp, F, **bf)[:M], s2=torch.empty(Mp, D, **bf)[:M], da1=torch.empty(Mp, D, **bf)[:M]) for _ in range(NSET)]
    state = {'i': 0}

    def nxt():
        state['i'] = (state['i'] + 1) % NSET
        return sets[state['i']]
    a1, ds2 = sets[0]['a1'], sets[0]['ds2']
    w1, w2 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16), (torch.randn(D, F, device=dev) * F ** -0.5).to(torch.bfloat16)
    b1, b2 = torch.randn(F, device=dev), torch.randn(D, device=dev)
    def fwd2():
        S = nxt()
        ops.linear_fwd(S['a1'], w1, b1, S['f'], gelu=True, aux=S['u'])
        ops.linear_fwd(S['f'], w2, b2, S['s2'], residual=S['a1'])

    def bwd2():
        S = nxt()
        ops.linear_dgrad(S['ds2'], w2, S['du'])
        ops.gelu_bwd(S['u'], S['du'])
        ops.linear_dgrad(S['du'], w1, S['da1'], residual=S['ds2'])

    def fwd1():
        S = nxt()
        ops.ffn_fwd(S['a1'], w1, b1, w2, b2, S['u'], S['f'], S['s2'])

    def bwd1():
        S = nxt()
        ops.ffn_bwd_data(S['ds2'], w1, w2, S['u'], S['du'], S['da1'])

    runs = [('fwd  two launches', fwd2), ('fwd  fused', fwd1), ('bwd  three launches', bwd2), ('bwd  fused', bwd1)]

    def timed(fn, n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / n
    for _ in range(NSET):
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
