#!/usr/bin/env python3
"""A/B of one mts_set_option key on the big GEMM shapes inside one process (run-to-run and box-to-box noise is ~5 %).
    python tools/gemm_ab.py gemm_order 0 1"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

key, vals = sys.argv[1].encode(), [int(v) for v in sys.argv[2:]]
SHAPES = [('NT', 16384, 5376, 1792), ('NT', 16384, 1792, 1792), ('NN', 16384, 1792, 5376), ('NN', 16384, 1792, 1792),
          ('TN', 5376, 1792, 16384), ('TN', 1792, 1792, 16384), ('TT', 5376, 1792, 16384), ('TT', 1792, 1792, 16384)]
dev = 'cuda'
for lay, M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    if lay == 'NT':
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    elif lay == 'NN':
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
    elif lay == 'TT':
        A, B = torch.randn(K, M, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    else:
        A, B = torch.randn(K, M, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
    A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if lay in ('TN', 'TT') else torch.bfloat16, device=dev)
    code = {'NT': L.NT, 'NN': L.NN, 'TN': L.TN, 'TT': L.TT}[lay]
    res = {v: [] for v in vals}
    for rep in range(3):
        for v in vals:
            L.check(L.lib.mts_set_option(key, v))
            for _ in range(2):
                ops.gemm(code, A, B, out, M=M, N=N, K=K)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.gemm(code, A, B, out, M=M, N=N, K=K)
            e.record()
            torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) * 1e3 / 20)
    print(f'{lay} M={M:6d} N={N:5d} K={K:6d} ' + '  '.join(f'{key.decode()}={v}: {min(res[v]):7.1f} us ({2.0 * M * N * K / min(res[v]) / 1e6:6.1f} TF/s)' for v in vals), flush=True)
