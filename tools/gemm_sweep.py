#!/usr/bin/env python3
"""Sweep tile size x split-K for the weight-gradient shapes (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops
dev = 'cuda'
def run(lay, M, N, K, reps=10):
    g = torch.Generator(device=dev).manual_seed(1)
    if lay == 'TN':
        A, B = torch.randn(K, M, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
        out = torch.empty(M, N, dtype=torch.float32, device=dev)
    elif lay == 'NT':
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    else:
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
    code = {'NT': L.NT, 'NN': L.NN, 'TN': L.TN}[lay]
    for _ in range(2):
        ops.gemm(code, A, B, out, M=M, N=N, K=K)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm(code, A, B, out, M=M, N=N, K=K)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for (M, N, K) in [(5376, 1792, 16384), (1792, 1792, 16384), (1792, 256, 16384)]:
    for tile in (128, 256):
        L.lib.mts_set_option(b'gemm_tile', tile)
        row = []
        for sp in (1, 2, 3, 4, 5, 6, 7, 8, 11, 16):
            L.lib.mts_set_option(b'gemm_splits', sp)
            us = run('TN', M, N, K)
            row.append(f'{sp}:{us:.0f}')
        print(f'TN {M}x{N}x{K} tile{tile}: ' + ' '.join(row), flush=True)
L.lib.mts_set_option(b'gemm_splits', 0)
for lay, M, N, K in [('NT', 16384, 5376, 1792), ('NT', 16384, 1792, 1792), ('NT', 16384, 256, 1792), ('NT', 16384, 1792, 256),
                     ('NN', 16384, 1792, 5376), ('NN', 16384, 1792, 1792), ('NN', 16384, 256, 1792), ('NN', 16384, 1792, 256)]:
    r = []
    for tile in (128, 256):
        L.lib.mts_set_option(b'gemm_tile', tile)
        r.append(f'tile{tile}:{run(lay, M, N, K):.0f}us')
    print(lay, M, N, K, ' '.join(r), flush=True)
