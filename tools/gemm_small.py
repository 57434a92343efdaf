#!/usr/bin/env python3
"""Option sweep of mts_gemm on the skinny FFN shapes of the BASELINE step (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

SHAPES = [('NT', 16384, 256, 1792), ('NT', 16384, 1792, 256), ('NN', 16384, 256, 1792), ('NN', 16384, 1792, 256),
          ('TN', 1792, 256, 16384), ('TN', 256, 1792, 16384)]
dev = 'cuda'
for lay, M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    if lay == 'NT':
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
    elif lay == 'NN':
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
    else:
        A, B = torch.randn(K, M, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
    A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if lay == 'TN' else torch.bfloat16, device=dev)
    code = {'NT': L.NT, 'NN': L.NN, 'TN': L.TN}[lay]
    for tile, glds, splits in [(0, 1, 0), (128, 1, 1), (128, 0, 1), (128, 1, 2), (128, 1, 4), (256, 1, 1), (256, 1, 2), (256, 1, 4)]:
        if splits > 1 and lay != 'TN':
            continue
        L.lib.mts_set_option(b'gemm_tile', tile); L.lib.mts_set_option(b'gemm_glds', glds); L.lib.mts_set_option(b'gemm_splits', splits)
        try:
            for _ in range(3):
                ops.gemm(code, A, B, out, M=M, N=N, K=K)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ops.gemm(code, A, B, out, M=M, N=N, K=K)
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 1e3 / 20
            print(f'{lay} M={M:6d} N={N:5d} K={K:6d} tile={tile:3d} glds={glds} splits={splits}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s', flush=True)
        except Exception as ex:
            print(lay, M, N, K, tile, glds, splits, 'ERR', str(ex)[:80])
