#!/usr/bin/env python3
"""Weight-gradient GEMM (TN, 128x128 tile) over K splits."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
for M, N, K in [(5376, 1792, 16384), (1792, 1792, 16384), (1792, 256, 16384), (256, 1792, 16384)]:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(K, M, device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(K, N, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev)
    ref = None
    for tile in (128, 224, 256):
        for sp in (1, 2, 3, 4, 6, 8):
            L.check(L.lib.mts_set_option(b'gemm_tile', tile))
            L.check(L.lib.mts_set_option(b'gemm_splits', sp))
            try:
                for _ in range(2):
                    ops.gemm(L.TN, A, B, out, M=M, N=N, K=K)
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(10):
                    ops.gemm(L.TN, A, B, out, M=M, N=N, K=K)
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) * 100
                if ref is None:
                    ref = out.clone()
                err = float((out - ref).abs().max() / ref.abs().max())
                print(f'M={M:5d} N={N:5d} tile={tile:3d} splits={sp}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s  rel.diff {err:.1e}', flush=True)
            except Exception as ex:
                print(M, N, tile, sp, 'ERR', str(ex)[:80])
L.check(L.lib.mts_set_option(b'gemm_tile', 0))
L.check(L.lib.mts_set_option(b'gemm_splits', 0))
