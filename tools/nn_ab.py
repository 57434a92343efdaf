#!/usr/bin/env python3
"""Data-gradient (NN, bf16 C) GEMMs of the step, in one process: the four-wave kernel (gemm224n.hip, gemm_variant 10 = for every epilogue; the default takes it without a residual)
against the eight-wave kernel (gemm_variant 6) and torch.matmul, each at steady state under its own load (100 warm launches, median of 3 x 50), twice, alternating."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402
from tools.blas_compare_util import timeit  # noqa: E402

dev = 'cuda'
for M, N, K, with_res in [(16384, 1792, 5376, True), (16384, 1792, 5376, False), (16384, 1792, 1792, False), (16384, 1792, 1792, True), (8192, 7168, 8192, False)]:
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(K, N, device=dev, generator=g).to(torch.bfloat16)
    R = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if with_res else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    def ours(variant):
        def f():
            ops.gemm(L.NN, A, B, out, M=M, N=N, K=K, residual=R)
        def run():
            L.check(L.lib.mts_set_option(b'gemm_variant', variant))
            try:
                return timeit(f)
            finally:
                L.check(L.lib.mts_set_option(b'gemm_variant', 0))
        return run

    legs = [('four-wave', ours(10)), ('eight-wave', ours(6)), ('torch.matmul', lambda: timeit(lambda: torch.matmul(A, B)))]
    res = {name: [] for name, _ in legs}
    for _ in range(2):
        for name, run in legs:
            res[name].append(run())
    fl = 2.0 * M * N * K / 1e6
    print(f'NN M={M:6d} N={N:5d} K={K:6d} residual={int(with_res)}  ' + '   '.join(
        f'{name} {min(v):7.1f} us ({fl / min(v):6.1f} TF/s) [{v[0]:.1f}, {v[1]:.1f}]' for name, v in res.items()), flush=True)
