#!/usr/bin/env python3
"""torch.matmul (hipBLASLt / rocBLAS underneath) on the step's big GEMM shapes, next to mts_gemm: how far the hand-written
kernels are from the vendor library on this chip (same process; per shape ours, the vendor's, ours again, the vendor's again -- each after 100 warm launches of its own, median
of 3 x 50 timed launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
SHAPES = [('NT', 16384, 5376, 1792), ('NT', 16384, 1792, 1792), ('NN', 16384, 1792, 5376), ('NN', 16384, 1792, 1792),
          ('TN', 5376, 1792, 16384), ('TN', 1792, 1792, 16384), ('NT', 8192, 8192, 8192)]


from tools.blas_compare_util import timeit  # noqa: E402

for lay, M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    shp = {'NT': ((M, K), (N, K)), 'NN': ((M, K), (K, N)), 'TN': ((K, M), (K, N))}[lay]
    A = torch.randn(*shp[0], device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(*shp[1], device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if lay == 'TN' else torch.bfloat16, device=dev)
    f_ours = lambda: ops.gemm(getattr(L, lay), A, B, out, M=M, N=N, K=K)
    if lay == 'NT':
        f_ref = lambda: torch.matmul(A, B.t())
    elif lay == 'NN':
        f_ref = lambda: torch.matmul(A, B)
    else:
        f_ref = lambda: torch.matmul(A.t(), B)          # bf16 output (the library has no fp32-out path through torch)
    o1, r1, o2, r2 = timeit(f_ours), timeit(f_ref), timeit(f_ours), timeit(f_ref)
    ours, ref = min(o1, o2), min(r1, r2)
    fl = 2.0 * M * N * K / 1e6
    print(f'{lay} M={M:6d} N={N:5d} K={K:6d}  mts {ours:7.1f} us ({fl / ours:6.1f} TF/s) [{o1:.1f}, {o2:.1f}]   torch.matmul {ref:7.1f} us ({fl / ref:6.1f} TF/s) '
          f'[{r1:.1f}, {r2:.1f}]   mts / vendor time {ours / ref:.3f}', flush=True)
