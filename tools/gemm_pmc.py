#!/usr/bin/env python3
"""A few launches of each big GEMM variant, for `rocprofv3 --pmc … -- python3 tools/gemm_pmc.py` (stall / LDS counters)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
CASES = [('NT', 16384, 5376, 1792, 224), ('NN', 16384, 1792, 5376, 224), ('TN', 5376, 1792, 16384, 128), ('TN', 5376, 1792, 16384, 224),
         ('TT', 5376, 1792, 16384, 224), ('NT', 8192, 8192, 8192, 256)]
for lay, M, N, K, tile in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    shp = {'NT': ((M, K), (N, K)), 'NN': ((M, K), (K, N)), 'TN': ((K, M), (K, N)), 'TT': ((K, M), (N, K))}[lay]
    A = torch.randn(*shp[0], device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(*shp[1], device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if lay in ('TN', 'TT') else torch.bfloat16, device=dev)
    L.check(L.lib.mts_set_option(b'gemm_tile', tile))
    for _ in range(3):
        ops.gemm(getattr(L, lay), A, B, out, M=M, N=N, K=K)
    torch.cuda.synchronize()
    del A, B, out
