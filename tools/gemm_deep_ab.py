#!/usr/bin/env python3
"""A/B of the 128x128 kernel's four-buffer copy pipeline ("gemm_deep") on the feed-forward shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
cases = [('NT', 16384, 256, 1792), ('NN', 16384, 256, 1792), ('NT', 16384, 1792, 256), ('NN', 16384, 1792, 256), ('NT', 4096, 512, 1024)]
for lay, M, N, K in cases:
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    B = (torch.randn(N, K, device=dev, generator=g) if lay == 'NT' else torch.randn(K, N, device=dev, generator=g)).to(torch.bfloat16)
    outs = []
    for deep in (0, 1):
        L.check(L.lib.mts_set_option(b'gemm_deep', deep))
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(getattr(L, lay), A, B, out, M=M, N=N, K=K)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            ops.gemm(getattr(L, lay), A, B, out, M=M, N=N, K=K)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 50
        outs.append(out.float())
        print(f'{lay} M={M} N={N} K={K} deep={deep}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s', flush=True)
    print('   max |diff| deep vs not:', float((outs[0] - outs[1]).abs().max()), flush=True)
L.check(L.lib.mts_set_option(b'gemm_deep', 1))
