#!/usr/bin/env python3
"""Epilogue cost on the big 224-tile GEMMs: plain vs bias vs bias+residual (NT Wo forward) and plain vs residual (NN QKV dgrad)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
M = 16384


def t(fn):
    best = 1e9
    for _ in range(3):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 100)
    return best


x1792 = torch.randn(M, 1792, device=dev).to(torch.bfloat16)
x5376 = torch.randn(M, 5376, device=dev).to(torch.bfloat16)
res = torch.randn(M, 1792, device=dev).to(torch.bfloat16)
o1792 = torch.empty(M, 1792, device=dev, dtype=torch.bfloat16)
o5376 = torch.empty(M, 5376, device=dev, dtype=torch.bfloat16)
wo = torch.randn(1792, 1792, device=dev).to(torch.bfloat16)
wqkv = torch.randn(5376, 1792, device=dev).to(torch.bfloat16)
b1792 = torch.randn(1792, device=dev)
b5376 = torch.randn(5376, device=dev)
print('NT Wo   plain         %.1f us' % t(lambda: ops.gemm(L.NT, x1792, wo, o1792, M=M, N=1792, K=1792)))
print('NT Wo   bias          %.1f us' % t(lambda: ops.gemm(L.NT, x1792, wo, o1792, M=M, N=1792, K=1792, bias=b1792)))
print('NT Wo   bias+residual %.1f us' % t(lambda: ops.gemm(L.NT, x1792, wo, o1792, M=M, N=1792, K=1792, bias=b1792, residual=res)))
print('NT QKV  plain         %.1f us' % t(lambda: ops.gemm(L.NT, x1792, wqkv, o5376, M=M, N=5376, K=1792)))
print('NT QKV  bias+colscale %.1f us' % t(lambda: ops.gemm(L.NT, x1792, wqkv, o5376, M=M, N=5376, K=1792, bias=b5376, colscale=0.0668, ncols_scaled=1792)))
print('NN dQKV plain         %.1f us' % t(lambda: ops.gemm(L.NN, x5376, wqkv, o1792, M=M, N=1792, K=5376)))
print('NN dQKV residual      %.1f us' % t(lambda: ops.gemm(L.NN, x5376, wqkv, o1792, M=M, N=1792, K=5376, residual=res)))
print('NN dWo  plain         %.1f us' % t(lambda: ops.gemm(L.NN, x1792, wo, o1792, M=M, N=1792, K=1792)))
