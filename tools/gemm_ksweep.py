#!/usr/bin/env python3
"""Time vs K at fixed M,N for both tile kernels: separates per-K cost from per-tile fixed overhead (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops
dev = 'cuda'
def run(M, N, K, reps=10):
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(2):
        ops.gemm(L.NT, A, B, out, M=M, N=N, K=K)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm(L.NT, A, B, out, M=M, N=N, K=K)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
for (M, N) in [(16384, 5376), (16384, 1792), (8192, 2048)]:
    for tile in (128, 256):
        L.lib.mts_set_option(b'gemm_tile', tile)
        row = [f'{K}:{run(M, N, K):.0f}' for K in (512, 1024, 1792, 3584, 7168)]
        print(f'NT {M}x{N} tile{tile}: ' + ' '.join(row), flush=True)
