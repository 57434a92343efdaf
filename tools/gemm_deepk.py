#!/usr/bin/env python3
"""The 256x224 NT kernels (gemm_variant 0 = four waves, persistent; 9 = four waves, one tile per workgroup; 6 = eight waves) and the vendor library at a deep-K
shape where tile prologues / epilogues vanish: NT 8192 x 8064 x 8192.  One process, best of 3 x 10 launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops
dev = 'cuda'
M, N, K = int(os.environ.get("MTS_M", 8192)), int(os.environ.get("MTS_N", 8064)), int(os.environ.get("MTS_K", 8192))
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
B = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
def t(fn):
    best = 1e9
    for _ in range(3):
        fn(); fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / 10)
    return best
fl = 2.0 * M * N * K / 1e6
L.check(L.lib.mts_set_option(b'gemm_tile', 224))
for v in (0, 9, 6):
    L.check(L.lib.mts_set_option(b'gemm_variant', v))
    us = t(lambda: ops.gemm(L.NT, A, B, out, M=M, N=N, K=K))
    print(f'gemm_variant {v}: {us:7.1f} us  {fl / us:7.1f} TF/s', flush=True)
L.lib.mts_set_option(b'gemm_variant', 0); L.lib.mts_set_option(b'gemm_tile', 0)
us = t(lambda: torch.matmul(A, B.t()))
print(f'torch.matmul  : {us:7.1f} us  {fl / us:7.1f} TF/s', flush=True)
